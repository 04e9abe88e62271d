"""MI355X-native RecNet train-step path (gfx950 HIP kernels behind the reference's Python API).

Host-side mirror of the reference interface for the hot path:
  models/decoder.py  Decoder                 -> modules.Decoder
  models/global_reconstructor.py             -> modules.GlobalReconstructor
  models/local_reconstructor.py              -> modules.LocalReconstructor
  train.py forward_decoder / forward_*_reconstructor / build_decoder / build_reconstructor
                                             -> api.*
  train.py:248-273 (the step body)           -> api.TrainStep (single stream, hipGraph-capturable)
  eval.py greedy_search / beam_search        -> search.* (device-side loops)
All compute goes through csrc/librecnet_hip.so (C ABI: include/recnet_hip.h).
"""
from .config import TrainConfig, make_config  # noqa: F401
from .modules import Decoder, GlobalReconstructor, LocalReconstructor  # noqa: F401
from .api import (build_decoder, build_reconstructor, forward_decoder, forward_global_reconstructor,  # noqa: F401
                  forward_local_reconstructor, clip_grad_norm_, TrainStep, GraphedStep, FusedAdam, decode_len,
                  step_weights)
from .dp import DataParallelTrainStep, shard_bounds  # noqa: F401
from .search import greedy_search, beam_search  # noqa: F401
from .loop import Trainer, evaluate  # noqa: F401
from .checkpoint import save_checkpoint, load_checkpoint, read_checkpoint  # noqa: F401
