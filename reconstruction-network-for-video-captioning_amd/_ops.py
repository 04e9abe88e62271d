"""torch.ops.recnet.* — loads csrc/librecnet_torch_ops.so, the TORCH_LIBRARY registration of the hot path over the C ABI
(csrc/torch_ops.cpp).  Like the C-ABI library itself there is no fallback: a missing
extension raises."""
import os

import torch

from . import _lib

OPS_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "librecnet_torch_ops.so")
OP_NAMES = ("forward_decoder", "forward_decoder_free", "backward_decoder", "forward_reconstructor", "backward_reconstructor",
            "add_reg_grad", "train_step_fwd_bwd", "train_step", "optimizer_step", "clip_grad_norm", "decoder_step",
            "reconstructor_step", "greedy_search", "beam_search")
_loaded = False


def load():
    """Registers the `recnet` op namespace (idempotent) and returns torch.ops.recnet."""
    global _loaded
    if not _loaded:
        _lib.load()
        if not os.path.exists(OPS_PATH):
            raise _lib.RecNetLibraryError(
                "custom-op extension %s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(csrc/build_torch_ops.py). There is no fallback." % OPS_PATH)
        torch.ops.load_library(OPS_PATH)
        for n in OP_NAMES:
            if not hasattr(torch.ops.recnet, n):
                raise _lib.RecNetLibraryError("librecnet_torch_ops.so does not register recnet::%s (stale build?)" % n)
        _loaded = True
    return torch.ops.recnet
