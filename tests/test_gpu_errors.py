"""Error behaviour of the boundary on a GPU box: the library reports misuse through return codes (surfaced as
RecNetError with recnet_last_error()'s text), the Python mirror raises like the reference does — nothing falls back
silently to another path."""
import numpy as np
import pytest
import torch

import recnet_amd as R
from recnet_amd import _lib
from recnet_amd.engine import Engine
from tests import golden_util as GU
from tests.gpu_util import make_models

pytestmark = pytest.mark.gpu

DIMS = [4, 5, 40, 37, 10, 24, 16, 16]


def _models(kind="global"):
    B, F, D, V, E, H, A, RA = DIMS
    decP = GU.formula_params(GU.decoder_shapes(V, E, H, A, D), 1)
    recP = GU.formula_params(GU.rec_shapes(kind, H, D, RA), 2)
    return make_models(DIMS, kind, "f32", decP, recP)


def test_wrong_shapes_devices_and_dtypes_are_rejected():
    C, dec, rec = _models()
    step = R.TrainStep(dec, rec)
    enc, tg = GU.make_batch(4, 5, 40, 37, [3, 1, 4, 2], 3)
    T, w = step.prepare(tg.numpy())
    with pytest.raises(RuntimeError):                       # CPU tensor: there is no CPU path
        step(enc, tg.cuda(), T, w)
    with pytest.raises(RuntimeError):                       # wrong batch size
        step(enc[:3].cuda(), tg.cuda(), T, w)
    with pytest.raises(RuntimeError):                       # targets must be int64 [31, B]
        step(enc.cuda(), tg.cuda().int(), T, w)
    with pytest.raises(RuntimeError):
        step(enc.cuda(), tg[:10].cuda(), T, w)


def test_library_return_codes():
    C, dec, rec = _models()
    step = R.TrainStep(dec, rec)
    eng = step.engine
    enc, tg = GU.make_batch(4, 5, 40, 37, [3, 1, 4, 2], 3)
    encd, tgd = enc.cuda(), tg.cuda()
    T, w = step.prepare(tg.numpy())
    with pytest.raises(_lib.RecNetError, match="T out of range"):
        eng.train_step_fwd_bwd(encd, tgd, 32, torch.ones(32, device="cuda"), 1)
    with pytest.raises(_lib.RecNetError, match="before forward"):
        eng.backward_decoder(encd, tgd, None, 1.0)
    with pytest.raises(_lib.RecNetError, match="beam_width"):
        eng.beam_search(encd, 9)
    # a handle without a reconstructor refuses reconstructor calls
    e2 = Engine(dict(B=4, F=5, D=40, E=10, H=24, A=16, V=37), None, "f32")
    with pytest.raises(_lib.RecNetError):
        e2.forward_reconstructor(encd, None, T)
    # and an unbound handle refuses to run
    with pytest.raises(_lib.RecNetError, match="not bound"):
        e2.train_step_fwd_bwd(encd, tgd, T, w, 1)


def test_config_validation():
    with pytest.raises(_lib.RecNetError, match="reconstructor_hidden_size == encoder_output_size"):
        Engine(dict(B=2, F=2, D=8, E=4, H=8, A=4, V=8, R=16, RA=4), "local", "bf16")
    with pytest.raises(_lib.RecNetError, match="non-positive"):
        Engine(dict(B=0, F=2, D=8, E=4, H=8, A=4, V=8), None, "bf16")
    with pytest.raises(KeyError):
        Engine(dict(B=2, F=2, D=8, E=4, H=8, A=4, V=8, dec_cell="RNN"), None, "bf16")
