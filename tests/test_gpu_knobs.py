"""Round-4 switches of the fused step, each held against the default: the grouped GEMM launches (RN_GEMM_GROUP), the Adam update in
the epilogue of the pending d W_hh product (RN_ADAM_EPILOGUE), the decoder forward chain with the attention projection formed by
the caption's own workgroup (RN_DEC_LOCAL_WH), the residency waits of the side branches (RN_WAIT_CHAIN) and the MSE + d loss / d out
of the local reconstructor's output layer in that product's epilogue (RN_MSE_EPILOGUE).  Every switch changes
the SCHEDULE or the summation order of a product, never the arithmetic: parameters after four replayed steps (split reconstructor
update, flushed) agree to rounding with the default's, the losses of every step to 1e-4 (bf16 operands)."""
import os

import numpy as np
import pytest
import torch

import recnet_amd as R
from tests import golden_util as GU
from tests.gpu_util import make_models

pytestmark = pytest.mark.gpu

DIMS = [24, 6, 64, 61, 16, 32, 16, 16]      # persistent-chain shape (H % 32 == 0, R % 32 == 0)


def _run(kind, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        B, F, D, V, E, H, A, RA = DIMS
        decP = GU.formula_params(GU.decoder_shapes(V, E, H, A, D), 3)
        recP = GU.formula_params(GU.rec_shapes(kind, H, D, RA), 4)
        _, dec, rec = make_models(list(DIMS), kind, "bf16", decP, recP)
        step = R.DataParallelTrainStep(dec, rec, B, 0, 1, n_frames=F)
        rs = np.random.RandomState(2)
        enc, targets = GU.make_batch(B, F, D, V, [30] + [int(x) for x in rs.randint(1, 30, size=B - 1)], 11)
        T, w = step.prepare(targets.numpy())
        g = R.GraphedStep(step, enc.cuda(), targets.cuda(), T, w, warmup=0, defer_reconstructor_update="recurrent")
        losses = [g().clone() for _ in range(4)]
        g.flush()
        torch.cuda.synchronize()
        assert step.step_impl.engine.chain_status() == 0
        out = {}
        for name, md in (("dec", dec), ("rec", rec)):
            for k, v in md["model"].state_dict().items():
                out[name + "." + k] = v.detach().clone()
        return out, torch.stack(losses).cpu().numpy()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("kind", ["global", "local"])
@pytest.mark.parametrize("knob", ["RN_GEMM_GROUP", "RN_ADAM_EPILOGUE", "RN_DEC_LOCAL_WH", "RN_WAIT_CHAIN", "RN_MSE_EPILOGUE"])
def test_switch_off_equals_default(knob, kind):
    p0, l0 = _run(kind, {})
    p1, l1 = _run(kind, {knob: "0"})
    assert np.allclose(l0[:, :7], l1[:, :7], rtol=1e-4, atol=0), (knob, l0[:, 6], l1[:, 6])
    for k in p0:
        assert torch.allclose(p0[k], p1[k], rtol=5e-4, atol=2e-7), (knob, k, float((p0[k] - p1[k]).abs().max()))
    # and the parameters moved at all
    assert any(not torch.equal(p0[k], torch.zeros_like(p0[k])) for k in p0)
