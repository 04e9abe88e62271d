"""Round-4 switches of the fused step, each held against the default: the grouped GEMM launches (RN_GEMM_GROUP), the Adam update in
the epilogue of the pending d W_hh product (RN_ADAM_EPILOGUE), the decoder forward chain with the attention projection formed by
the caption's own workgroup (RN_ALT=dec_wh_in_phase_a switches back), the residency waits of the side branches (RN_WAIT_CHAIN) and the MSE + d loss / d out
of the local reconstructor's output layer in that product's epilogue (RN_MSE_EPILOGUE); round 5: the decoder chains' hand-over
inside a row part (RN_ALT=dec_relayed_barrier) and the forward chain's phase A tiled by row parts (RN_ALT=dec_all_rows).  (Round 6: the vendor-library switches of
round 5 are gone with the library — every product runs on the hand-written kernels.)  Every switch changes
the SCHEDULE or the summation order of a product, never the arithmetic: parameters after four replayed steps (split reconstructor
update, flushed) agree to rounding with the default's, the losses of every step to 1e-4 (bf16 operands).
Both optimisers run at a learning rate of 1e-2 here (at the defaults 1e-5 / 1e-6 four steps move a weight by less than
the comparison tolerance, so a missing update or a stale operand image would have passed): every step moves every parameter by
~1e-2, the losses of steps 2-4 are those of the updated weights, and the parameters are compared relative to how far they moved."""
import os

import numpy as np
import pytest
import torch

import recnet_amd as R
from tests import golden_util as GU
from tests.gpu_util import make_models

pytestmark = pytest.mark.gpu

DIMS = [24, 6, 64, 61, 16, 32, 16, 16]      # persistent-chain shape (H % 32 == 0, R % 32 == 0)
LR = 1e-2


def _run(kind, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        B, F, D, V, E, H, A, RA = DIMS
        decP = GU.formula_params(GU.decoder_shapes(V, E, H, A, D), 3)
        recP = GU.formula_params(GU.rec_shapes(kind, H, D, RA), 4)
        _, dec, rec = make_models(list(DIMS), kind, "bf16", decP, recP, decoder_learning_rate=LR, reconstructor_learning_rate=LR)
        step = R.DataParallelTrainStep(dec, rec, B, 0, 1, n_frames=F)
        rs = np.random.RandomState(2)
        enc, targets = GU.make_batch(B, F, D, V, [30] + [int(x) for x in rs.randint(1, 30, size=B - 1)], 11)
        T, w = step.prepare(targets.numpy())
        g = R.GraphedStep(step, enc.cuda(), targets.cuda(), T, w, warmup=0, defer_reconstructor_update="recurrent")
        losses = [g().clone() for _ in range(4)]
        g.flush()
        torch.cuda.synchronize()
        assert step.step_impl.engine.chain_status() == 0
        assert step.step_impl.engine.images_stale() == 0
        out = {}
        for name, md, P in (("dec", dec, decP), ("rec", rec, recP)):
            for k, v in md["model"].state_dict().items():
                out[name + "." + k] = (v.detach().double().cpu(), P[k].double())
        return out, torch.stack(losses).cpu().numpy()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("kind", ["global", "local"])
@pytest.mark.parametrize("knob", ["RN_GEMM_GROUP=0", "RN_ADAM_EPILOGUE=0", "RN_ALT=dec_wh_in_phase_a", "RN_WAIT_CHAIN=0", "RN_MSE_EPILOGUE=0",
                                  "RN_ALT=dec_relayed_barrier", "RN_ALT=dec_all_rows"])
def test_switch_off_equals_default(knob, kind):
    p0, l0 = _run(kind, {})
    p1, l1 = _run(kind, dict([knob.split("=")]))
    assert np.allclose(l0[:, :7], l1[:, :7], rtol=1e-4, atol=0), (knob, l0[:, 6], l1[:, 6])
    # the comparison has teeth: the updates moved the loss by far more than its tolerance ...
    assert abs(l0[3, 6] - l0[0, 6]) > 100 * 1e-4 * abs(l0[0, 6]), l0[:, 6]
    for k in p0:
        (a, init), (b, _) = p0[k], p1[k]
        moved = float((a - init).norm())
        # ... and every parameter tensor by a sizeable fraction of lr per element and step
        assert moved > 3e-3 * np.sqrt(a.numel()), (knob, k, moved)
        # a different summation order changes a gradient by fp32 rounding; Adam turns that into a different update only where a
        # gradient element is within rounding of zero
        assert float((a - b).norm()) <= 2e-3 * moved, (knob, k, float((a - b).norm()), moved)
