"""Size-independent properties of the train step at the benchmark's FULL size (BASELINE.json configs[1]/[2]: B=100,
28x1536 features, V=4188, E=468, H=512, A=128, R=1536, T=31).  (Element-wise comparison with the oracle at these sizes is
tests/test_gpu_configs.py; these are the properties that hold whatever the checker.)

  * caption order does not matter: permuting the batch (dropout off) leaves the losses and every gradient unchanged;
  * the batch shards exactly: a 60 + 40 split with the GLOBAL normalisers and the global caption index in the dropout
    masks gives, summed, the gradients of the whole batch — dropout on;
  * the step is a pure function of (parameters, batch, seed): replaying it gives the same losses bit for bit and the
    same gradients — bit for bit for every GEMM-produced tensor, to fp32 rounding for the ones summed with atomics
    (bias / attn_b column sums, the embedding scatter).
"""
import numpy as np
import pytest
import torch

import recnet_amd as R
from recnet_amd.synthetic import synthetic_features, synthetic_targets
from tests import golden_util as GU
from tests.gpu_util import make_models, rel_err

pytestmark = pytest.mark.gpu

B, F, D, V, E, H, A, RA = 100, 28, 1536, 4188, 468, 512, 128, 128
DIMS = [B, F, D, V, E, H, A, RA]


def _models(kind, prec, batch, C_over=None):
    decP = GU.formula_params(GU.decoder_shapes(V, E, H, A, D), 21)
    recP = GU.formula_params(GU.rec_shapes(kind, H, D, RA), 22)
    C, dec, rec = make_models([batch] + DIMS[1:], kind, prec, decP, recP)
    if C_over:
        for k, v in C_over.items():
            setattr(C, k, v)
            dec["_hyper"][k] = v
            rec["_hyper"][k] = v
    return C, dec, rec


def _grads(md):
    return {k: v.clone() for k, v in md["_state"].flat()["grad"].views.items()}


@pytest.mark.parametrize("kind", ["global", "local"])
def test_caption_order_does_not_matter(kind):
    nodrop = dict(embedding_dropout=0.0, decoder_out_dropout=0.0, reconstructor_decoder_dropout=0.0)
    C, dec, rec = _models(kind, "f32", B, nodrop)
    enc = synthetic_features(B, F, D, seed=5).cuda()
    tg = synthetic_targets(B, V, seed=5)
    step = R.TrainStep(dec, rec)
    T, w = step.prepare(tg.numpy())
    step.fwd_bwd(enc, tg.cuda(), T, w, seed=1)
    torch.cuda.synchronize()
    s0 = step.engine.scalar_dict()
    g0 = {"dec": _grads(dec), "rec": _grads(rec)}
    perm = torch.from_numpy(np.random.RandomState(0).permutation(B))
    step.fwd_bwd(enc[perm.cuda()].contiguous(), tg[:, perm].contiguous().cuda(), T, w, seed=1)
    torch.cuda.synchronize()
    s1 = step.engine.scalar_dict()
    for k in ("dec_ce", "rec_mse", "total_loss"):
        assert abs(s0[k] - s1[k]) <= 2e-6 * abs(s0[k]), k
    for grp, md in (("dec", dec), ("rec", rec)):
        for k, v in _grads(md).items():
            assert rel_err(v.cpu().numpy(), g0[grp][k].cpu().numpy()) <= 2e-5, (grp, k)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_batch_shards_exactly_with_global_normalisers(prec):
    kind = "global"
    C, dec, rec = _models(kind, prec, B)
    enc = synthetic_features(B, F, D, seed=6).cuda()
    tg = synthetic_targets(B, V, seed=6)
    full = R.TrainStep(dec, rec)
    T, w = full.prepare(tg.numpy())
    full.fwd_bwd(enc, tg.cuda(), T, w, seed=9)
    torch.cuda.synchronize()
    ref = {"dec": _grads(dec), "rec": _grads(rec)}
    sref = full.engine.scalar_dict()
    acc = None
    ce = mse = 0.0
    for lo, hi in ((0, 60), (60, 100)):
        sh = R.TrainStep(dec, rec, batch_size=hi - lo, global_batch=B, batch_offset=lo)
        sh.fwd_bwd(enc[lo:hi].contiguous(), tg[:, lo:hi].contiguous().cuda(), T, w, seed=9)
        torch.cuda.synchronize()
        g = {"dec": _grads(dec), "rec": _grads(rec)}
        acc = g if acc is None else {grp: {k: acc[grp][k] + g[grp][k] for k in g[grp]} for grp in g}
        sc = sh.engine.scalar_dict()
        ce, mse = ce + sc["dec_ce"], mse + sc["rec_mse"]
    tol = 2e-5 if prec == "f32" else 2e-2       # bf16: the shards round their operands in a different tile context
    assert abs(ce - sref["dec_ce"]) <= (1e-5 if prec == "f32" else 1e-3) * abs(sref["dec_ce"])
    assert abs(mse - sref["rec_mse"]) <= (1e-5 if prec == "f32" else 1e-3) * abs(sref["rec_mse"])
    for grp in ref:
        for k in ref[grp]:
            assert rel_err(acc[grp][k].cpu().numpy(), ref[grp][k].cpu().numpy()) <= tol, (grp, k)


def test_step_is_a_pure_function_of_its_inputs():
    C, dec, rec = _models("global", "bf16", B)
    enc = synthetic_features(B, F, D, seed=7).cuda()
    tg = synthetic_targets(B, V, seed=7)
    step = R.TrainStep(dec, rec)
    T, w = step.prepare(tg.numpy())
    outs = []
    for _ in range(2):
        step.fwd_bwd(enc, tg.cuda(), T, w, seed=3)
        torch.cuda.synchronize()
        outs.append(({k: v.clone() for k, v in _grads(dec).items()}, {k: v.clone() for k, v in _grads(rec).items()},
                     step.engine.scalars.clone()))
    for a, b in zip(outs[0][:2], outs[1][:2]):
        for k in a:
            if a[k].dim() == 1 or a[k].shape[0] == 1 or k == "embedding.weight":   # column sums / scatter-add use atomics
                assert rel_err(a[k].cpu().numpy(), b[k].cpu().numpy()) <= 1e-5, k
            else:
                assert torch.equal(a[k], b[k]), k
    assert torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("kind", ["global", "local"])
@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_chain_kernels_over_changing_batches_match_per_step_kernels(cell, kind, monkeypatch):
    """The persistent chain kernels keep per-time-step exchange buffers in the workspace and do not invalidate caches at
    their barriers (csrc/rec_chain.hpp, RC_ACQUIRE_INV): run three DIFFERENT batches through one engine — every
    buffer then holds the previous batch's data when the next one starts — and hold each against an engine running the
    per-step kernels.  A stale line anywhere would show up as an O(1) error, not as rounding."""
    decP = GU.formula_params(GU.decoder_shapes(V, E, H, A, D, cell), 21)
    recP = GU.formula_params(GU.rec_shapes(kind, H, D, RA, cell), 22)
    _, dec_a, rec_a = make_models(list(DIMS), kind, "bf16", decP, recP, cells=(cell, cell))
    step_a = R.TrainStep(dec_a, rec_a)
    monkeypatch.setenv("RN_PER_STEP", "rec,dec,loc,loc_bwd")
    _, dec_b, rec_b = make_models(list(DIMS), kind, "bf16", decP, recP, cells=(cell, cell))
    step_b = R.TrainStep(dec_b, rec_b)
    for seed in (11, 12, 13):
        enc = synthetic_features(B, F, D, seed=seed).cuda()
        tg = synthetic_targets(B, V, seed=seed)
        T, w = step_a.prepare(tg.numpy())
        for st in (step_a, step_b):
            st.fwd_bwd(enc, tg.cuda(), T, w, seed=seed)
        torch.cuda.synchronize()
        sa, sb = step_a.engine.scalar_dict(), step_b.engine.scalar_dict()
        for k in ("dec_ce", "rec_mse"):
            assert abs(sa[k] - sb[k]) <= 2e-4 * abs(sb[k]), (seed, k, sa[k], sb[k])
        for grp, ma, mb in (("dec", dec_a, dec_b), ("rec", rec_a, rec_b)):
            ga, gb = _grads(ma), _grads(mb)
            for k in ga:
                assert rel_err(ga[k].cpu().numpy(), gb[k].cpu().numpy()) <= 1e-2, (seed, grp, k)
