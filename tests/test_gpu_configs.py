"""Every BASELINE.json configuration at its FULL per-GPU size through the HIP path, held element by element against the
CPU oracle (one train step forward + backward, dropout on with the product's counter-based masks):

  C2  decoder + global reconstructor, B=100, 28x1536                         (configs[1], the headline)
  C3  decoder + local reconstructor (attn 128), B=100, 28x1536               (configs[2])
  C4  decoder + local reconstructor, 40x2048, 32 captions per rank of 256    (configs[3]; global normalisers, rank 3's shard)
  C5  decoder + local reconstructor, 28x3584, 64 captions per rank of 512    (configs[4]; rank 5's shard)

in both precisions.  Compared: the CE and MSE parts of the losses, and EVERY element of every gradient tensor
(relative L2 error and cosine per tensor; the goldens at B=8 store norms and slices only).  The oracle takes 2-20 s per
case on the GPU box's host cores.  Reference: train.py:17-131, 248-268.
"""
import numpy as np
import pytest
import torch

import recnet_amd as R
from oracle import recnet_oracle as O
from recnet_amd.synthetic import synthetic_features, synthetic_targets
from tests import golden_util as GU
from tests.gpu_util import TOL, cosine, make_models, rel_err

pytestmark = pytest.mark.gpu

V, E, H, A, RA = 4188, 468, 512, 128, 128
# name: (kind, per-rank B, F, D, global B, batch offset)
CONFIGS = {
    "C2": ("global", 100, 28, 1536, 100, 0),
    "C3": ("local", 100, 28, 1536, 100, 0),
    "C4": ("local", 32, 40, 2048, 256, 96),
    "C5": ("local", 64, 28, 3584, 512, 320),
}


def oracle_case(kind, B, F, D, Bg, off, seed):
    """Oracle gradients of rank-local captions [off, off+B) of a global batch Bg with the GLOBAL normalisers
    (SURVEY.md section 8e): per-step counts n_t and N from the global targets, MSE count from the global batch."""
    decP = GU.formula_params(GU.decoder_shapes(V, E, H, A, D), 31)
    recP = GU.formula_params(GU.rec_shapes(kind, H, D, RA), 32)
    tg_g = synthetic_targets(Bg, V, seed=77)
    enc = synthetic_features(B, F, D, seed=78)
    tg = tg_g[:, off:off + B].contiguous()
    masks_g = tg_g > 0
    T = O.decode_len(masks_g)
    n_t = [int(masks_g[t].sum()) for t in range(T)]
    st = O.TrainState(decP, recP, kind)
    drop = O.Dropper("hash", seed=seed, B_global=Bg, b_offset=off)
    # lambda_reg = 0 here: the regulariser's gradient lambda * p / ||p|| is added below with the norm in float64.  The
    # reference's own `torch.norm` (train.py:69,103,129) on the CPU in float32 is off by 2e-4 (9 M elements) to 3.2e-3
    # (the 51 M elements of C5's W_hh) relative; with 1/8 of the data gradient per rank that error alone exceeds the
    # fp32 parity bar, and it is the oracle's error, not the product's (the device norm agrees with float64 to 1e-6).
    dl, hid, _, ce, _ = O.forward_decoder(st.dec, enc, tg, tg > 0, drop=drop, global_counts=(n_t, sum(n_t)), lambda_reg=0.0,
                                          return_parts=True)
    if kind == "global":
        rl, mse, _ = O.forward_global_reconstructor(st.rec, hid, enc, drop=drop, mse_count=Bg * D, lambda_reg=0.0, return_parts=True)
    else:
        rl, mse, _ = O.forward_local_reconstructor(st.rec, hid, enc, drop=drop, mse_count=Bg * F * D, lambda_reg=0.0, return_parts=True)
    (dl + rl).backward()
    def with_reg(P, lam):
        return {k: (v.grad.double() + lam * v.detach().double() / v.detach().double().norm()).numpy() for k, v in P.items()}
    ref = dict(ce=float(ce.detach()), mse=float(mse.detach()), dec=with_reg(st.dec, 1e-3), rec=with_reg(st.rec, 1e-2))
    return decP, recP, enc, tg, tg_g, ref


@pytest.fixture(scope="module", params=list(CONFIGS))
def case(request):
    name = request.param
    kind, B, F, D, Bg, off = CONFIGS[name]
    torch.set_num_threads(min(32, torch.get_num_threads() * 4))
    return (name,) + oracle_case(kind, B, F, D, Bg, off, seed=5)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_full_size_step_matches_oracle_elementwise(case, prec):
    name, decP, recP, enc, tg, tg_g, ref = case
    kind, B, F, D, Bg, off = CONFIGS[name]
    _, dec, rec = make_models([B, F, D, V, E, H, A, RA], kind, prec, decP, recP)
    step = R.TrainStep(dec, rec, batch_size=B, n_frames=F, global_batch=Bg, batch_offset=off)
    T, w = step.prepare(tg_g.numpy())
    step.engine.poison_lds()
    step.fwd_bwd(enc.cuda(), tg.cuda(), T, w, seed=5)
    step.engine.add_reg_grad(0, 1.0)
    step.engine.add_reg_grad(1, 1.0)
    torch.cuda.synchronize()
    sc = step.engine.scalar_dict()
    tol = TOL[prec]
    assert abs(sc["dec_ce"] - ref["ce"]) <= tol["loss"] * abs(ref["ce"]), (name, prec, sc["dec_ce"], ref["ce"])
    assert abs(sc["rec_mse"] - ref["mse"]) <= tol["loss"] * abs(ref["mse"]), (name, prec, sc["rec_mse"], ref["mse"])
    worst = {}
    for grp, md in (("dec", dec), ("rec", rec)):
        gv = md["_state"].flat()["grad"].views
        for k in gv:
            got = gv[k].cpu().numpy()
            assert got.shape == ref[grp][k].shape and np.isfinite(got).all(), (name, prec, grp, k)
            worst[grp + "." + k] = (rel_err(got, ref[grp][k]), cosine(got, ref[grp][k]))
    bad = {k: v for k, v in worst.items() if v[0] > tol["grad"] or v[1] < tol["cos"]}
    print("%s %s: ce %.6f/%.6f mse %.6f/%.6f worst grad rel err %.2e (%s)" % (
        name, prec, sc["dec_ce"], ref["ce"], sc["rec_mse"], ref["mse"], max(v[0] for v in worst.values()),
        max(worst, key=lambda k: worst[k][0])))
    assert not bad, (name, prec, bad)


# ---------------------------------------------------------------------------------------------------------------------
# ALL eight shards of C4 / C5: the tests above hold ONE rank's shard against the oracle; an offset
# or normaliser bug that only shows on another rank would pass them.  Here the 8 per-rank steps run one after the other
# on this GPU (each with its own caption offset, the global normalisers and the global caption index in the dropout
# masks), their gradients are summed — what the SUM all-reduce does — and the sum is held against the oracle's
# gradients of the WHOLE batch of 256 / 512 captions (train.py:17-131 on the full batch, regulariser aside: it is
# applied once after the reduction, and its float32-norm problem is described above).
ALL_SHARDS = {"C4": ("local", 40, 2048, 256), "C5": ("local", 28, 3584, 512)}


@pytest.fixture(scope="module", params=list(ALL_SHARDS))
def full_batch_case(request):
    name = request.param
    kind, F, D, Bg = ALL_SHARDS[name]
    torch.set_num_threads(min(32, torch.get_num_threads() * 4))
    decP = GU.formula_params(GU.decoder_shapes(V, E, H, A, D), 31)
    recP = GU.formula_params(GU.rec_shapes(kind, H, D, RA), 32)
    tg_g = synthetic_targets(Bg, V, seed=77)
    enc_g = synthetic_features(Bg, F, D, seed=79)
    st = O.TrainState(decP, recP, kind)
    drop = O.Dropper("hash", seed=5, B_global=Bg, b_offset=0)
    dl, hid, _, ce, _ = O.forward_decoder(st.dec, enc_g, tg_g, tg_g > 0, drop=drop, lambda_reg=0.0, return_parts=True)
    rl, mse, _ = O.forward_local_reconstructor(st.rec, hid, enc_g, drop=drop, lambda_reg=0.0, return_parts=True)
    (dl + rl).backward()
    ref = dict(ce=float(ce.detach()), mse=float(mse.detach()), dec={k: v.grad.numpy() for k, v in st.dec.items()},
               rec={k: v.grad.numpy() for k, v in st.rec.items()})
    return name, decP, recP, enc_g, tg_g, ref


@pytest.mark.parametrize("prec", ["bf16", "f32"])
def test_all_eight_shards_sum_to_the_full_batch(full_batch_case, prec):
    name, decP, recP, enc_g, tg_g, ref = full_batch_case
    kind, F, D, Bg = ALL_SHARDS[name]
    if prec == "f32" and name == "C5":
        pytest.skip("C5's exact-fp32 shard is covered by test_full_size_step_matches_oracle_elementwise; 8 of them take minutes")
    world = 8
    acc, ce, mse = None, 0.0, 0.0
    dec = rec = None
    for rank in range(world):
        lo, hi = R.shard_bounds(Bg, world, rank)
        if dec is None:
            _, dec, rec = make_models([hi - lo, F, D, V, E, H, A, RA], kind, prec, decP, recP)
        step = R.TrainStep(dec, rec, batch_size=hi - lo, n_frames=F, global_batch=Bg, batch_offset=lo)
        T, w = step.prepare(tg_g.numpy())
        step.fwd_bwd(enc_g[lo:hi].cuda(), tg_g[:, lo:hi].contiguous().cuda(), T, w, seed=5)
        torch.cuda.synchronize()
        assert step.engine.chain_status() == 0
        sc = step.engine.scalar_dict()
        ce += sc["dec_ce"]; mse += sc["rec_mse"]
        g = {grp: {k: v.double().cpu().numpy() for k, v in md["_state"].flat()["grad"].views.items()} for grp, md in (("dec", dec), ("rec", rec))}
        if acc is None:
            acc = g
        else:
            for grp in g:
                for k in g[grp]:
                    acc[grp][k] += g[grp][k]
        del step
    tol = TOL[prec]
    assert abs(ce - ref["ce"]) <= tol["loss"] * abs(ref["ce"]), (name, prec, ce, ref["ce"])
    assert abs(mse - ref["mse"]) <= tol["loss"] * abs(ref["mse"]), (name, prec, mse, ref["mse"])
    worst = {grp + "." + k: (rel_err(acc[grp][k], ref[grp][k]), cosine(acc[grp][k], ref[grp][k])) for grp in acc for k in acc[grp]}
    bad = {k: v for k, v in worst.items() if v[0] > tol["grad"] or v[1] < tol["cos"]}
    print("%s %s all shards: ce %.6f/%.6f mse %.6f/%.6f worst grad rel err %.2e (%s)" % (
        name, prec, ce, ref["ce"], mse, ref["mse"], max(v[0] for v in worst.values()), max(worst, key=lambda k: worst[k][0])))
    assert not bad, (name, prec, bad)


# ---------------------------------------------------------------------------------------------------------------------
# The mode bench.py times by default — GraphedStep with the split reconstructor update (`--defer 2`: d W_hh and its Adam step left
# pending by replay n and run by replay n + 1 beside its decoder forward chain, Adam in the epilogue of that grouped product, both
# operand images of W_hh written there) — at the FULL size of configs[1] / [2], against three oracle iterations
# (train.py:248-273 with torch.optim.Adam).  Learning rates of 1e-3 for both optimisers (config.py:86-87 attributes; the defaults
# 1e-5 / 1e-6 move a weight by less than any usable tolerance): every update moves every weight by ~4 % of its magnitude, so the
# losses of replay 2 and 3 are those of the UPDATED weights and their operand images, and the parameters are compared relative to
# how far they moved (a missing update = 1.0, one of three = 0.33).
FULL_LR = 1e-3
MOVE_TOL = 8e-2


@pytest.mark.parametrize("name", ["C2", "C3", "C2-optimiser-kernel"])
def test_replayed_default_bench_mode_against_the_oracle_at_full_size(name, monkeypatch):
    # C2 / C3 (default): the pending d W_hh product is a grouped launch of gemm_lds.hpp with the Adam update in its epilogue;
    # "C2-optimiser-kernel": the same step with the plain product and the optimiser kernel behind it (RN_ADAM_EPILOGUE=0).
    # Round 6: every product of these steps runs on the hand-written kernels (the vendor library of round 5 is gone).
    if name == "C2-optimiser-kernel":
        monkeypatch.setenv("RN_ADAM_EPILOGUE", "0")
        name = "C2"
    kind, B, F, D, Bg, off = CONFIGS[name]
    torch.set_num_threads(min(32, torch.get_num_threads() * 4))
    decP = GU.formula_params(GU.decoder_shapes(V, E, H, A, D), 31)
    recP = GU.formula_params(GU.rec_shapes(kind, H, D, RA), 32)
    tg = synthetic_targets(B, V, seed=77)
    enc = synthetic_features(B, F, D, seed=78)
    st = O.TrainState(decP, recP, kind, dec_lr=FULL_LR, rec_lr=FULL_LR)
    n, seed0 = 3, 5
    ref = [st.step(enc, tg, tg > 0, O.Dropper("hash", seed=seed0 + it)) for it in range(n)]
    _, dec, rec = make_models([B, F, D, V, E, H, A, RA], kind, "bf16", decP, recP, decoder_learning_rate=FULL_LR,
                              reconstructor_learning_rate=FULL_LR)
    step = R.DataParallelTrainStep(dec, rec, B, 0, 1, n_frames=F)
    step.step_impl.seed_base = seed0 - 1              # device-side rule: dropout seed of optimiser step n (1-based) = base + n
    T, w = step.prepare(tg.numpy())
    run = R.GraphedStep(step, enc.cuda(), tg.cuda(), T, w, warmup=0, defer_reconstructor_update="recurrent")
    eng = step.step_impl.engine
    assert run.deferred and eng.lib.recnet_dim(eng.handle, 10) == 1, "the split update is applied at the benchmark shapes"
    got = [run().clone() for _ in range(n)]
    run.flush()
    torch.cuda.synchronize()
    assert eng.chain_status() == 0
    assert eng.images_stale() == 0       # every operand image = a fresh pack of the updated parameters, word for word
    tol = TOL["bf16"]
    steps_differ = min(abs(ref[i + 1][2] - ref[i][2]) for i in range(n - 1))
    assert steps_differ > 10 * tol["loss"] * abs(ref[0][2]), [r[2] for r in ref]
    for it in range(n):
        sc = got[it].cpu().numpy()
        dl, rl, total, gn = ref[it]
        # (the oracle's regulariser is a float32 CPU norm, 1e-4 off on these tensors: 3e-4 of the value on top of the bf16 bar)
        assert abs(float(sc[2]) - dl) <= (tol["loss"] + 3e-4) * abs(dl), (name, it, float(sc[2]), dl)
        assert abs(float(sc[5]) - rl) <= (tol["loss"] + 3e-4) * abs(rl), (name, it, float(sc[5]), rl)
        assert abs(float(sc[6]) - total) <= (tol["loss"] + 3e-4) * abs(total), (name, it, float(sc[6]), total)
    worst = {}
    for grp, md, P, P0 in (("dec", dec, st.dec, decP), ("rec", rec, st.rec, recP)):
        for k, v in P.items():
            a = md["model"].state_dict()[k].cpu().numpy().astype(np.float64)
            r, i0 = v.detach().numpy().astype(np.float64), P0[k].numpy().astype(np.float64)
            d = np.linalg.norm(r - i0)
            assert d > 0, (grp, k)
            worst[grp + "." + k] = float(np.linalg.norm((a - i0) - (r - i0)) / d)
    print("%s replayed split update, lr %g: losses %s / oracle %s; worst parameter-movement error %.3e (%s)" % (
        name, FULL_LR, [round(float(g[6]), 5) for g in got], [round(r[2], 5) for r in ref], max(worst.values()), max(worst, key=worst.get)))
    bad = {k: v for k, v in worst.items() if v > MOVE_TOL}
    assert not bad, (name, bad)
