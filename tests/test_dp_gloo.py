"""Data-parallel exactness on CPU (gloo, world_size 2): sharding the captions over ranks with the
product's shard_bounds / step_weights / all-reduce plumbing (dp.py) reproduces the full-batch gradients
when every rank uses the GLOBAL loss normalisers and gradients are SUMmed.  There is no
HIP here, so the per-rank compute engine is the CPU oracle — it only stands in for the kernels; what is
under test is the N>1 host logic the GPU path shares."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import recnet_amd as R
from oracle import recnet_oracle as O
from tests import golden_util as GU

DIMS = dict(B=7, F=5, D=24, V=41, E=10, H=16, A=8, RA=6)
LENS = [9, 2, 5, 12, 1, 7, 4]
SEED = 5


def _problem(kind, B=None):
    d = DIMS
    decP = GU.formula_params(GU.decoder_shapes(d["V"], d["E"], d["H"], d["A"], d["D"]), 21)
    recP = GU.formula_params(GU.rec_shapes(kind, d["H"], d["D"], d["RA"]), 22) if kind else None
    lens = LENS if B is None else [int(x) for x in np.random.RandomState(3).randint(1, 13, size=B)]
    enc, targets = GU.make_batch(B or d["B"], d["F"], d["D"], d["V"], lens, 33)
    return decP, recP, enc, targets


def _shard_grads(kind, decP, recP, enc, targets, lo, hi):
    """One rank's fwd+bwd on captions [lo, hi) with global normalisers (no regulariser: it is added once,
    after the reduction)."""
    d = dict(DIMS, B=targets.shape[1])                   # `targets` always covers the GLOBAL batch
    masks_g = (targets > 0).numpy()
    T = R.decode_len(masks_g)
    w = R.step_weights(masks_g, T)                       # 1 / (n_t * N), global counts
    n_t = masks_g[:T].sum(1)
    N = int(n_t.sum())
    np.testing.assert_allclose(w, 1.0 / (n_t * N), rtol=1e-6)
    dec = {k: v.clone().requires_grad_(True) for k, v in decP.items()}
    rec = None if recP is None else {k: v.clone().requires_grad_(True) for k, v in recP.items()}
    drop = O.Dropper("hash", seed=SEED, B_global=d["B"], b_offset=lo)
    e, t = enc[lo:hi], targets[:, lo:hi]
    dl, hid, _, ce, _ = O.forward_decoder(dec, e, t, t > 0, lambda_reg=0.0, drop=drop,
                                          global_counts=([int(x) for x in n_t], N), return_parts=True)
    loss = ce
    if kind == "global":
        _, mse, _ = O.forward_global_reconstructor(rec, hid, e, lambda_reg=0.0, drop=drop,
                                                   mse_count=d["B"] * d["D"], return_parts=True)
        loss = loss + mse
    elif kind == "local":
        _, mse, _ = O.forward_local_reconstructor(rec, hid, e, lambda_reg=0.0, drop=drop,
                                                  mse_count=d["B"] * d["F"] * d["D"], return_parts=True)
        loss = loss + mse
    loss.backward()
    flat = [torch.cat([dec[k].grad.reshape(-1) for k in O.decoder_param_order(dec)])]
    if rec is not None:
        flat.append(torch.cat([rec[k].grad.reshape(-1) for k in O.rec_param_order(rec)]))
    return flat


def _worker(rank, world, port, kind, out, B=None, dtype="f32", algo=None, out2=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    decP, recP, enc, targets = _problem(kind, B)
    lo, hi = R.shard_bounds(B or DIMS["B"], world, rank)
    flat = _shard_grads(kind, decP, recP, enc, targets, lo, hi)
    from recnet_amd.dp import allreduce_sum_
    mine = [f.clone() for f in flat]
    allreduce_sum_(list(reversed(flat)), dtype=dtype, algo=algo)    # reconstructor bucket first, as on the GPU path
    if rank == 0:
        out.put([f.numpy() for f in flat])
    if out2 is not None:
        out2.put((rank, [f.numpy() for f in mine], [f.numpy() for f in flat]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("kind", [None, "global", "local"])
def test_two_rank_gradients_equal_full_batch(kind):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    decP, recP, enc, targets = _problem(kind)
    ref = _shard_grads(kind, decP, recP, enc, targets, 0, DIMS["B"])      # the whole batch on one "rank"
    for a, b in zip(got, ref):
        b = b.numpy()
        assert np.linalg.norm(a - b) <= 2e-6 * np.linalg.norm(b)
    # and the whole-batch, global-normaliser formulation IS the reference loss (regulariser aside)
    st = O.TrainState(decP, recP, kind, dec_lambda_reg=0.0, rec_lambda_reg=0.0)
    dl, rl, _, _, _ = st.losses(enc, targets, targets > 0, O.Dropper("hash", seed=SEED))
    ((dl if rl is None else dl + rl)).backward()
    full = torch.cat([st.dec[k].grad.reshape(-1) for k in O.decoder_param_order(st.dec)]).numpy()
    assert np.linalg.norm(ref[0].numpy() - full) <= 2e-6 * np.linalg.norm(full)


def _run(world, kind, B=None, dtype="f32", algo=None, per_rank=None):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    out2 = ctx.Queue() if per_rank is not None else None
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, out, B, dtype, algo, out2)) for r in range(world)]
    for p in procs:
        p.start()
    got = out.get(timeout=300)
    if per_rank is not None:
        per_rank.extend(sorted((out2.get(timeout=300) for _ in range(world)), key=lambda x: x[0]))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    decP, recP, enc, targets = _problem(kind, B)
    ref = _shard_grads(kind, decP, recP, enc, targets, 0, B or DIMS["B"])
    return got, [r.numpy() for r in ref]


def test_eight_uneven_ranks_of_one_hundred_captions():
    """BASELINE's headline batch over a full node: 100 captions over 8 ranks = 13,13,13,13,12,12,12,12."""
    assert [R.shard_bounds(100, 8, r) for r in range(8)] == [(0, 13), (13, 26), (26, 39), (39, 52), (52, 64), (64, 76), (76, 88), (88, 100)]
    got, ref = _run(8, "local", B=100)
    for a, b in zip(got, ref):
        assert np.linalg.norm(a - b) <= 3e-6 * np.linalg.norm(b)


def test_bf16_gradient_transport_stays_inside_the_bf16_parity_bar():
    got, ref = _run(2, "global", dtype="bf16")
    for a, b in zip(got, ref):
        assert 1e-5 < np.linalg.norm(a - b) / np.linalg.norm(b) <= 6e-3        # rounded once per rank, 2^-9 per element


def test_bf16_transport_at_world_size_8_accumulates_in_fp32_at_the_destination():
    """a bf16 `all_reduce` accumulates in bf16 inside the collective (7 roundings of the running sum at
    8 ranks).  The direct transport (dp.GradTransport, algo "direct") rounds each rank's contribution once, sums the 8
    contributions in fp32 and rounds the sum once — checked here against exactly that arithmetic, redone in numpy from
    the per-rank gradients, bit for bit; and every rank ends with the same bytes."""
    per_rank = []
    got, ref = _run(8, "local", B=100, dtype="bf16", per_rank=per_rank)
    to_bf16 = lambda a: torch.from_numpy(a).to(torch.bfloat16)
    for i in range(len(got)):
        acc = to_bf16(per_rank[0][1][i]).float()
        for r in range(1, 8):
            acc += to_bf16(per_rank[r][1][i]).float()
        want = acc.to(torch.bfloat16).float().numpy()
        for r in range(8):
            assert np.array_equal(per_rank[r][2][i], want), (i, r)          # bit-identical replicas, fp32 accumulation
        rel = np.linalg.norm(got[i] - ref[i]) / np.linalg.norm(ref[i])
        assert 1e-5 < rel <= 6e-3, rel
        # what an in-collective bf16 accumulation would have produced is measurably worse
        run = to_bf16(per_rank[0][1][i])
        for r in range(1, 8):
            run = (run.float() + to_bf16(per_rank[r][1][i]).float()).to(torch.bfloat16)
        rel_ring = np.linalg.norm(run.float().numpy() - ref[i]) / np.linalg.norm(ref[i])
        assert rel < rel_ring, (rel, rel_ring)


def test_direct_fp32_transport_equals_the_full_batch_at_world_size_8():
    got, ref = _run(8, "global", B=100, dtype="f32", algo="direct")
    for a, b in zip(got, ref):
        assert np.linalg.norm(a - b) <= 3e-6 * np.linalg.norm(b)
