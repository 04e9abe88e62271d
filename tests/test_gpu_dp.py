"""Data-parallel GPU path end to end on ONE GPU: two processes share cuda:0 and exchange gradients over
gloo (RCCL refuses two ranks on one device; the collective call site is the same torch.distributed
all_reduce).  Two ranks x B/2 captions must reproduce one rank x B captions: same losses, same parameters
after the optimiser steps (exact-fp32 MFMA path, so only summation order differs)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DIMS = [8, 5, 64, 61, 16, 32, 16, 16]
LENS = [5, 2, 7, 3, 1, 4, 9, 6]


def _setup(kind, B, lo, hi, prec="f32"):
    import recnet_amd as R
    from tests import golden_util as GU
    from tests.gpu_util import make_models
    Bfull, F, D, V, E, H, A, RA = DIMS
    decP = GU.formula_params(GU.decoder_shapes(V, E, H, A, D), 3)
    recP = GU.formula_params(GU.rec_shapes(kind, H, D, RA), 4) if kind else None
    enc, targets = GU.make_batch(Bfull, F, D, V, LENS, 9)
    dims = [hi - lo] + DIMS[1:]
    C, dec, rec = make_models(dims, kind, prec, decP, recP, device="cuda:0")
    return R, dec, rec, enc, targets


def _worker(rank, world, port, kind, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # Two ranks share ONE GPU here (test rig only).  The persistent chain kernels need every workgroup of a launch resident
    # and spin on grid barriers: two such launches from two processes could each hold part of the CUs and wait for the
    # rest forever.  One process per GPU (the supported deployment) cannot get there; this rig switches them off.
    os.environ["RN_PER_STEP"] = "rec,dec,loc"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import recnet_amd as R
    lo, hi = R.shard_bounds(DIMS[0], world, rank)
    R_, dec, rec, enc, targets = _setup(kind, DIMS[0], lo, hi)
    step = R.DataParallelTrainStep(dec, rec, DIMS[0], rank, world, n_frames=DIMS[1])
    T, w = step.prepare(targets.numpy())
    # the early buckets (reconstructor + the decoder's output layer) and the late one partition the gradients exactly
    n_early = sum(b.numel() for b in step.early_buffers())
    n_late = sum(b.numel() for b in step.late_buffers())
    n_all = dec["_state"].flat()["grad"].flat.numel() + (rec["_state"].flat()["grad"].flat.numel() if rec else 0)
    assert n_early + n_late == n_all and step.late_buffers()[0].data_ptr() == dec["_state"].flat()["grad"].flat.data_ptr()
    assert step.early_buffers()[-1].numel() >= DIMS[3] * DIMS[5] + DIMS[3]          # out.weight [V,H] + out.bias [V]
    e, t = enc[lo:hi].cuda(), targets[:, lo:hi].contiguous().cuda()
    run = R.GraphedStep(step, e, t, T, w, warmup=0)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    sc = step.reduce_scalars().cpu().numpy()
    if rank == 0:
        q.put((sc, {k: v.detach().cpu().numpy() for k, v in dec["model"].state_dict().items()},
               {k: v.detach().cpu().numpy() for k, v in rec["model"].state_dict().items()} if rec else None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["global", "local"])
def test_two_ranks_on_one_gpu_match_single_rank(kind):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    sc2, dec2, rec2 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single rank, whole batch, same two steps (eager TrainStep: also cross-checks graph replay vs eager launch)
    R, dec, rec, enc, targets = _setup(kind, DIMS[0], 0, DIMS[0])
    step = R.TrainStep(dec, rec)
    T, w = step.prepare(targets.numpy())
    for _ in range(2):
        sc1 = step(enc.cuda(), targets.cuda(), T, w)
    torch.cuda.synchronize()
    sc1 = sc1.cpu().numpy()
    for i in (0, 3, 6):      # dec_ce, rec_mse, total
        if i == 6:
            continue         # the per-rank total mixes local partial sums; compare the parts
        assert abs(sc2[i] - sc1[i]) <= 2e-5 * max(abs(sc1[i]), 1e-3), (i, sc2[i], sc1[i])
    for k, v in dec["model"].state_dict().items():
        assert np.abs(dec2[k] - v.cpu().numpy()).max() <= 2e-6, k
    for k, v in rec["model"].state_dict().items():
        assert np.abs(rec2[k] - v.cpu().numpy()).max() <= 2e-6, k


def _worker_giveup(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RN_PER_STEP"] = "rec,dec,loc"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import recnet_amd as R
    lo, hi = R.shard_bounds(DIMS[0], world, rank)
    R_, dec, rec, enc, targets = _setup("global", DIMS[0], lo, hi)
    step = R.DataParallelTrainStep(dec, rec, DIMS[0], rank, world, n_frames=DIMS[1])
    T, w = step.prepare(targets.numpy())
    e, t = enc[lo:hi].cuda(), targets[:, lo:hi].contiguous().cuda()
    run = R.GraphedStep(step, e, t, T, w, warmup=0)
    run()
    torch.cuda.synchronize()
    eng = step.step_impl.engine
    snap = lambda: {k: v.detach().clone() for m in (dec, rec) for k, v in m["model"].state_dict().items()}
    before = snap()
    if rank == 1:
        eng.debug_raise_give_up(1)            # what rec_chain_kernel does when a bounded wait runs out, on ONE rank
    sc = run().clone()
    torch.cuda.synchronize()
    after = snap()
    unchanged = all(torch.equal(before[k], after[k]) for k in before)
    status = eng.chain_status()
    tr = R.Trainer.__new__(R.Trainer)
    tr.dp, tr.world, tr.rank, tr.decoder, tr._graphs, tr.iteration = step, world, rank, dec, {}, 2
    raised = ""
    try:
        tr.check_health(sc)
    except RuntimeError as ex:
        raised = str(ex)
    run2 = R.GraphedStep(step, e, t, T, w, warmup=0)       # graphs re-captured after the reset, both ranks keep training
    run2()
    torch.cuda.synchronize()
    moved = any(not torch.equal(after[k], v) for k, v in snap().items())
    q.put((rank, unchanged, status, raised, bool(torch.isfinite(sc[6])), moved, eng.chain_status()))
    dist.barrier()
    dist.destroy_process_group()


def test_give_up_on_one_rank_skips_the_update_on_every_rank():
    """chain give-up handling was rank-local — the healthy rank applied the garbage bucket and blocked in the
    next collective while the affected one raised.  Now the poison travels with the last gradient bucket and
    check_health is collective."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_giveup, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, unchanged, status, raised, finite, moved, status_after in res:
        assert unchanged, "rank %d applied an update from a poisoned step" % rank
        assert status & 256, (rank, status)
        assert not finite
        assert "gave up" in raised, (rank, raised)
        assert moved and status_after == 0, (rank, moved, status_after)
    assert res[0][2] & 0xFF == 0 and "another rank" in res[0][3]       # rank 0's own chains were healthy
    assert res[1][2] & 1


@pytest.mark.parametrize("wire", ["f32", "bf16"])
@pytest.mark.parametrize("world", [1, 2, 8, 16])
def test_direct_transport_staging_kernels_equal_the_torch_form(world, wire):
    """csrc/kernels_util.hpp dp_cast_kernel / dp_reduce_kernel (recnet_dp_cast / recnet_dp_reduce: the staging around the two
    collectives of the direct gradient transport) against the torch ops they replace in dp.GradTransport:
    fp32 -> wire type, W-way fp32 accumulation IN RANK ORDER with one rounding of the sum, wire type -> fp32.  Bit for bit,
    at a length that is a multiple of nothing."""
    import ctypes as C
    from recnet_amd import _lib
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    wt = torch.bfloat16 if wire == "bf16" else torch.float32
    bf = 1 if wire == "bf16" else 0
    g = torch.Generator().manual_seed(5 + world)
    n = 100003
    src = (torch.randn(n, generator=g) * 3).cuda()
    send = torch.zeros(n + 5, dtype=wt, device="cuda")
    _lib.check(lib.recnet_dp_cast(C.c_void_p(src.data_ptr()), 0, C.c_void_p(send.data_ptr()), bf, n, st), "recnet_dp_cast")
    assert torch.equal(send[:n], src.to(wt)) and float(send[n:].float().abs().max()) == 0.0
    chunk = 12504
    recv = (torch.randn(world * chunk, generator=g) * 2).cuda().to(wt)
    red = torch.empty(chunk, dtype=wt, device="cuda")
    _lib.check(lib.recnet_dp_reduce(C.c_void_p(recv.data_ptr()), world, chunk, C.c_void_p(red.data_ptr()), bf, st), "recnet_dp_reduce")
    acc = recv[:chunk].float()
    for r in range(1, world):
        acc = acc + recv[r * chunk:(r + 1) * chunk].float()
    assert torch.equal(red, acc.to(wt))
    back = torch.empty(n, device="cuda")
    _lib.check(lib.recnet_dp_cast(C.c_void_p(send.data_ptr()), bf, C.c_void_p(back.data_ptr()), 0, n, st), "recnet_dp_cast")
    torch.cuda.synchronize()
    assert torch.equal(back, send[:n].float())
