"""Data parallelism over the GPUs of one node: one process per GPU, captions sharded across ranks,
gradients SUM-all-reduced with RCCL (torch.distributed backend "nccl") over xGMI.

Exactness: the loss normalisers — per-step unmasked counts n_t, their sum N, the
loop length T and the MSE element count — are GLOBAL-batch quantities.  Every rank derives them from
the full [31, B_global] target matrix (it is tiny), weights its local CE terms by 1/(n_t N) and its
squared error by 1/count, and the gradients are then summed, not averaged.  The norm regulariser,
weight decay, clipping and Adam run identically on every rank after the reduction.
"""
import torch


def shard_bounds(global_batch, world_size, rank):
    """Caption range [lo, hi) of `rank`: sizes differ by at most one, larger shards first
    (B=100, G=8 -> 13,13,13,13,12,12,12,12)."""
    base, rem = divmod(global_batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradTransport:
    """SUM all-reduce of fp32 gradient buffers.

    algo "ring" (default for fp32): one `dist.all_reduce` per buffer — RCCL picks the algorithm.
    algo "direct" ("direct reduce-scatter + all-gather using all 7 links concurrently"): the
    buffer is cut into world_size chunks; every rank sends chunk r to rank r (`all_to_all_single`: one message per peer,
    i.e. per xGMI link), sums the world_size copies of its own chunk IN FP32 in rank order, and the reduced chunks are
    all-gathered.  Bytes per rank and direction: 2 (W-1)/W S like the ring, but spread over all peers at once instead
    of over one ring neighbour.

    dtype "bf16" halves the bytes on the wire (99-344 MB fp32 per step) and always uses the
    direct form, so that the accumulation happens in fp32 at the destination: each rank's contribution is rounded to
    bf16 once (2^-9 relative per element, the level of the bf16 compute path's own operand rounding), the W
    contributions are summed in fp32, the sum is rounded to bf16 once for the all-gather.  (A plain `all_reduce` of a
    bf16 buffer would accumulate in bf16 inside RCCL: W - 1 roundings of the running sum.)  Every rank receives the
    same bytes, so the replicas stay bit-identical.  gloo (CPU tests) has no all_to_all: there the chunks travel by
    `all_gather` of the staged buffers and the same fp32 rank-order sum is formed locally — same result bit for bit.
    fp32 ring is the default and the form the parity tests use."""

    def __init__(self, dtype="f32", group=None, algo=None):
        if dtype not in ("f32", "bf16"):
            raise NotImplementedError("gradient transport dtype %r" % dtype)
        algo = algo or ("direct" if dtype == "bf16" else "ring")
        if algo not in ("ring", "direct"):
            raise NotImplementedError("gradient transport algorithm %r" % algo)
        if dtype == "bf16" and algo != "direct":
            raise NotImplementedError("bf16 transport accumulates in fp32 at the destination: algo='direct'")
        self.dtype, self.group, self.algo, self._stage, self._side = dtype, group, algo, {}, None

    # ------------------------------------------------------------------ direct reduce-scatter + all-gather
    def _buffers(self, buf, W):
        key = (buf.data_ptr(), buf.numel(), W)
        st = self._stage.get(key)
        if st is None:
            wire = torch.bfloat16 if self.dtype == "bf16" else torch.float32
            chunk = (buf.numel() + W - 1) // W
            chunk = (chunk + 7) // 8 * 8                      # 16-byte multiples on the wire
            st = self._stage[key] = dict(chunk=chunk, send=torch.zeros(W * chunk, dtype=wire, device=buf.device),
                                         recv=torch.empty(W * chunk, dtype=wire, device=buf.device),
                                         red=torch.empty(chunk, dtype=wire, device=buf.device),
                                         out=torch.empty(W * chunk, dtype=wire, device=buf.device))
        return st

    def _start_direct(self, buf):
        import torch.distributed as dist
        W = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        st = self._buffers(buf, W)
        n, chunk = buf.numel(), st["chunk"]
        cuda = buf.is_cuda
        if cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=buf.device)
            self._side.wait_stream(torch.cuda.current_stream(buf.device))
            ctx = torch.cuda.stream(self._side)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        # RN_DP_STAGING=torch selects the torch form below on every backend (the HIP staging kernels have only ever
        # run at world size 1 on hardware; both forms do the same arithmetic — tests/test_gpu_dp.py holds them bit for bit)
        import os as _os
        hip_staging = _os.environ.get("RN_DP_STAGING", "hip") != "torch"
        if hip_staging and cuda and W <= 16 and dist.get_backend(self.group) != "gloo":
            # two HIP kernels around the collectives (csrc/kernels_util.hpp: dp_cast_kernel, dp_reduce_kernel) instead of W + 3 torch
            # launches: cast + stage, W-way fp32 sum in rank order + one rounding; the same arithmetic as the torch form below
            import ctypes as C
            from . import _lib
            lib = _lib.load()
            bf = 1 if self.dtype == "bf16" else 0
            with ctx:
                s_ = C.c_void_p(torch.cuda.current_stream(buf.device).cuda_stream)
                _lib.check(lib.recnet_dp_cast(C.c_void_p(buf.data_ptr()), 0, C.c_void_p(st["send"].data_ptr()), bf, n, s_), "recnet_dp_cast")
                dist.all_to_all_single(st["recv"], st["send"], group=self.group)      # chunk r of every rank -> rank r
                _lib.check(lib.recnet_dp_reduce(C.c_void_p(st["recv"].data_ptr()), W, chunk, C.c_void_p(st["red"].data_ptr()), bf, s_), "recnet_dp_reduce")
                work = dist.all_gather_into_tensor(st["out"], st["red"], group=self.group, async_op=True)
            return (work, buf, st)
        with ctx:
            st["send"][:n].copy_(buf)                          # one rounding per rank (bf16 wire) / plain copy (fp32 wire)
            if dist.get_backend(self.group) == "gloo":
                parts = [torch.empty_like(st["send"]) for _ in range(W)]
                dist.all_gather(parts, st["send"], group=self.group)
                mine = [p[rank * chunk:(rank + 1) * chunk] for p in parts]
            else:
                dist.all_to_all_single(st["recv"], st["send"], group=self.group)      # chunk r of every rank -> rank r
                mine = [st["recv"][r * chunk:(r + 1) * chunk] for r in range(W)]
            acc = mine[0].float()
            for r in range(1, W):                              # fp32 accumulation at the destination, rank order
                acc += mine[r]
            st["red"].copy_(acc)                               # one rounding of the sum (bf16 wire)
            work = dist.all_gather_into_tensor(st["out"], st["red"], group=self.group, async_op=True)
        return (work, buf, st)

    def start(self, buf):
        """Launch the collective on `buf` (a contiguous fp32 tensor or slice); returns a token for finish()."""
        import torch.distributed as dist
        if self.algo == "ring":
            return (dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), buf, None)
        return self._start_direct(buf)

    def finish(self, token):
        work, buf, st = token
        work.wait()
        if st is not None:
            if buf.is_cuda and self._side is not None:
                with torch.cuda.stream(self._side):
                    work.wait()
                torch.cuda.current_stream(buf.device).wait_stream(self._side)
            if buf.is_cuda:
                import ctypes as C
                from . import _lib
                lib = _lib.load()
                _lib.check(lib.recnet_dp_cast(C.c_void_p(st["out"].data_ptr()), 1 if self.dtype == "bf16" else 0, C.c_void_p(buf.data_ptr()), 0,
                                              buf.numel(), C.c_void_p(torch.cuda.current_stream(buf.device).cuda_stream)), "recnet_dp_cast")
                return
            buf.copy_(st["out"][:buf.numel()])


def allreduce_sum_(flat_buffers, group=None, dtype="f32", algo=None):
    """One collective per flat gradient buffer (reconstructor first: it is ready first)."""
    tr = GradTransport(dtype, group, algo)
    for tok in [tr.start(t) for t in flat_buffers]:
        tr.finish(tok)


class DataParallelTrainStep:
    """Wraps api.TrainStep for world_size ranks.  Each rank owns captions [lo, hi) of the global batch."""

    def __init__(self, decoder, reconstructor, global_batch, rank, world_size, n_frames=None, group=None,
                 always_reduce=False, grad_dtype="f32", grad_algo=None):
        import os
        from .api import TrainStep
        if world_size > 1:
            # RCCL's kernel of the reconstructor bucket is resident while the decoder's BPTT chain kernel runs: keep
            # that many CUs out of the chain kernel's residency check (csrc/api.hip: RN_RESERVE_CUS)
            os.environ.setdefault("RN_RESERVE_CUS", "64")
        self.transport = GradTransport(grad_dtype, group, grad_algo)
        self.rank, self.world = rank, world_size
        # reduce even with one rank (exercises the collective path under torchrun --nproc-per-node 1)
        self.reduce = world_size > 1 or always_reduce
        self.global_batch = global_batch
        self.lo, self.hi = shard_bounds(global_batch, world_size, rank)
        self.group = group
        self.step_impl = TrainStep(decoder, reconstructor, batch_size=self.hi - self.lo, n_frames=n_frames,
                                   global_batch=global_batch, batch_offset=self.lo)
        self.decoder, self.reconstructor = decoder, reconstructor
        if world_size > 1:
            # train.py:38 draws `random.random() <= ratio` per iteration; ranks whose generators differ would silently mix a
            # teacher-forced shard with free-running ones and all-reduce the result: every rank takes rank 0's draw
            self.step_impl.sync_draw = self._rank0_draw

    def _rank0_draw(self, tf):
        import torch.distributed as dist
        box = [bool(tf)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        return box[0]

    @property
    def scalars(self):
        return self.step_impl.scalars

    def prepare(self, global_targets_host):
        return self.step_impl.prepare(global_targets_host)

    def decoder_out_offset(self):
        """Start of the out.weight / out.bias gradients in the decoder's flat buffer (they are its tail: parameters are
        laid out in registration order, decoder.py:22-42).  They are complete before the decoder's BPTT starts."""
        return self.decoder["_state"].flat()["grad"].offsets["out.weight"]

    def early_buffers(self):
        """Gradients that are complete when the reconstructor's backward is: the reconstructor bucket and the decoder's
        output layer (all-reduced while the decoder's BPTT runs)."""
        bufs = []
        if self.reconstructor:
            bufs.append(self.reconstructor["_state"].flat()["grad"].flat)
        bufs.append(self.decoder["_state"].flat()["grad"].flat[self.decoder_out_offset():])
        return bufs

    def late_buffers(self):
        return [self.decoder["_state"].flat()["grad"].flat[:self.decoder_out_offset()]]

    def grad_buffers(self):
        return self.early_buffers() + self.late_buffers()

    def __call__(self, enc_local, targets_local, T, step_weight, seed=None):
        self.step_impl.fwd_bwd(enc_local, targets_local, T, step_weight, seed)
        if self.reduce:
            for tok in [self.transport.start(b) for b in self.grad_buffers()]:
                self.transport.finish(tok)
        self.step_impl.optimizer_step()
        return self.step_impl.scalars

    def reduce_scalars(self):
        """Global loss values (the CE / MSE parts are sums of per-rank partial sums; the regulariser
        terms are replicated)."""
        import torch.distributed as dist
        s = self.step_impl.scalars.clone()
        if self.world > 1:
            part = torch.stack([s[0], s[3]])
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
            s[0], s[3] = part[0], part[1]
        return s
