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
    """SUM all-reduce of fp32 gradient buffers, optionally carried in bf16 (`dtype="bf16"`): the gradient bucket is as
    expensive as the compute on xGMI (99-344 MB fp32 per step), bf16 halves the bytes on the wire.
    The fp32 buffer is rounded once into a bf16 staging buffer, reduced, and widened back; the rounding (2^-9 relative
    per element) is at the level of the bf16 compute path's own operand rounding.  fp32 is the default and the form the
    parity tests use."""

    def __init__(self, dtype="f32", group=None):
        if dtype not in ("f32", "bf16"):
            raise NotImplementedError("gradient transport dtype %r" % dtype)
        self.dtype, self.group, self._stage = dtype, group, {}

    def start(self, buf):
        """Launch the collective on `buf` (a contiguous fp32 tensor or slice); returns a token for finish()."""
        import torch.distributed as dist
        if self.dtype == "f32":
            return (dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), buf, None)
        key = (buf.data_ptr(), buf.numel())
        st = self._stage.get(key)
        if st is None:
            st = self._stage[key] = torch.empty(buf.numel(), dtype=torch.bfloat16, device=buf.device)
        st.copy_(buf)
        return (dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group, async_op=True), buf, st)

    @staticmethod
    def finish(token):
        work, buf, st = token
        work.wait()
        if st is not None:
            buf.copy_(st)


def allreduce_sum_(flat_buffers, group=None, dtype="f32"):
    """One collective per flat gradient buffer (reconstructor first: it is ready first)."""
    tr = GradTransport(dtype, group)
    for tok in [tr.start(t) for t in flat_buffers]:
        tr.finish(tok)


class DataParallelTrainStep:
    """Wraps api.TrainStep for world_size ranks.  Each rank owns captions [lo, hi) of the global batch."""

    def __init__(self, decoder, reconstructor, global_batch, rank, world_size, n_frames=None, group=None,
                 always_reduce=False, grad_dtype="f32"):
        import os
        from .api import TrainStep
        if world_size > 1:
            # RCCL's kernel of the reconstructor bucket is resident while the decoder's BPTT chain kernel runs: keep
            # that many CUs out of the chain kernel's residency check (csrc/api.hip: RN_RESERVE_CUS)
            os.environ.setdefault("RN_RESERVE_CUS", "64")
        self.transport = GradTransport(grad_dtype, group)
        self.rank, self.world = rank, world_size
        # reduce even with one rank (exercises the collective path under torchrun --nproc-per-node 1)
        self.reduce = world_size > 1 or always_reduce
        self.global_batch = global_batch
        self.lo, self.hi = shard_bounds(global_batch, world_size, rank)
        self.group = group
        self.step_impl = TrainStep(decoder, reconstructor, batch_size=self.hi - self.lo, n_frames=n_frames,
                                   global_batch=global_batch, batch_offset=self.lo)
        self.decoder, self.reconstructor = decoder, reconstructor

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
