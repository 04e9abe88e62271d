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


def allreduce_sum_(flat_buffers, group=None):
    """One collective per flat gradient buffer (reconstructor first: it is ready first)."""
    import torch.distributed as dist
    works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True) for t in flat_buffers]
    for w in works:
        w.wait()


class DataParallelTrainStep:
    """Wraps api.TrainStep for world_size ranks.  Each rank owns captions [lo, hi) of the global batch."""

    def __init__(self, decoder, reconstructor, global_batch, rank, world_size, n_frames=None, group=None,
                 always_reduce=False):
        from .api import TrainStep
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

    def grad_buffers(self):
        bufs = []
        if self.reconstructor:
            bufs.append(self.reconstructor["_state"].flat()["grad"].flat)
        bufs.append(self.decoder["_state"].flat()["grad"].flat)
        return bufs

    def __call__(self, enc_local, targets_local, T, step_weight, seed=None):
        self.step_impl.fwd_bwd(enc_local, targets_local, T, step_weight, seed)
        if self.reduce:
            allreduce_sum_(self.grad_buffers(), self.group)
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
