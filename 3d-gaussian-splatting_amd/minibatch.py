"""View-minibatch data parallelism: the views of one optimisation step are independent renders, so
they are split over the ranks; every rank accumulates the parameter gradients of its views in one
flat fp32 bucket and the buckets are summed with a single all-reduce (RCCL over xGMI on the GPUs,
gloo in the CPU tests).  The reference renders one view per step on one device (``Trainer.py:54-57``);
this module is what carries that step to N GPUs.
"""
from typing import List, Sequence, Tuple

import torch

# floats per Gaussian in the bucket: coords 3, scale 3, rotation 4, opacity 1, sh 3*16
FIELD_SIZES = (3, 3, 4, 1, 48)
FLOATS_PER_GAUSSIAN = sum(FIELD_SIZES)


def shard_views(n_views: int, world_size: int, rank: int) -> List[int]:
    """Contiguous block of views for ``rank``; the first ``n_views % world_size`` ranks get one more."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside 0..{world_size - 1}")
    base, extra = divmod(n_views, world_size)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


class GradBucket:
    """One contiguous fp32 buffer holding the five dense parameter gradients of N Gaussians."""

    def __init__(self, n: int, device):
        self.n = n
        self.flat = torch.zeros(n * FLOATS_PER_GAUSSIAN, dtype=torch.float32, device=device)
        self.views: Tuple[torch.Tensor, ...] = self._make_views()

    def _make_views(self):
        out, off = [], 0
        for size in FIELD_SIZES:
            chunk = self.flat[off:off + self.n * size]
            out.append(chunk.view(self.n, 3, 16) if size == 48 else chunk.view(self.n, size))
            off += self.n * size
        return tuple(out)

    @property
    def coords(self): return self.views[0]

    @property
    def scale(self): return self.views[1]

    @property
    def rotation(self): return self.views[2]

    @property
    def opacity(self): return self.views[3]

    @property
    def sh(self): return self.views[4]

    def add_(self, grads: Sequence[torch.Tensor]):
        """Accumulate five gradient tensors (host-side path; the HIP backward adds in place)."""
        for dst, g in zip(self.views, grads):
            dst.add_(g.reshape(dst.shape))

    def all_reduce(self, group=None):
        """Sum the bucket over the ranks: one collective for all five tensors."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        return self
