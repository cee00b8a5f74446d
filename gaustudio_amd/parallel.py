"""Multi-GPU data parallelism for the rasterizer: one camera per GPU, replicated Gaussians, ONE
all-reduce of the per-Gaussian gradients after backward (BASELINE.json north_star).

The reference has no distributed code at all ("Collective / NCCL call sites: none");
this is new design for an 8 x MI355X node: one process per GPU, torch.distributed backend "nccl"
(= RCCL over xGMI on ROCm).  Views are independent given replicated parameters, so forward and
backward need no exchange; the only collective is a SUM over ranks of a single flat fp32 buffer
[means3D 3 | sh 3M | opacity 1 | scales 3 | rotations 4] = 59 floats (236 B) per Gaussian at M = 16,
which equals accumulating the N views on one GPU.  One flat buffer -> one large collective: xGMI is
point-to-point (7 links x ~153 GB/s per GPU), so a few large transfers beat many small ones.

The same code runs on CPU tensors with the "gloo" backend (tests/test_distributed.py, world_size 2).
"""
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


ARENA_ROLES = ("means3D", "shs", "opacities", "scales", "rotations")


class FlatGradBucket:
    """A persistent flat fp32 buffer holding the gradients of `params` back to back.

    `roles` (optional) names which parameter plays which operator input (keys of ARENA_ROLES).  With roles,
    `arm()` makes the NEXT rasterizer backward write its gradients straight into this buffer, so that autograd
    adopts slices of it as `p.grad` and the all-reduce needs no pack copy (236 MB read + written per step at 1 M
    Gaussians otherwise).  Anything that does not end up aliased (another op in the graph, autograd deciding to
    copy) is still handled by pack()."""

    def __init__(self, params: Sequence[torch.Tensor], roles: Optional[dict] = None):
        self.params: List[torch.Tensor] = list(params)
        self.roles = None
        if roles is not None:
            if set(roles) != set(ARENA_ROLES):
                raise ValueError(f"roles must name exactly {ARENA_ROLES}")
            idx = {id(p): i for i, p in enumerate(self.params)}
            if any(id(roles[r]) not in idx for r in ARENA_ROLES):
                raise ValueError("every role parameter must be one of the bucket's parameters")
            self.roles = [idx[id(roles[r])] for r in ARENA_ROLES]
        if not self.params:
            raise ValueError("FlatGradBucket needs at least one parameter")
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("all parameters must be float32 on one device")
        self.sizes = [p.numel() for p in self.params]
        self.offsets = [0]
        for n in self.sizes:
            self.offsets.append(self.offsets[-1] + n)
        self.flat = torch.zeros(self.offsets[-1], dtype=torch.float32, device=dev)

    def views(self):
        return [self.flat[o:o + n].view_as(p) for o, n, p in zip(self.offsets, self.sizes, self.params)]

    def arm(self):
        """Directs the next rasterizer backward into this buffer.  Only when every p.grad is None (an existing
        p.grad could BE a slice of this buffer from the previous step: writing the new gradient over it and then
        accumulating would be wrong), on a ROCm device, and with roles given; otherwise a no-op (pack() copies)."""
        if self.roles is None or not self.flat.is_cuda or any(p.grad is not None for p in self.params):
            return False
        from . import _C
        views = self.views()
        _C.set_grad_arena([views[i] for i in self.roles])
        return True

    def disarm(self):
        if self.roles is not None and self.flat.is_cuda:
            from . import _C
            _C.set_grad_arena([])

    def pack(self):
        """Brings every p.grad into the flat buffer: zeros where a parameter received none, nothing to do where
        p.grad already IS the slice (arm()), a copy otherwise."""
        for v, p in zip(self.views(), self.params):
            if p.grad is None:
                v.zero_()
            elif not (p.grad.data_ptr() == v.data_ptr() and p.grad.shape == v.shape and p.grad.is_contiguous()):
                v.copy_(p.grad)
        return self.flat

    def unpack(self):
        """Points every p.grad at its (reduced) slice of the flat buffer."""
        for v, p in zip(self.views(), self.params):
            p.grad = v
        return self.params

    @property
    def nbytes(self):
        return self.flat.numel() * 4


def allreduce_gaussian_grads(bucket: FlatGradBucket, group: Optional[dist.ProcessGroup] = None,
                             async_op: bool = False):
    """SUM-all-reduces the packed gradients across ranks in one collective and re-attaches them.
    With world_size 1 (or no process group) there is nothing to exchange: gradients stay where autograd
    put them and nothing is copied."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return None
    bucket.pack()
    work = dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    if work is None or not async_op:
        bucket.unpack()
    return work


def shard_views(views: Sequence, rank: Optional[int] = None, world_size: Optional[int] = None):
    """The cameras rank `rank` renders this iteration: views[rank::world_size] (one camera per GPU when
    len(views) == world_size, the C4 configuration)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    return list(views[rank::world_size])


def render_views_and_reduce(render_fn, views: Iterable, bucket: FlatGradBucket,
                            group: Optional[dist.ProcessGroup] = None):
    """One data-parallel iteration: `render_fn(view)` must run forward+backward for one camera and
    accumulate into p.grad; afterwards gradients are summed over ranks.  Returns what render_fn
    returned for each local view."""
    for p in bucket.params:
        p.grad = None
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if multi:
        bucket.arm()            # the first view's gradients are born in the flat buffer
    try:
        outs = [render_fn(v) for v in views]
    finally:
        if multi:
            bucket.disarm()     # nothing consumed it (no backward ran): do not leak into a later step
    allreduce_gaussian_grads(bucket, group)
    return outs
