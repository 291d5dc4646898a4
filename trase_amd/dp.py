"""Data parallelism over camera views: one process per GPU renders a different
view; the only exchange step is ONE all-reduce of a flat per-Gaussian gradient bucket (RCCL over
xGMI on the GPU box, gloo in the CPU tests).  The reference itself is single-process."""
from __future__ import annotations

from typing import Iterable, List

import torch


class FlatGradBucket:
    """Makes ``.grad`` of every parameter a view into one contiguous buffer, so that the step ends
    with a single collective of sum(numel) floats (364 B/Gaussian for the TRASE parameter set)."""

    def __init__(self, params: Iterable[torch.Tensor]):
        self.params: List[torch.Tensor] = list(params)
        assert self.params, "no parameters"
        dev, dt = self.params[0].device, self.params[0].dtype
        self.flat = torch.zeros(sum(p.numel() for p in self.params), device=dev, dtype=dt)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def sink(self):
        """{param.data_ptr(): view of its slice} for ``trase_amd.renderer.set_grad_sink``: the fused backward writes
        each gradient once, straight into the bucket.  Use with ``detach_grads()`` before every backward (autograd adopts a
        gradient without copying only when ``.grad`` is None); parameters that receive no gradient in a step keep stale
        bucket contents -- ``zero()`` first if that can happen."""
        out, off = {}, 0
        for p in self.params:
            out[p.data_ptr()] = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        return out

    def detach_grads(self):
        for p in self.params:
            p.grad = None

    def adopted(self) -> bool:
        """True when every parameter's ``.grad`` lives inside the bucket (the sink path was taken)."""
        lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + self.flat.numel() * self.flat.element_size()
        return all(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in self.params)

    def allreduce(self, average: bool = False):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if average:
            self.flat.div_(dist.get_world_size())


def allreduce_densify_stats(xyz_gradient_accum, denom, max_radii2D):
    """Keeps replicas' densify/prune decisions identical (train.py:364-366): SUM, SUM, MAX."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM)
    dist.all_reduce(denom, op=dist.ReduceOp.SUM)
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX)
