"""Data-parallel training over the GPUs of one node: one process per GPU, meshes sharded across
ranks as independent batch items, ONE RCCL all-reduce of a single flat fp32 gradient bucket per step
(SURVEY.md 8e).  The reference has no distributed code at all; this is new.

``FlatParams`` re-homes every parameter (and its .grad) of a module into one contiguous buffer so
that (a) the gradient sync is a single collective over xGMI instead of 40 small ones (the whole
model is ~1.9 MB, latency-bound), and (b) the optimizer is a single-tensor update.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


class FlatParams:
    def __init__(self, module: torch.nn.Module, direct_sinks: bool = True):
        """direct_sinks: let the HIP ops accumulate parameter gradients straight into the bucket (ops._deliver).  Turn it off
        when backward passes run concurrently on several streams: the direct adds are not ordered across streams, autograd's
        AccumulateGrad is."""
        self.direct_sinks = bool(direct_sinks)
        params = [p for p in module.parameters()]
        if not params:
            raise ValueError("module has no parameters")
        dev, dt = params[0].device, params[0].dtype
        sizes = [p.numel() for p in params]
        # keep every slice 16-byte aligned so the HIP kernels' vector paths stay usable
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 3) // 4 * 4
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        self.grad = torch.zeros(total, dtype=dt, device=dev)
        with torch.no_grad():
            for p, o, n in zip(params, offs, sizes):
                self.flat[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[o:o + n].view(p.shape)
                p.grad = self.grad[o:o + n].view(p.shape)
                p._dn_grad_sink = p.grad if self.direct_sinks else None   # ops._deliver: one multi-tensor add per op
        self.params, self.offsets, self.sizes = params, offs, sizes
        self.master = torch.nn.Parameter(self.flat, requires_grad=True)   # what the optimizer updates
        self.master.grad = self.grad

    def rebind(self):
        """Re-point parameters at the flat buffers (needed after code that re-binds ``p.data``,
        e.g. the diffusion-time clamp of layers.py:48-49, which keeps values but not storage)."""
        with torch.no_grad():
            for p, o, n in zip(self.params, self.offsets, self.sizes):
                if p.data_ptr() != self.flat.data_ptr() + o * self.flat.element_size():
                    self.flat[o:o + n].copy_(p.detach().reshape(-1))
                    p.data = self.flat[o:o + n].view(p.shape)
                if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + o * self.grad.element_size():
                    p.grad = self.grad[o:o + n].view(p.shape)
                p._dn_grad_sink = p.grad if self.direct_sinks else None

    def zero_grad(self):
        self.grad.zero_()

    def all_reduce_mean(self, group=None):
        """One collective for the whole model: sum over ranks, then divide by the world size."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
            self.grad.div_(dist.get_world_size(group))


def shard_by_cost(costs: Sequence[float], world: int) -> List[List[int]]:
    """Greedy longest-processing-time assignment of items (meshes) to ranks; returns the item
    indices of every rank.  Cost of a mesh ~ its vertex count (work is linear in V)."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    loads = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: loads[k])
        out[r].append(i)
        loads[r] += costs[i]
    return [sorted(ix) for ix in out]
