"""Data-parallel training over the GPUs of one node: one process per GPU, meshes sharded across
ranks as independent batch items, the gradient sync is an RCCL all-reduce over ONE flat fp32 gradient buffer
(SURVEY.md 8e): per-block ranges go out on a side stream as soon as their backward is done, the small rest after it.
The reference has no distributed code at all; this is new.

``FlatParams`` re-homes every parameter (and its .grad) of a module into one contiguous buffer so
that (a) the gradient sync is a single collective over xGMI instead of 40 small ones (the whole
model is ~1.9 MB, latency-bound), and (b) the optimizer is a single-tensor update.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


class FlatParams:
    def __init__(self, module: torch.nn.Module, direct_sinks: bool = True, overlap: bool = True):
        """direct_sinks: let the HIP ops accumulate parameter gradients straight into the bucket (ops._deliver).  Turn it off
        when backward passes run concurrently on several streams: the direct adds are not ordered across streams, autograd's
        AccumulateGrad is.
        overlap: in a multi-rank job, all-reduce the gradient range of a DiffusionNetBlock as soon as its backward has delivered it
        (on a side stream, under the backward of the blocks before it); all_reduce_mean() then only sends what is left."""
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
        # ---- per-block buckets: the contiguous flat range of every DiffusionNetBlock, in parameter order
        self.buckets: List[tuple] = []
        self._pending, self._sent = {}, []
        self._hold = 0                    # > 0 inside no_sync(): backward passes accumulate locally, nothing is sent
        self._side = None
        index_of = {id(p): i for i, p in enumerate(params)}
        blocks = getattr(module, "blocks", None) if overlap else None
        if blocks and self.direct_sinks:
            for bi, blk in enumerate(blocks):
                ids = sorted(index_of[id(p)] for p in blk.parameters())
                if not ids or ids != list(range(ids[0], ids[-1] + 1)):
                    self.buckets = []
                    break
                lo, hi = offs[ids[0]], offs[ids[-1]] + (sizes[ids[-1]] + 3) // 4 * 4
                self.buckets.append((lo, hi))
                blk._cfg.grad_hook = (lambda k: (lambda: self._bucket_ready(k)))(bi)
                blk._cfg.grad_pre_hook = (lambda k: (lambda: self._bucket_reopen(k)))(bi)
        self._force_collectives = False   # tests: run the collective path at world size 1
        self.suspend_overlap = False      # graphs.GraphedTrainStep(all_reduce="eager"): backward is being captured, send nothing from it

    # ------------------------------------------------------------------ overlap machinery
    def _active(self, group=None):
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or self._force_collectives)

    def no_sync(self):
        """Context manager for gradient accumulation: backward passes inside it only accumulate into the local bucket; the first
        backward outside it (the last micro-batch) sends the per-block ranges.  Without it every backward of a step still gives
        the right result (see ``_bucket_reopen``), at the price of one collective per block and backward."""
        flat = self

        class _Hold:
            def __enter__(self_):
                flat._hold += 1

            def __exit__(self_, *exc):
                flat._hold -= 1
                return False
        return _Hold()

    def _bucket_reopen(self, k):
        """Called by ops.BlockFn.backward BEFORE it writes block k's gradients.  If the block's range already went out in this step
        (a second backward before all_reduce_mean(): gradient accumulation, a block used twice), the bucket holds the SUM over
        ranks of the earlier contributions and its collective may still be in flight: wait for it, scale the range back by
        1 / world (so that the next SUM over ranks restores it), and mark it unsent -- the new local gradients are then added to it
        and it is sent again.  (without this the later contributions were never reduced and raced the collective.)"""
        if k not in self._sent:
            return
        w = self._pending.pop(k, None)
        if w is not None:
            w.wait()
        if self._side is not None and self.grad.is_cuda:
            torch.cuda.current_stream(self.grad.device).wait_stream(self._side)
        lo, hi = self.buckets[k]
        self.grad[lo:hi].div_(dist.get_world_size())
        self._sent.remove(k)

    def _bucket_ready(self, k):
        """Called by ops.BlockFn.backward once block k's gradients sit in the flat bucket (stream-ordered on the current stream)."""
        if self.suspend_overlap or self._hold or not self._active() or k in self._sent:
            return
        lo, hi = self.buckets[k]
        self._sent.append(k)
        view = self.grad[lo:hi]
        if self.grad.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(self.grad.device)
            self._side.wait_stream(torch.cuda.current_stream(self.grad.device))   # the bucket's adds have been enqueued
            with torch.cuda.stream(self._side):
                self._pending[k] = dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True)
        else:
            self._pending[k] = dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True)

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
        self._sent, self._pending = [], {}
        if self.direct_sinks:     # a zeroed sink may be written by the gradient kernel itself (ops._grad_out): store == accumulate
            for p in self.params:
                sink = getattr(p, "_dn_grad_sink", None)
                if sink is not None:
                    sink._dn_fresh = True

    def all_reduce_mean(self, group=None):
        """Sum over ranks, then divide by the world size.  Blocks whose range went out during backward are skipped; what is left
        (head / tail linears, or everything when nothing was overlapped) goes in as few contiguous collectives as possible."""
        if not self._active(group):
            return
        world = dist.get_world_size(group)
        done = sorted(self.buckets[k] for k in self._sent)
        cur, rest = 0, []
        for lo, hi in done:
            if lo > cur:
                rest.append((cur, lo))
            cur = max(cur, hi)
        if cur < self.grad.numel():
            rest.append((cur, self.grad.numel()))
        for lo, hi in rest:
            dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=group)
        for w in self._pending.values():
            w.wait()          # orders the current stream behind the side-stream collectives
        if self._side is not None:
            torch.cuda.current_stream(self.grad.device).wait_stream(self._side)
        self._pending, self._sent = {}, []
        self.grad.div_(world)


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
