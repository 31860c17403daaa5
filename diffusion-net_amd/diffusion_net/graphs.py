"""HIP-graph capture of the training step for the launch-bound regime.

A DiffusionNet step is ~150 kernel launches.  At 160k vertices per step the GPU is the bottleneck and that is invisible; at the
sizes of the reference's own experiments (one ~7k-vertex mesh per step; 64 x ~2k-vertex meshes) the host cannot enqueue them as fast
as the GPU retires them.  ``GraphedTrainStep`` captures forward + loss + backward + optimizer update of ONE packed batch into a HIP
graph (``torch.cuda.CUDAGraph``: the library's launches are plain stream work -- no allocation, no synchronisation -- so they capture
like torch's own kernels) and replays it: one host call per step.

What makes the step capturable:
  * dropout masks are drawn inside the kernels from a seed; under capture the seed's varying part lives in a DEVICE word
    (``dn_block_params_t.drop_seed_dev``) that the graph itself advances with its first node, so every replay draws fresh masks;
  * the parameters live in ``dist.FlatParams`` (one buffer, gradients delivered by the ops straight into it), the optimizer is torch's
    Adam with ``capturable=True``;
  * inputs are static buffers owned by the object: ``step(x=..., labels=...)`` copies new values in before the replay;
  * the learning rate is a DEVICE tensor: a captured ``opt.step()`` would otherwise bake the Python float of capture time into the graph
    and silently ignore the reference loop's schedule (``param_group['lr'] = lr`` every 50 epochs, human_segmentation_original.py:91-96).
    The object installs one float32 device tensor per parameter group (shared by every graph of the optimizer) and, before each
    replay, copies in whatever the caller has assigned to ``param_group['lr']`` since -- a float or a tensor -- and re-installs the
    tensor, so the reference's idiom keeps working unchanged.
Data-parallel runs: ``all_reduce="eager"`` replays forward + loss + backward from the graph and then issues ONE flat RCCL all-reduce
and the optimizer update from the host (no collective is ever captured: the robust choice, and at 1.85 MB of gradients the lost
overlap is ~20 us); ``all_reduce=True`` captures the bucketed all-reduce of ``FlatParams`` with the step (the per-block collectives
fork from and join the captured stream; every rank replays its own graph, the collectives inside keep them in lock step).
"""
from __future__ import annotations

from typing import Optional

import torch

from .batch import GatherPattern, MeshBatch
from .dist import FlatParams

_GOLDEN = 0x9E3779B97F4A7C15


class GraphedTrainStep:
    def __init__(self, model, flat: FlatParams, opt: torch.optim.Optimizer, mb: MeshBatch, gather: Optional[GatherPattern], x: torch.Tensor,
                 labels: torch.Tensor, smoothing: float = 0.0, warmup: int = 3, all_reduce: bool = False, pool=None):
        dev = x.device
        if dev.type != "cuda":
            raise RuntimeError("graph capture needs a ROCm device")
        if warmup < 1:
            raise ValueError("at least one eager warm-up step is needed: it creates the optimizer state the captured update works on")
        self.model, self.flat, self.opt, self.mb, self.gather, self.smoothing = model, flat, opt, mb, gather, float(smoothing)
        if all_reduce not in (False, True, "eager", "captured"):
            raise ValueError("all_reduce must be False, True / 'captured', or 'eager'")
        self.all_reduce = "captured" if all_reduce is True else all_reduce
        if self.all_reduce == "eager":
            flat.suspend_overlap = True            # backward only ever runs inside the graph: no collective may be launched from it
        self.x = x.detach().clone()
        self.labels = labels.detach().clone()
        self._lr = []                              # one device tensor per parameter group: what the captured update reads
        for pg in opt.param_groups:
            lr = pg["lr"]
            if not (isinstance(lr, torch.Tensor) and lr.device == dev):
                lr = torch.tensor(float(lr), dtype=torch.float32, device=dev)
                pg["lr"] = lr
            self._lr.append(lr)
        # dropout: constant host part per block + one device word advanced by the graph.  The host part is drawn from torch's CPU
        # generator (so torch.manual_seed governs the masks, as in the eager path, layers._dropout_masks) and mixed with the rank:
        # replicas of a data-parallel job are seeded identically, their masks must not be
        self.seed = torch.zeros(1, dtype=torch.int64, device=dev)
        base = int(torch.randint(1, 2 ** 62, (1,), dtype=torch.int64).item())
        rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
        for bi, blk in enumerate(model.blocks):
            blk._graph_seed = (((base + 0x51ED27 * (bi + 1) + _GOLDEN * (rank + 1)) % (2 ** 62)) | 1, self.seed)
        step = _GOLDEN - (1 << 64)                 # the 64-bit increment as a signed value
        self._advance = lambda: self.seed.add_(step)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):              # eager warm-up on a side stream (allocator, per-device kernel attributes)
            for _ in range(warmup):
                wl, _ = self._body()
                self._tail()
            self.warm_loss = wl.detach().clone()   # loss of the last eager warm-up step (GraphedEpoch returns it for a batch's first visit)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: the RCCL watchdog thread (event queries on finished collectives) or a data-loader thread may call
        # into HIP while this thread captures; in the default global mode such a call invalidates the capture
        # pool: graphs that never replay concurrently may share one private memory pool (GraphedEpoch: one graph per packed batch)
        with torch.cuda.graph(self.graph, pool=pool, capture_error_mode="thread_local"):
            self.loss, self.preds = self._body()
        torch.cuda.synchronize(dev)

    def _body(self):
        self._advance()
        self.flat.zero_grad()
        preds, loss = self.model.forward_packed_loss(self.x, self.mb, self.gather, self.labels, self.smoothing)
        loss.backward()
        if self.all_reduce == "eager":             # the rest of the step (all-reduce, update) is issued from the host: _tail()
            return loss, preds
        if self.all_reduce == "captured":          # RCCL collectives are stream work too: the per-block side-stream all-reduces fork
            self.flat.all_reduce_mean()            # from and join the captured stream
        self.opt.step()
        return loss, preds

    def _tail(self):
        if self.all_reduce == "eager":
            self.flat.all_reduce_mean()
            self.opt.step()

    def release(self):
        for blk in self.model.blocks:
            blk._graph_seed = None
        if self.all_reduce == "eager":
            self.flat.suspend_overlap = False

    def _sync_lr(self):
        """A learning rate assigned since the capture (``param_group['lr'] = 5e-4``) goes INTO the tensor the graph reads."""
        for pg, t in zip(self.opt.param_groups, self._lr):
            cur = pg["lr"]
            if cur is not t:
                if isinstance(cur, torch.Tensor):
                    t.copy_(cur.detach().reshape(()).to(t.dtype), non_blocking=True)
                else:
                    t.fill_(float(cur))
                pg["lr"] = t

    def step(self, x: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None):
        """One optimizer step.  Returns the (static) loss tensor of this replay; ``self.preds`` holds the log-probabilities."""
        self._sync_lr()
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if labels is not None:
            self.labels.copy_(labels, non_blocking=True)
        self.graph.replay()
        self._tail()
        return self.loss



class GraphedEpoch:
    """Graph replay over CHANGING batches: the reference's loop visits a different mesh (here: a different packed batch) every step
    (human_segmentation_original.py:105-120), while a ``GraphedTrainStep`` is bound to the operators of ONE batch.  This keeps one captured
    step per packed batch -- keyed by the identity of its ``MeshBatch`` and gather pattern, least recently used dropped beyond
    ``max_graphs`` -- all sharing the model, the flat parameter / gradient buffers, the optimizer state and ONE private memory pool (the
    graphs never replay concurrently, so the activations of one step may live where another step's did).  A batch is captured the first
    time it is seen: that visit runs ONE eager step (the visit's optimizer step) and captures the graph; from its second visit on a
    step is one host call.

    ``step(mb, gather, x, labels)`` -> the loss tensor of that replay.  It is a static tensor of the graph just replayed and lives in the
    shared pool: read it (``.item()``, a copy) before the next ``step``."""

    def __init__(self, model, flat: FlatParams, opt: torch.optim.Optimizer, smoothing: float = 0.0, max_graphs: int = 64, all_reduce=False):
        from collections import OrderedDict
        self.model, self.flat, self.opt, self.smoothing, self.max_graphs, self.all_reduce = model, flat, opt, smoothing, int(max_graphs), all_reduce
        self._graphs = OrderedDict()
        self._pool = None
        self.stats = {"captures": 0, "replays": 0, "evictions": 0}

    def step(self, mb: MeshBatch, gather: Optional[GatherPattern], x: torch.Tensor, labels: torch.Tensor):
        key = (id(mb), id(gather))
        ent = self._graphs.get(key)
        if ent is not None and (ent[1] is not mb or ent[2] is not gather):     # an id re-used by a new object
            ent = None
        if ent is None:
            for gs_, _, _ in self._graphs.values():      # (a learning rate assigned since the last step: into the shared tensors first)
                gs_._sync_lr()
                break
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            # one eager warm-up step (THIS visit's optimizer step) + the capture
            gs = GraphedTrainStep(self.model, self.flat, self.opt, mb, gather, x, labels, smoothing=self.smoothing,
                                  warmup=1, all_reduce=self.all_reduce, pool=self._pool)
            self._graphs[key] = (gs, mb, gather)
            self.stats["captures"] += 1
            while len(self._graphs) > self.max_graphs:
                _, (old, _, _) = self._graphs.popitem(last=False)
                self.stats["evictions"] += 1
                del old
            return gs.warm_loss
        self._graphs.move_to_end(key)
        self.stats["replays"] += 1
        return ent[0].step(x, labels)

    def release(self):
        for gs, _, _ in self._graphs.values():
            gs.release()
        self._graphs.clear()
