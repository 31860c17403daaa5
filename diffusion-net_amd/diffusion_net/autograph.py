"""Automatic HIP-graph replay behind the REFERENCE signature (launch-bound regime).

The reference's own train loop (human_segmentation_original.py:105-148) calls ``model(x, mass, L, evals, evecs, gradX, gradY, faces)`` on
ONE ~7k-vertex mesh per step: ~100 launches forward and ~150 backward, each 5-30 us of device time, which the host cannot enqueue as fast
as the device retires them.  ``graphs.GraphedTrainStep`` solves that for callers who restructure their loop; this module does it for the
ones who do not: once the operator cache (batch.OperatorCache) has seen a mesh a few times with the same model, ``DiffusionNet.forward``
captures the packed forward of that (model, mesh, mode) into a HIP graph, and -- when gradients are wanted -- the backward
(``torch.autograd.grad`` of the captured forward) into a second one.  Later calls copy the features into a static buffer, replay, and hand
the result to autograd through one ``autograd.Function`` whose backward replays the second graph: two host calls instead of ~250, the
same kernels on the same operands.  The loss, the optimizer and everything else in the caller's loop stay as they are.

Rules that keep this exact:
  * a graph is bound to the parameters' storage (addresses, dtypes, requires_grad flags), the mode (train / eval, grad / no-grad) and the
    feature shape; any change falls back to the eager path and captures again after the usual number of sightings;
  * all graphs of a device share one memory pool, so the activations a captured forward saves for its backward live in memory that other
    graphs use too.  They are safe because graphs of one pool never interleave: while a replayed forward still waits for its backward
    (its autograd node is alive and has not run), every other call takes the eager path (``_Gate``).  A backward that arrives after its
    activations were overwritten cannot happen silently: the generation counter raises;
  * dropout: the masks come from the in-kernel generator with a device-side seed word that the graph advances itself, so every replay
    draws fresh masks (as ``graphs.GraphedTrainStep``); the constant part is drawn from torch's CPU generator at capture time;
  * parameter gradients leave the graph as ONE flat buffer, cloned once per backward and handed to autograd as views: accumulation over
    several backward calls, ``zero_grad(set_to_none=...)`` either way, parameter hooks and DistributedDataParallel behave as in eager mode;
  * anything unusual -- features that require grad, module hooks inside the model, test hooks, gradient sinks of ``dist.FlatParams``,
    an enclosing capture or compiler trace, a failed capture -- means the eager path, permanently for that (model, mesh, mode) after a
    failed capture (one warning).
Only batches in the launch-bound regime are captured (``max_work``: vertices x C_width <= 4 M, i.e. up to 32k vertices at C_width 128).
``diffusion_net.autograph.enabled = False`` (or DN_AUTOGRAPH=0 in the environment) switches it off.
"""
from __future__ import annotations

import contextlib
import os
import warnings
import weakref

import torch

enabled = os.environ.get("DN_AUTOGRAPH", "1") != "0"
warm_calls = 2          # eager sightings of a (model, mesh, mode) before it is captured
max_work = 1 << 22      # capture only where launches are the bottleneck: vertices x C_width up to this (32k vertices at C_width = 128).  Above
                        # it the device is the bottleneck, a replay buys nothing, and the graph would pin the batch's activations in the pool
stats = {"captures": 0, "replays_fwd": 0, "replays_bwd": 0, "eager_pending": 0, "failed": 0}

_GOLDEN = 0x9E3779B97F4A7C15


class _Token:
    __slots__ = ("__weakref__",)


class _Gate:
    """One pending forward per device: between the replay of a captured forward and the replay of its backward no other graph of the
    shared pool may run."""

    def __init__(self):
        self._armed = {}

    def pending(self, dev):
        r = self._armed.get(dev)
        if r is None:
            return False
        if r() is None:            # the autograd node died without a backward (the caller dropped the result)
            del self._armed[dev]
            return False
        return True

    def arm(self, dev, token):
        self._armed[dev] = weakref.ref(token)

    def disarm(self, dev, token):
        r = self._armed.get(dev)
        if r is not None and r() is token:
            del self._armed[dev]


_gate = _Gate()


class _Order:
    """Graphs of one device share pool memory, so their replays must not overlap on the DEVICE either: the gate above serialises the
    host-side protocol, this orders the streams.  Per device it remembers the stream of the last replay and an event recorded behind it;
    a replay issued from another stream first waits for that event (forward calls from two streams could otherwise run two
    graphs concurrently in the same pool memory).  Also the pool-wide forward counter: a backward is valid only while no other forward
    of ANY graph of the pool has replayed since its own (the activations live in shared memory)."""

    def __init__(self):
        self._last = {}
        self.forwards = {}

    def before(self, dev):
        if dev.type != "cuda":
            return
        cur = torch.cuda.current_stream(dev)
        last = self._last.get(dev)
        if last is not None and last[0] != cur:
            cur.wait_event(last[1])

    def after(self, dev):
        if dev.type != "cuda":
            return
        cur = torch.cuda.current_stream(dev)
        last = self._last.get(dev)
        ev = last[1] if last is not None else torch.cuda.Event()
        ev.record(cur)
        self._last[dev] = (cur, ev)

    def bump(self, dev):
        self.forwards[dev] = self.forwards.get(dev, 0) + 1
        return self.forwards[dev]


_order = _Order()


class _Pool:
    """The memory pool the graphs of one device share.  A pool dies with its last graph (the allocator drops it when its use count
    reaches zero) and its handle must not be used again after that: ``users`` tracks the live owners, and a new handle is drawn when
    none is left."""

    def __init__(self):
        self.handle = torch.cuda.graph_pool_handle()
        self.users = weakref.WeakSet()


_pools = {}


def _pool_for(device, owner):
    p = _pools.get(device)
    if p is None or len(p.users) == 0:
        p = _pools[device] = _Pool()
    p.users.add(owner)
    return p


class _HipGraphs:
    """Capture backend: ``torch.cuda.CUDAGraph`` (a HIP graph on ROCm) in the device's shared pool."""

    @staticmethod
    def usable(x):
        # (a capture synchronises with the device: not inside a region where the caller has asked torch to flag synchronisations)
        return x.is_cuda and not torch.cuda.is_current_stream_capturing() and torch.cuda.get_sync_debug_mode() == 0

    _side = {}       # one warm-up stream per device (the library keeps a scratch buffer per stream it has run on)

    @staticmethod
    def warm(fn, device):
        side = _HipGraphs._side.get(device)
        if side is None:
            side = _HipGraphs._side[device] = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream(device).wait_stream(side)

    reruns = False

    @staticmethod
    def new_pool(device, owner):
        return _pool_for(device, owner)

    @staticmethod
    def capture(fn, device, pool, adopt=None):
        g = torch.cuda.CUDAGraph()
        # thread-local capture mode: a data-loader or watchdog thread may call into HIP while this thread captures
        with torch.cuda.graph(g, pool=pool.handle, capture_error_mode="thread_local"):
            res = fn()
        return g.replay, res


class RerunBackend:
    """TEST backend (no GPU in the CPU tier): a "replay" executes the recorded closure again and copies what it produced into the static
    buffers -- the buffer plumbing, the gate and the autograd wiring are the product's, only the graph is replaced."""
    reruns = True

    @staticmethod
    def usable(x):
        return True

    @staticmethod
    def warm(fn, device):
        fn()

    @staticmethod
    def new_pool(device, owner):
        return None

    @staticmethod
    def capture(fn, device, pool, adopt=None):
        res = fn()
        return (lambda: adopt(fn())), res


backend = _HipGraphs


def _blocks_plain(model):
    for blk in model.blocks:
        if blk.mask_provider is not None or blk.drop_seed_provider is not None or getattr(blk, "_graph_seed", None) is not None:
            return False
    return True


def _has_inner_hooks(model):
    for m in model.modules():
        if m is model:
            continue
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
            return True
    return False


class GraphedForward:
    """Captured ``model.forward_packed(x, mb, gather)`` (+ its backward when ``grad``) for one static feature buffer."""

    def __init__(self, model, x, mb, gather, grad):
        self.model = weakref.ref(model)
        self.device, self.grad = x.device, grad
        # The captured forward runs on ALIASES of the parameters (fresh leaves on the same storage), swapped into the modules for the
        # duration of a capture.  Reason: a parameter's gradient-accumulator node remembers the stream that was current when it was
        # created, and it lives as long as ANY autograd graph references it -- e.g. the caller's previous `loss`, built on the default
        # stream.  A captured backward that flows into such a node makes the engine synchronise the capture stream with that other
        # stream (event record + wait with a temporary event): the other stream joins the capture, the event is destroyed, and
        # hipStreamEndCapture dereferences it (observed: segmentation fault in hip::Stream::EndCapture).  Fresh leaves get fresh
        # accumulator nodes on the capture stream, and the graphs hold no reference into the caller's autograd state.
        slots, alias = [], {}
        for mod in model.modules():
            for name, p in mod._parameters.items():
                if p is not None:
                    if id(p) not in alias:
                        alias[id(p)] = p.detach().requires_grad_(p.requires_grad)
                    slots.append((mod, name, p, alias[id(p)]))
        self.real_params = [p for p in model.parameters() if p.requires_grad] if grad else []
        self.params = [alias[id(p)] for p in self.real_params]
        self.sig = self._signature(model)

        @contextlib.contextmanager
        def swapped():
            try:
                for mod, name, _, a in slots:
                    mod._parameters[name] = a
                yield
            finally:
                for mod, name, p, _ in slots:
                    mod._parameters[name] = p
        self.static_x = x.detach().clone()
        self.generation = 0
        self._mb = weakref.ref(mb)            # (weak: the batch owns this object through its graph table; a pending backward holds it strongly)
        pool = backend.new_pool(self.device, self)
        drop = model.training and any(blk.dropout for blk in model.blocks)
        self.seed = torch.zeros(1, dtype=torch.int64, device=x.device) if drop else None
        if drop:
            base = int(torch.randint(1, 2 ** 62, (1,), dtype=torch.int64).item())
            rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
            seeds = [(((base + 0x51ED27 * (bi + 1) + _GOLDEN * (rank + 1)) % (2 ** 62)) | 1, self.seed) for bi in range(len(model.blocks))]
        step = _GOLDEN - (1 << 64)

        def fwd():
            if drop:
                self.seed.add_(step)
            with swapped(), torch.set_grad_enabled(grad):
                return model.forward_packed(self.static_x, mb, gather)

        def fwd_bwd():
            out = fwd()
            if grad:
                torch.autograd.grad(out, self.params, torch.ones_like(out), allow_unused=True)

        try:
            if drop:
                for blk, s in zip(model.blocks, seeds):
                    blk._graph_seed = s
            backend.warm(fwd_bwd, self.device)

            def adopt_fwd(new):            # (RerunBackend only)
                self.static_out.copy_(new.detach())
                self._live_out = new
            self.fwd_replay, self._live_out = backend.capture(fwd, self.device, pool, adopt_fwd)
            self.static_out = self._live_out.detach()
            self.static_gout = self.flat = None
            self.slices = []
            if grad:
                self.static_gout = torch.zeros_like(self.static_out)

                def bwd():
                    gs = torch.autograd.grad(self._live_out, self.params, self.static_gout, allow_unused=True)
                    live = [g.reshape(-1) for g in gs if g is not None]
                    return gs, (torch.cat(live) if live else None)

                def adopt_bwd(new):        # (RerunBackend only)
                    if self.flat is not None:
                        self.flat.copy_(new[1])
                self.bwd_replay, (gs, self.flat) = backend.capture(bwd, self.device, pool, adopt_bwd)
                off = 0
                for g in gs:
                    if g is None:
                        self.slices.append(None)
                    else:
                        self.slices.append((off, g.numel(), tuple(g.shape)))
                        off += g.numel()
            if not backend.reruns:
                self._live_out = None      # the autograd graph of the captured forward has served its purpose
        finally:
            if drop:
                for blk in model.blocks:
                    blk._graph_seed = None

    @staticmethod
    def _signature(model):
        return tuple((p.data_ptr(), p.dtype, p.requires_grad) for p in model.parameters())

    def valid_for(self, model):
        return self.model() is model and self.sig == self._signature(model)

    def __call__(self, x):
        if not self.grad:
            _order.before(self.device)
            self.static_x.copy_(x, non_blocking=True)
            self.fwd_replay()
            _order.bump(self.device)
            stats["replays_fwd"] += 1
            out = self.static_out.clone()
            _order.after(self.device)
            return out
        return _Replay.apply(self, x, *self.real_params)


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gf, x, *params):
        _order.before(gf.device)
        gf.static_x.copy_(x, non_blocking=True)
        gf.fwd_replay()
        stats["replays_fwd"] += 1
        gf.generation += 1
        # generation: of this graph; pool_gen: of the device's shared pool (any other graph's forward replay overwrites these activations too)
        ctx.gf, ctx.generation, ctx.pool_gen, ctx.token = gf, gf.generation, _order.bump(gf.device), _Token()
        ctx.mb = gf._mb()         # the packed operators the captured kernels read stay alive until the backward has run
        _gate.arm(gf.device, ctx.token)
        out = gf.static_out.clone()
        _order.after(gf.device)
        return out

    @staticmethod
    def backward(ctx, g):
        gf = ctx.gf
        if ctx.generation != gf.generation or ctx.pool_gen != _order.forwards.get(gf.device):
            raise RuntimeError("diffusion_net.autograph: the activations of this forward were overwritten by a later replay (of this or "
                               "another graph of the device's shared pool) before its backward ran (set diffusion_net.autograph.enabled = "
                               "False for this access pattern)")
        _order.before(gf.device)
        gf.static_gout.copy_(g, non_blocking=True)
        gf.bwd_replay()
        stats["replays_bwd"] += 1
        _gate.disarm(gf.device, ctx.token)
        if gf.flat is None:
            _order.after(gf.device)
            return (None, None) + (None,) * len(gf.slices)
        flat = gf.flat.clone()        # the caller owns its gradients: the static buffer is rewritten by the next replay
        _order.after(gf.device)
        return (None, None) + tuple(None if s is None else flat[s[0]:s[0] + s[1]].view(s[2]) for s in gf.slices)


class _Record:
    __slots__ = ("calls", "gf", "failed")

    def __init__(self):
        self.calls, self.gf, self.failed = 0, None, False


def run(model, x2d, mb, gather):
    """The packed forward through a captured graph, or None when this call has to take the eager path."""
    if not enabled or x2d.requires_grad or not backend.usable(x2d) or torch.compiler.is_compiling():
        return None
    if torch.is_autocast_enabled() or torch.is_anomaly_enabled() or not _blocks_plain(model) or x2d.shape[0] * model.C_width > max_work:
        return None
    grad = torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())
    table = mb.__dict__.setdefault("_autograph", {})
    key = (id(model), model.training, grad, tuple(x2d.shape), x2d.dtype, id(gather), model.outputs_at, id(model.last_activation))
    rec = table.get(key)
    if rec is None:
        for k, r in list(table.items()):      # graphs of models that no longer exist (they pin the old parameters' storage through their aliases)
            if r.gf is not None and r.gf.model() is None:
                del table[k]
        rec = table[key] = _Record()
    rec.calls += 1
    if rec.failed:
        return None
    if rec.gf is not None and not rec.gf.valid_for(model):      # parameters re-allocated (model.to / .half / new requires_grad flags)
        rec.gf, rec.calls = None, 1
    if _gate.pending(x2d.device):
        stats["eager_pending"] += 1
        return None
    if rec.gf is None:
        if rec.calls <= warm_calls:
            return None
        if any(getattr(p, "_dn_grad_sink", None) is not None for p in model.parameters()) or _has_inner_hooks(model):
            rec.failed = True
            return None
        try:
            rec.gf = GraphedForward(model, x2d, mb, gather, grad)
            stats["captures"] += 1
        except Exception as exc:      # noqa: BLE001  (a capture that fails for any reason leaves the eager path, which is always correct)
            rec.failed = True
            stats["failed"] += 1
            warnings.warn("diffusion_net.autograph: capture failed (%s: %s); this mesh stays on the eager path" % (type(exc).__name__, exc))
            return None
    return rec.gf(x2d)
