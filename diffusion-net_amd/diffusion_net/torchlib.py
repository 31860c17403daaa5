"""``torch.library`` registration of the hot ops ("called from the Python torch.library op implementations").

The eager path calls the ``torch.autograd.Function``s of ``ops.py`` directly.  They take Python objects (a packed ``MeshBatch``, a
``BlockConfig``) and call the C ABI through ctypes -- opaque to ``torch.compile`` / ``torch.export``, which would either try to trace
into the ctypes calls or break the graph around them.  Here every op of the packed forward is ALSO registered as a custom operator
(``torch.ops.diffusion_net.*``) with a fake (meta) kernel and an autograd formula whose backward is itself a registered op, so that a
compiled ``DiffusionNet.forward_packed`` is one graph with the ops as opaque nodes (``tests/test_gpu_parity.py::
test_torch_compile_packed_forward``: ``fullgraph=True``).  The implementations are the SAME code: each custom op runs the
forward / backward of the matching ``ops.*Fn`` on a stand-in context.  ``layers.py`` routes through these ops only while a compiler is
tracing (``torch.compiler.is_compiling()``); eager execution keeps its direct path (gradient sinks, dist hooks, magnitude tags).

Non-tensor operands travel as integer handles into a weak registry (the caller owns the objects, as with the eager path).  The eager
autograd formulas pin the objects their backward looks up for as long as the autograd node lives (``_hold``); a COMPILED graph calls the
forward and backward ops directly and holds only the integers: there the caller must keep ``mb`` / ``gather`` alive until the backward has
run (a dead handle raises a RuntimeError that says so).  Handles are graph constants for ``torch.compile``: a compiled
``forward_packed`` is specialised to its ``MeshBatch`` and recompiles for another one -- compile per static batch (as
``graphs.GraphedTrainStep`` captures per batch), not inside a loop over meshes.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import Tensor
from torch.library import custom_op

from . import _hip, ops
import weakref

from .batch import handle_object as _obj   # handles are attributes of the objects (MeshBatch.handle, ...), assigned at construction
from .batch import pin_handle as _pin, unpin_handle as _unpin


def _hold(ctx, *handles):
    """Keep the objects behind ``handles`` alive as long as the autograd context ``ctx`` is (no-op while a compiler traces with fake tensors:
    nothing will look the handles up through this context)."""
    if torch.compiler.is_compiling():
        return
    for k in handles:
        if isinstance(k, int) and k:
            try:
                _pin(k)
                weakref.finalize(ctx, _unpin, k)
            except TypeError:          # a context type that cannot be weakly referenced: fall back to the caller-owns rule
                _unpin(k)


class _Ctx:
    """What the Functions of ops.py use of an autograd context."""

    def __init__(self, needs):
        self.needs_input_grad = tuple(needs)
        self.saved_tensors = ()

    def save_for_backward(self, *ts):
        self.saved_tensors = ts


def _e(like: Tensor) -> Tensor:
    return like.new_empty(0)


# ------------------------------------------------------------------------------------------------ nn.Linear on the row axis
@custom_op("diffusion_net::linear", mutates_args=())
def linear(x: Tensor, W: Tensor, b: Tensor, mb: int) -> Tensor:
    return ops.LinearFn.forward(_Ctx((True, True, True, False)), x, W, b, _obj(mb))


@linear.register_fake
def _(x, W, b, mb):
    return x.new_empty(x.shape[0], W.shape[0])


@custom_op("diffusion_net::linear_bwd", mutates_args=())
def linear_bwd(d_out: Tensor, x: Tensor, W: Tensor, mb: int) -> List[Tensor]:
    ctx = _Ctx((True, True, True, False))
    ctx.sinks, ctx.mb, ctx.saved_tensors = [None, None], _obj(mb), (x, W)
    d_x, dW, db, _ = ops.LinearFn.backward(ctx, d_out)
    return [d_x, dW, db]


@linear_bwd.register_fake
def _(d_out, x, W, mb):
    return [torch.empty_like(x), torch.empty_like(W), W.new_empty(W.shape[0])]


def _linear_setup(ctx, inputs, output):
    x, W, b, mb = inputs
    ctx.save_for_backward(x, W)
    ctx.mb = mb
    _hold(ctx, mb)


def _linear_backward(ctx, g):
    x, W = ctx.saved_tensors
    d_x, dW, db = linear_bwd(g.contiguous(), x, W, ctx.mb)
    return d_x, dW, db, None


linear.register_autograd(_linear_backward, setup_context=_linear_setup)


# ------------------------------------------------------------------------------------------------ fused DiffusionNetBlock
def _masks(seed: int, seed_dev: Optional[Tensor]):
    if seed == 0:
        return None
    return (seed, seed_dev) if seed_dev is not None else seed


@custom_op("diffusion_net::block", mutates_args=())
def block(x: Tensor, time: Tensor, A_re: Optional[Tensor], A_im: Optional[Tensor], wb: List[Tensor], mb: int, cfg: int, seed: int,
          seed_dev: Optional[Tensor], n_mesh: int, k_eig: int) -> List[Tensor]:
    """-> [out, xs, xd, words, (gx, gy, g, bre, bim), h_0 ...]: the output and what the backward needs besides the inputs."""
    ctx = _Ctx([False] * 3 + [True] * (4 + len(wb)))
    ctx.no_clamp = True        # `time` is an input of a custom op that declares no mutation: the caller has clamped it (layers.forward_packed)
    _obj(cfg)._grad_enabled = True      # (this op always returns what its backward op needs)
    out = ops.BlockFn.forward(ctx, _obj(mb), _obj(cfg), _masks(seed, seed_dev), x, time, A_re, A_im, *wb)
    return [out] + list(ctx.saved_tensors[2:5 + ctx.n_feat + ctx.n_h])


@block.register_fake
def _(x, time, A_re, A_im, wb, mb, cfg, seed, seed_dev, n_mesh, k_eig):
    # (shapes from the operands alone: the handles may be symbolic while tracing)
    V, Cw = x.shape[0], x.shape[1]
    new = lambda *s: x.new_empty(*s)
    outs = [new(V, Cw), new(n_mesh, k_eig, Cw), new(V, Cw), new(_hip.BLOCK_AMAX_WORDS + 1)]
    outs += [new(V, Cw) for _ in range(5 if A_re is not None else 0)]
    outs += [new(V, w.shape[0]) for w in wb[0:-2:2]]
    return outs


@custom_op("diffusion_net::block_bwd", mutates_args=())
def block_bwd(d_out: Tensor, x: Tensor, time: Tensor, A_re: Optional[Tensor], A_im: Optional[Tensor], wb: List[Tensor], saved: List[Tensor],
              mb: int, cfg: int, seed: int, seed_dev: Optional[Tensor]) -> List[Tensor]:
    """-> [d_x, d_time, dA_re | empty, dA_im | empty, dW_0, db_0, ...]"""
    c = _obj(cfg)
    ctx = _Ctx([False] * 3 + [True] * (4 + len(wb)))
    ctx.mb, ctx.cfg, ctx.masks = _obj(mb), c, _masks(seed, seed_dev)
    ctx.n_feat, ctx.n_h = (5 if c.with_grad else 0), c.n_mlp - 1
    ctx.has = (A_re is not None, A_im is not None)
    ctx.sinks = [None] * (3 + len(wb))
    ctx.saved_tensors = (x, time, *saved, *wb[0::2], *wb[1::2], *([A_re] if A_re is not None else []), *([A_im] if A_im is not None else []))
    g = ops.BlockFn.backward(ctx, d_out)          # (None, None, None, d_x, d_time, dA_re, dA_im, *wb)
    d_x, d_time, dA_re, dA_im = g[3:7]
    return [d_x, d_time, dA_re if dA_re is not None else _e(x), dA_im if dA_im is not None else _e(x), *g[7:]]


@block_bwd.register_fake
def _(d_out, x, time, A_re, A_im, wb, saved, mb, cfg, seed, seed_dev):
    return [torch.empty_like(x), torch.empty_like(time), torch.empty_like(A_re) if A_re is not None else _e(x),
            torch.empty_like(A_im) if A_im is not None else _e(x)] + [torch.empty_like(t) for t in wb]


def _block_setup(ctx, inputs, output):
    x, time, A_re, A_im, wb, mb, cfg, seed, seed_dev, _n_mesh, _k_eig = inputs
    ctx.n_wb = len(wb)
    ctx.has = (A_re is not None, A_im is not None, seed_dev is not None)
    ctx.ints = (mb, cfg, seed)
    _hold(ctx, mb, cfg)
    ctx.save_for_backward(x, time, *([A_re] if A_re is not None else []), *([A_im] if A_im is not None else []), *wb, *output[1:],
                          *([seed_dev] if seed_dev is not None else []))


def _block_backward(ctx, grads):
    sav = list(ctx.saved_tensors)
    x, time = sav[0], sav[1]
    pos = 2
    A_re = A_im = seed_dev = None
    if ctx.has[0]:
        A_re = sav[pos]; pos += 1
    if ctx.has[1]:
        A_im = sav[pos]; pos += 1
    wb = sav[pos:pos + ctx.n_wb]; pos += ctx.n_wb
    if ctx.has[2]:
        seed_dev = sav[-1]
        saved = sav[pos:-1]
    else:
        saved = sav[pos:]
    mb, cfg, seed = ctx.ints
    g = block_bwd(grads[0].contiguous(), x, time, A_re, A_im, wb, saved, mb, cfg, seed, seed_dev)
    return (g[0], g[1], g[2] if ctx.has[0] else None, g[3] if ctx.has[1] else None, list(g[4:]), None, None, None, None, None, None)


block.register_autograd(_block_backward, setup_context=_block_setup)


# ------------------------------------------------------------------------------------------------ output remaps
@custom_op("diffusion_net::gather_mean", mutates_args=())
def gather_mean(x: Tensor, pat: int, n_out: int) -> Tensor:
    return ops.GatherMeanFn.forward(_Ctx((True, False)), x, _obj(pat))


@gather_mean.register_fake
def _(x, pat, n_out):
    return x.new_empty(n_out, x.shape[1])


@custom_op("diffusion_net::gather_mean_bwd", mutates_args=())
def gather_mean_bwd(d_out: Tensor, pat: int, n_src: int) -> Tensor:
    ctx = _Ctx((True, False))
    ctx.pat = _obj(pat)
    return ops.GatherMeanFn.backward(ctx, d_out)[0]


@gather_mean_bwd.register_fake
def _(d_out, pat, n_src):
    return d_out.new_empty(n_src, d_out.shape[1])


def _gm_setup(ctx, inputs, output):
    ctx.pat, ctx.n_src = inputs[1], inputs[0].shape[0]
    _hold(ctx, ctx.pat)


gather_mean.register_autograd(lambda ctx, g: (gather_mean_bwd(g.contiguous(), ctx.pat, ctx.n_src), None, None), setup_context=_gm_setup)


@custom_op("diffusion_net::mass_mean", mutates_args=())
def mass_mean(x: Tensor, mb: int, n_mesh: int) -> List[Tensor]:
    ctx = _Ctx((True, False))
    out = ops.MassMeanFn.forward(ctx, x, _obj(mb))
    return [out, ctx.saved_tensors[0]]


@mass_mean.register_fake
def _(x, mb, n_mesh):
    return [x.new_empty(n_mesh, x.shape[1]), x.new_empty(n_mesh)]


@custom_op("diffusion_net::mass_mean_bwd", mutates_args=())
def mass_mean_bwd(d_out: Tensor, msum: Tensor, mb: int, v_total: int) -> Tensor:
    ctx = _Ctx((True, False))
    ctx.mb, ctx.saved_tensors = _obj(mb), (msum,)
    return ops.MassMeanFn.backward(ctx, d_out)[0]


@mass_mean_bwd.register_fake
def _(d_out, msum, mb, v_total):
    return d_out.new_empty(v_total, d_out.shape[1])


def _mm_setup(ctx, inputs, output):
    ctx.mb, ctx.v_total = inputs[1], inputs[0].shape[0]
    _hold(ctx, ctx.mb)
    ctx.save_for_backward(output[1])


mass_mean.register_autograd(lambda ctx, grads: (mass_mean_bwd(grads[0].contiguous(), ctx.saved_tensors[0], ctx.mb, ctx.v_total), None, None),
                            setup_context=_mm_setup)


# ------------------------------------------------------------------------------------------------ fused head (remap + log_softmax + loss)
@custom_op("diffusion_net::head", mutates_args=())
def head(x: Tensor, pat: int, labels: Optional[Tensor], log_softmax: bool, smoothing: float, n_out: int) -> List[Tensor]:
    """-> [log-probabilities, loss | empty, valid-row count | empty]"""
    ctx = _Ctx((True,) + (False,) * 5)
    logp, loss = ops.HeadFn.forward(ctx, x, _obj(pat), labels, log_softmax, smoothing, True)
    count = ctx.saved_tensors[-1] if labels is not None else _e(x)
    return [logp, loss if loss is not None else _e(x), count]


@head.register_fake
def _(x, pat, labels, log_softmax, smoothing, n_out):
    sc = lambda: x.new_empty(()) if labels is not None else _e(x)
    return [x.new_empty(n_out, x.shape[1]), sc(), sc()]


@custom_op("diffusion_net::head_bwd", mutates_args=())
def head_bwd(d_logp: Optional[Tensor], d_loss: Optional[Tensor], logp: Tensor, labels: Optional[Tensor], count: Optional[Tensor], pat: int,
             log_softmax: bool, smoothing: float, n_src: int) -> Tensor:
    ctx = _Ctx((True,) + (False,) * 5)
    p = _obj(pat)
    ctx.pat, ctx.lsm, ctx.smoothing = p, log_softmax, smoothing
    ctx.shape = (n_src, logp.shape[0], logp.shape[1])
    ctx.has = (True, labels is not None)
    ctx.saved_tensors = tuple(t for t in (logp, labels, count) if t is not None)
    return ops.HeadFn.backward(ctx, d_logp, d_loss)[0]


@head_bwd.register_fake
def _(d_logp, d_loss, logp, labels, count, pat, log_softmax, smoothing, n_src):
    return logp.new_empty(n_src, logp.shape[1])


def _head_setup(ctx, inputs, output):
    x, pat, labels, lsm, smoothing, _n_out = inputs
    ctx.meta = (pat, lsm, smoothing, x.shape[0], labels is not None)
    _hold(ctx, pat)
    ctx.save_for_backward(output[0], *([labels, output[2]] if labels is not None else []))


def _head_backward(ctx, grads):
    pat, lsm, smoothing, n_src, has_lab = ctx.meta
    sav = ctx.saved_tensors
    d_logp = grads[0].contiguous() if grads[0] is not None else None
    d_loss = grads[1].contiguous() if (has_lab and grads[1] is not None) else None
    d_x = head_bwd(d_logp, d_loss, sav[0], sav[1] if has_lab else None, sav[2] if has_lab else None, pat, lsm, smoothing, n_src)
    return d_x, None, None, None, None, None


head.register_autograd(_head_backward, setup_context=_head_setup)
