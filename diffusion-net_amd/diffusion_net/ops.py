"""torch.autograd bindings of the HIP hot ops (C ABI in include/diffnet_hip.h).

Each Function enqueues the library's kernels on the current torch stream; forward saves the
activations the hand-written backward needs.  No op has a torch/CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
from typing import List, Optional

import torch

from . import _hip
from .batch import GatherPattern, MeshBatch, register_handle


def _on_device(fn):
    """Run a Function's forward/backward with the tensors' device current: the library reads the current device for its
    per-device function attributes and CU counts, and a launch on another device's stream is an error -- `model.to('cuda:1')`
    without a surrounding `torch.cuda.device` must just work, as it does for torch's own ops."""
    @functools.wraps(fn)
    def wrapped(ctx, *args):
        dev = next((a.device for a in args if isinstance(a, torch.Tensor) and a.is_cuda), None)
        if dev is None or torch.cuda.current_device() == dev.index:
            return fn(ctx, *args)
        with torch.cuda.device(dev):
            return fn(ctx, *args)
    return wrapped


def _require(t: torch.Tensor, mb=None):
    """Tensors on a ROCm device, and on the SAME device as the packed operators they are used with (an operator cache hit or a
    caller-built MeshBatch from another GPU would otherwise be dereferenced on the wrong device)."""
    _hip.require_device(t)
    if mb is not None and mb.device is not None and torch.device(mb.device) != t.device:
        raise RuntimeError("MeshBatch lives on %s but the features are on %s" % (mb.device, t.device))


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError("diffusion_net HIP ops are fp32 (got %s)" % t.dtype)
    return t.contiguous()


def _ws(mb: MeshBatch, nbytes: int):
    buf = _hip.workspace(mb.device, max(int(nbytes), 1024))
    return buf, buf.numel()


# ----------------------------------------------------------------------------------------------
# basis transforms (geometry.py:572-598)
# ----------------------------------------------------------------------------------------------
def _to_basis_raw(mb, x, use_mass):
    L = _hip.lib()
    Cc = x.shape[1]
    spec = torch.empty(mb.n_mesh, mb.k_eig, Cc, dtype=torch.float32, device=x.device)
    ws, n = _ws(mb, L.dn_to_basis_workspace_bytes(mb.ref(), Cc))
    _hip.check(L.dn_to_basis_f32(mb.ref(), x.data_ptr(), Cc, int(use_mass), spec.data_ptr(), ws.data_ptr(), n,
                                 _hip.stream_of(x)), "dn_to_basis_f32")
    return spec


def _from_basis_raw(mb, spec, scale_by_mass=False):
    L = _hip.lib()
    Cc = spec.shape[-1]
    out = torch.empty(mb.v_total, Cc, dtype=torch.float32, device=spec.device)
    _hip.check(L.dn_from_basis_f32(mb.ref(), spec.data_ptr(), Cc, int(scale_by_mass), out.data_ptr(),
                                   _hip.stream_of(spec)), "dn_from_basis_f32")
    return out


class ToBasisFn(torch.autograd.Function):
    """spec[m] = evecs_m^T (x_m * mass_m)  -> [n_mesh, K, C]"""

    @staticmethod
    @_on_device
    def forward(ctx, x, mb):
        _require(x, mb)
        ctx.mb = mb
        return _to_basis_raw(mb, _f32c(x), True)

    @staticmethod
    @_on_device
    def backward(ctx, d_spec):
        return _from_basis_raw(ctx.mb, _f32c(d_spec), scale_by_mass=True), None


class FromBasisFn(torch.autograd.Function):
    """x_m = evecs_m spec[m]  -> [v_total, C]"""

    @staticmethod
    @_on_device
    def forward(ctx, spec, mb):
        _require(spec, mb)
        ctx.mb = mb
        return _from_basis_raw(mb, _f32c(spec))

    @staticmethod
    @_on_device
    def backward(ctx, d_x):
        return _to_basis_raw(ctx.mb, _f32c(d_x), False), None


# ----------------------------------------------------------------------------------------------
# learned-time spectral diffusion (layers.py:44-67)
# ----------------------------------------------------------------------------------------------
class DiffusionFn(torch.autograd.Function):
    @staticmethod
    @_on_device
    def forward(ctx, x, time, mb):
        _require(x, mb)
        L = _hip.lib()
        x, time = _f32c(x), _f32c(time)
        Cc = x.shape[1]
        xs = torch.empty(mb.n_mesh, mb.k_eig, Cc, dtype=torch.float32, device=x.device)
        xd = torch.empty_like(x)
        ws, n = _ws(mb, L.dn_diffusion_workspace_bytes(mb.ref(), Cc))
        _hip.check(L.dn_diffusion_fwd_f32(mb.ref(), x.data_ptr(), time.data_ptr(), Cc, xs.data_ptr(), xd.data_ptr(),
                                          ws.data_ptr(), n, _hip.stream_of(x)), "dn_diffusion_fwd_f32")
        ctx.mb = mb
        ctx.save_for_backward(xs, time)
        return xd

    @staticmethod
    @_on_device
    def backward(ctx, d_xd):
        L = _hip.lib()
        mb = ctx.mb
        xs, time = ctx.saved_tensors
        d_xd = _f32c(d_xd)
        Cc = d_xd.shape[1]
        d_x = torch.empty_like(d_xd)
        d_t = torch.empty_like(time)
        ws, n = _ws(mb, L.dn_diffusion_workspace_bytes(mb.ref(), Cc))
        _hip.check(L.dn_diffusion_bwd_f32(mb.ref(), d_xd.data_ptr(), xs.data_ptr(), time.data_ptr(), Cc, None,
                                          d_x.data_ptr(), d_t.data_ptr(), ws.data_ptr(), n, _hip.stream_of(d_xd)),
                   "dn_diffusion_bwd_f32")
        return d_x, d_t, None


# ----------------------------------------------------------------------------------------------
# spatial gradients + gradient features (layers.py:217-223, 117-130)
# ----------------------------------------------------------------------------------------------
class GradApplyFn(torch.autograd.Function):
    @staticmethod
    @_on_device
    def forward(ctx, x, mb):
        _require(x, mb)
        x = _f32c(x)
        gx, gy = torch.empty_like(x), torch.empty_like(x)
        _hip.check(_hip.lib().dn_grad_apply_fwd_f32(mb.ref(), x.data_ptr(), x.shape[1], gx.data_ptr(), gy.data_ptr(),
                                                    _hip.stream_of(x)), "dn_grad_apply_fwd_f32")
        ctx.mb = mb
        return gx, gy

    @staticmethod
    @_on_device
    def backward(ctx, d_gx, d_gy):
        d_gx, d_gy = _f32c(d_gx), _f32c(d_gy)
        d_x = torch.empty_like(d_gx)
        _hip.check(_hip.lib().dn_grad_apply_bwd_f32(ctx.mb.ref(), d_gx.data_ptr(), d_gy.data_ptr(), None, d_gx.shape[1],
                                                    d_x.data_ptr(), _hip.stream_of(d_gx)), "dn_grad_apply_bwd_f32")
        return d_x, None


class GradFeatFn(torch.autograd.Function):
    @staticmethod
    @_on_device
    def forward(ctx, gx, gy, A_re, A_im, mb):
        _require(gx, mb)
        gx, gy, A_re = _f32c(gx), _f32c(gy), _f32c(A_re)
        A_im = _f32c(A_im) if A_im is not None else None
        g, bre, bim = torch.empty_like(gx), torch.empty_like(gx), torch.empty_like(gx)
        _hip.check(_hip.lib().dn_gradfeat_fwd_f32(mb.ref(), gx.data_ptr(), gy.data_ptr(), A_re.data_ptr(), _hip.ptr(A_im),
                                                  gx.shape[1], g.data_ptr(), bre.data_ptr(), bim.data_ptr(),
                                                  _hip.stream_of(gx)), "dn_gradfeat_fwd_f32")
        ctx.mb = mb
        ctx.has_im = A_im is not None
        ctx.save_for_backward(gx, gy, g, bre, bim, A_re, A_im if A_im is not None else A_re)
        return g

    @staticmethod
    @_on_device
    def backward(ctx, d_g):
        L = _hip.lib()
        mb = ctx.mb
        gx, gy, g, bre, bim, A_re, A_im = ctx.saved_tensors
        d_g = _f32c(d_g)
        Cc = gx.shape[1]
        d_gx, d_gy = torch.empty_like(gx), torch.empty_like(gx)
        dA_re = torch.empty_like(A_re)
        dA_im = torch.empty_like(A_re) if ctx.has_im else None
        ws, n = _ws(mb, L.dn_gradfeat_workspace_bytes(mb.ref(), Cc))
        _hip.check(L.dn_gradfeat_bwd_f32(mb.ref(), d_g.data_ptr(), g.data_ptr(), gx.data_ptr(), gy.data_ptr(), bre.data_ptr(),
                                         bim.data_ptr(), A_re.data_ptr(), A_im.data_ptr() if ctx.has_im else None, Cc,
                                         d_gx.data_ptr(), d_gy.data_ptr(), dA_re.data_ptr(), _hip.ptr(dA_im),
                                         ws.data_ptr(), n, _hip.stream_of(d_g)), "dn_gradfeat_bwd_f32")
        return d_gx, d_gy, dA_re, dA_im, None


# ----------------------------------------------------------------------------------------------
# nn.Linear on the row axis (first_lin / last_lin / stand-alone MiniMLP layers)
# ----------------------------------------------------------------------------------------------
def hks(evals, evecs, scales):
    """out[.., v, s] = sum_k exp(-evals[.., k] * scales[.., s]) * evecs[.., v, k]^2 (geometry.py:600-628), forward only."""
    _hip.require_device(evecs)
    squeeze = evals.dim() == 1
    if squeeze:
        evals, evecs, scales = evals[None], evecs[None], scales[None]
    evals, evecs, scales = _f32c(evals), _f32c(evecs), _f32c(scales)
    B, V, K = evecs.shape
    S = scales.shape[-1]
    out = torch.empty(B, V, S, dtype=torch.float32, device=evecs.device)
    per_batch = int(scales.dim() == 2 and scales.shape[0] == B and B > 1)
    _hip.check(_hip.lib().dn_hks_f32(evals.data_ptr(), evecs.data_ptr(), scales.data_ptr(), B, V, K, S, per_batch, out.data_ptr(),
                                     _hip.stream_of(evecs)), "dn_hks_f32")
    return out[0] if squeeze else out


# ----------------------------------------------------------------------------------------------
# gradient sinks: a parameter may carry ``_dn_grad_sink`` (set by dist.FlatParams: its slice of the flat
# gradient bucket).  Backward then ACCUMULATES the parameter gradients of a whole op into their sinks with
# one multi-tensor add and returns None for them, instead of handing ~10 small tensors per block to
# autograd's AccumulateGrad (one 5 us add kernel each: 40 launches per step of the 4-block network).
# ----------------------------------------------------------------------------------------------
def _sinks(params):
    """The sink of a parameter counts only while the parameter still owns it: it requires grad and its ``.grad`` IS the sink
    (``zero_grad(set_to_none=True)`` or a frozen parameter switch back to returning the gradient to autograd)."""
    out = []
    for p in params:
        s = getattr(p, "_dn_grad_sink", None) if p is not None else None
        if s is not None and not (p.requires_grad and p.grad is not None and p.grad.data_ptr() == s.data_ptr()):
            s = None
        out.append(s)
    return out


def _grad_out(sink, like):
    """Output buffer for a parameter gradient.  A sink that is known to hold zeros (``dist.FlatParams.zero_grad`` marks it fresh) is
    handed to the kernel itself: its plain store IS the accumulation (0 + g), and the add launch of ``_deliver`` disappears.  The
    mark is consumed, so a second gradient for the same parameter before the next zero_grad (shared weights, several backward
    passes) goes through a temporary and is added."""
    if like is None:
        return None
    if sink is not None and getattr(sink, "_dn_fresh", False) and sink.shape == like.shape and sink.is_contiguous() \
            and sink.dtype == torch.float32 and sink.data_ptr() % 16 == 0:
        sink._dn_fresh = False
        return sink
    return torch.empty_like(like, dtype=torch.float32)


def _deliver(sinks, grads):
    pairs = [(s, g) for s, g in zip(sinks, grads) if s is not None and g is not None and g is not s]
    if pairs:
        torch._foreach_add_([s for s, g in pairs], [g for s, g in pairs])
    return [g if s is None else None for s, g in zip(sinks, grads)]


class LinearFn(torch.autograd.Function):
    @staticmethod
    @_on_device
    def forward(ctx, x, W, b, mb):
        _require(x, mb)
        ctx.sinks = _sinks([W, b])
        x, W, b = _f32c(x), _f32c(W), _f32c(b)
        out = torch.empty(x.shape[0], W.shape[0], dtype=torch.float32, device=x.device)
        # first_lin (thin input -> block width): its kernel leaves max |out| in a device word for block 0's split-fp16 engine for free
        word = torch.zeros(1, dtype=torch.float32, device=x.device) if (W.shape[1] <= 16 and W.shape[0] >= 128 and _amax_tags_on()) else None
        _hip.check(_hip.lib().dn_linear_fwd_amax_f32(mb.ref(), x.data_ptr(), W.shape[1], W.data_ptr(), b.data_ptr(), W.shape[0],
                                                     0, None, out.data_ptr(), _hip.ptr(word), _hip.stream_of(x)), "dn_linear_fwd_amax_f32")
        ctx.mb = mb
        ctx.save_for_backward(x, W)
        if word is not None:
            _tag_amax(out, word)
        return out

    @staticmethod
    @_on_device
    def backward(ctx, d_out):
        L = _hip.lib()
        mb = ctx.mb
        x, W = ctx.saved_tensors
        d_out = _f32c(d_out)
        C_out, C_in = W.shape
        d_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = _grad_out(ctx.sinks[0], W)
        db = _grad_out(ctx.sinks[1], W.new_empty(C_out))
        ws, n = _ws(mb, L.dn_linear_workspace_bytes(mb.ref(), C_in, C_out))
        # last_lin (block width -> a few classes): the input gradient's magnitude for the last block's backward, likewise for free
        word = torch.zeros(1, dtype=torch.float32, device=x.device) if (d_x is not None and C_out <= 16 and C_in >= 128 and _amax_tags_on()) else None
        _hip.check(L.dn_linear_bwd_amax_f32(mb.ref(), d_out.data_ptr(), x.data_ptr(), W.data_ptr(), C_in, C_out, _hip.ptr(d_x),
                                            dW.data_ptr(), db.data_ptr(), _hip.ptr(word), ws.data_ptr(), n, _hip.stream_of(d_out)),
                   "dn_linear_bwd_amax_f32")
        dW, db = _deliver(ctx.sinks, [dW, db])
        if word is not None:
            _tag_amax(d_x, word)
        return d_x, dW, db, None


# ----------------------------------------------------------------------------------------------
# fused DiffusionNetBlock (layers.py:200-241)
# ----------------------------------------------------------------------------------------------
class BlockConfig:
    """Static description of one block: channel width, MiniMLP sizes, feature switches."""

    def __init__(self, C, widths, with_grad, with_rot):
        self.C, self.widths, self.with_grad, self.with_rot = int(C), [int(w) for w in widths], bool(with_grad), bool(with_rot)
        self.n_mlp = len(self.widths) - 1
        self.handle = register_handle(self)
        self.grad_hook = None        # dist.FlatParams: called after the block's gradients were delivered
        self.grad_pre_hook = None    # ... and before they are written
        self.clamp_time = False      # True: the block forward clamps diffusion_time to >= 1e-8 in place inside its first launch (layers.py:48-49)
        self.flags = 0               # per-call engine choice (include/diffnet_hip.h: DN_BLOCK_*), 0 = follow the library's option table
        if self.n_mlp > _hip.MAX_MLP:
            raise ValueError("MiniMLP deeper than %d layers is not supported by the HIP block" % _hip.MAX_MLP)


def _params_struct(cfg: BlockConfig, time, A_re, A_im, Ws, bs, masks, x_amax=None, out_amax=None):
    p = _hip.BlockParamsStruct()
    p.x_amax, p.out_amax = _hip.ptr(x_amax), _hip.ptr(out_amax)
    p.C, p.n_mlp, p.with_grad, p.with_rot = cfg.C, cfg.n_mlp, int(cfg.with_grad), int(cfg.with_rot)
    p.flags = int(getattr(cfg, "flags", 0))
    for i, w in enumerate(cfg.widths):
        p.widths[i] = w
    p.time, p.A_re, p.A_im = time.data_ptr(), _hip.ptr(A_re), _hip.ptr(A_im)
    dev_seed = None
    if isinstance(masks, tuple):      # (host seed, device seed word): in-kernel dropout under a captured graph
        masks, dev_seed = masks
    seeded = isinstance(masks, int)   # in-kernel dropout: `masks` is the 64-bit seed
    p.drop_seed = masks if seeded else 0
    p.drop_seed_dev = dev_seed.data_ptr() if dev_seed is not None else None
    for i in range(cfg.n_mlp):
        p.W[i], p.b[i] = Ws[i].data_ptr(), bs[i].data_ptr()
        p.mask[i] = _hip.ptr(masks[i]) if (masks is not None and not seeded) else None
    return p


def keep_mask_reference(seed: int, layer: int, n_rows: int, width: int):
    """numpy restatement of the in-kernel dropout bits (dn_common.h: dn_keep_bits, dn_api.hip: layer_seed) -- what
    ``mask[layer]`` would have to be for the explicit-mask path to reproduce the seeded one.  Test infrastructure."""
    import numpy as np
    M64, M32 = (1 << 64) - 1, np.uint64(0xFFFFFFFF)
    s = (seed + 0x9E3779B97F4A7C15 * (layer + 1)) & M64
    s = s or 0x9E3779B97F4A7C15
    groups = (width + 3) // 4
    rows = np.arange(n_rows, dtype=np.uint64)[:, None]
    c4 = np.arange(groups, dtype=np.uint64)[None, :]
    x = ((rows * np.uint64(groups) + c4) & M32) ^ np.uint64(s & 0xFFFFFFFF)
    x = (x + np.uint64(((s >> 32) * 0x9E3779B9) & 0xFFFFFFFF)) & M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & M32
    x ^= x >> np.uint64(16)
    bits = (x >> np.uint64(28)).astype(np.uint8)
    keep = np.stack([(bits >> e) & 1 for e in range(4)], axis=-1).reshape(n_rows, groups * 4)[:, :width]
    return torch.from_numpy(np.ascontiguousarray(keep))


def _tag_amax(t, word):
    """Attach the device word holding max |t| to the tensor object (a plain attribute: it lives and dies with that object, so it can
    never describe other data).  Autograd hands the same object on when a gradient has a single consumer; otherwise the tag is simply
    absent and the consumer measures the tensor itself."""
    try:
        t._dn_amax = (word, t.data_ptr(), t._version)
    except Exception:      # noqa: BLE001
        pass


def _amax_tags_on():
    return not os.environ.get("DN_NO_AMAX_TAGS")      # diagnostic: every block measures its input itself


def _amax_of(t):
    if not _amax_tags_on():
        return None
    tag = getattr(t, "_dn_amax", None)
    if tag is None or tag[1] != t.data_ptr() or tag[2] != t._version or tag[0].device != t.device:
        return None
    return tag[0]


# Debug hook for the parity tests (off = None): a list that receives, per BlockFn.forward call with gradients wanted, a dict of the
# activations the call saved for its backward, BY NAME -- {"xs", "xd", "gx", "gy", "g", "bre", "bim", "h": [post-ReLU hidden activations]}
# (the flip-aware gradient criterion of tests/parity_cases.py needs the hidden activations the HIP forward actually produced).
debug_saved = None


class BlockFn(torch.autograd.Function):
    """forward(x, time, A_re, A_im, *W_and_b) with non-tensor (mb, cfg, masks) first."""

    @staticmethod
    @_on_device
    def forward(ctx, mb, cfg, masks, x, time, A_re, A_im, *wb):
        _require(x, mb)
        ctx.sinks = _sinks([time, A_re, A_im, *wb])
        L = _hip.lib()
        time_param = time
        x, time = _f32c(x), _f32c(time)
        # (the torch.library form of this op must not write to an input it does not declare as mutated: layers.py clamps before it, torchlib.py
        # sets ctx.no_clamp)
        clamp_in_call = bool(getattr(cfg, "clamp_time", False)) and not getattr(ctx, "no_clamp", False)
        if clamp_in_call and time.data_ptr() != time_param.data_ptr():     # a converted copy: the library would clamp the copy, not the Parameter
            time_param.data.clamp_(min=1e-8)
            time, clamp_in_call = _f32c(time_param), False
        A_re = _f32c(A_re) if A_re is not None else None
        A_im = _f32c(A_im) if A_im is not None else None
        Ws = [_f32c(w) for w in wb[0::2]]
        bs = [_f32c(b) for b in wb[1::2]]
        dev, V, Cc = x.device, x.shape[0], cfg.C
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        # magnitude words of the split-fp16 engine (dn_api.hip): the input's travels with the tensor from the block that produced it,
        # this block's output gets one for the next block; a tensor without one is measured by the library (one extra pass)
        x_amax = _amax_of(x)
        # [saved-activation words | max |out|].  Uninitialised on purpose (a fill per block and step is a launch): a call that tracks
        # magnitudes zeroes and writes them itself; one that does not leaves them alone, and then nothing reads them -- the tag below and the
        # backward's use are both conditional on dn_block_tracks_amax
        words = new(_hip.BLOCK_AMAX_WORDS + 1)
        out_amax = words[_hip.BLOCK_AMAX_WORDS:]
        p = _params_struct(cfg, time, A_re, A_im, Ws, bs, masks, x_amax, out_amax)
        p.clamp_time = 1 if clamp_in_call else 0
        out = new(V, Cc)
        # ctx.needs_input_grad is True for every Parameter even under torch.no_grad() (torch 2.10), and grad mode is always off inside a
        # Function's forward: the caller samples torch.is_grad_enabled() and leaves it on the config (layers.forward_packed).  Until round 6
        # every "inference" call through the package ran the TRAINING forward (eight saved arrays of [V, C] written per block: +11 % at
        # BASELINE config 4's shape) -- and with it never took the inference-only kernel choices.
        need_grad = any(ctx.needs_input_grad) and bool(getattr(cfg, "_grad_enabled", True))
        if need_grad:
            sv = _hip.BlockSavedStruct()
            xs, xd = new(mb.n_mesh, mb.k_eig, Cc), new(V, Cc)
            feats = [new(V, Cc) for _ in range(5)] if cfg.with_grad else []
            hs = [new(V, cfg.widths[i + 1]) for i in range(cfg.n_mlp - 1)]
            sv.xs, sv.xd = xs.data_ptr(), xd.data_ptr()
            if cfg.with_grad:
                sv.gx, sv.gy, sv.g, sv.bre, sv.bim = (t.data_ptr() for t in feats)
            for i, h in enumerate(hs):
                sv.h[i] = h.data_ptr()
            sv.amax = words.data_ptr()
            sv_ref = C.byref(sv)
        else:
            sv_ref = None
        ws, n = _ws(mb, L.dn_block_fwd_workspace_bytes(mb.ref(), C.byref(p), int(need_grad)))
        _hip.check(L.dn_block_fwd_f32(mb.ref(), C.byref(p), x.data_ptr(), out.data_ptr(), sv_ref, ws.data_ptr(), n,
                                      _hip.stream_of(x)), "dn_block_fwd_f32")
        if need_grad:
            ctx.mb, ctx.cfg, ctx.masks = mb, cfg, masks
            ctx.n_feat, ctx.n_h = len(feats), len(hs)
            ctx.has = (A_re is not None, A_im is not None)
            ctx.save_for_backward(x, time, xs, xd, words, *feats, *hs, *Ws, *bs,
                                  *([A_re] if A_re is not None else []), *([A_im] if A_im is not None else []))
            if debug_saved is not None:
                debug_saved.append({"xs": xs, "xd": xd, **dict(zip(("gx", "gy", "g", "bre", "bim"), feats)), "h": list(hs), "words": words})
        if L.dn_block_tracks_amax(mb.ref(), C.byref(p), 1 if need_grad else 0):
            _tag_amax(out, out_amax)  # the next block reads it off its input (a call that does not track magnitudes leaves the word alone: no tag)
        return out

    @staticmethod
    @_on_device
    def backward(ctx, d_out):
        L = _hip.lib()
        mb, cfg, masks = ctx.mb, ctx.cfg, ctx.masks
        pre = getattr(cfg, "grad_pre_hook", None)
        if pre is not None:
            pre()
        sav = list(ctx.saved_tensors)
        x, time, xs, xd, words = sav[:5]
        pos = 5
        feats = sav[pos:pos + ctx.n_feat]; pos += ctx.n_feat
        hs = sav[pos:pos + ctx.n_h]; pos += ctx.n_h
        Ws = sav[pos:pos + cfg.n_mlp]; pos += cfg.n_mlp
        bs = sav[pos:pos + cfg.n_mlp]; pos += cfg.n_mlp
        A_re = A_im = None
        if ctx.has[0]:
            A_re = sav[pos]; pos += 1
        if ctx.has[1]:
            A_im = sav[pos]; pos += 1
        d_out = _f32c(d_out)
        p = _params_struct(cfg, time, A_re, A_im, Ws, bs, masks)
        sv = _hip.BlockSavedStruct()
        sv.xs, sv.xd = xs.data_ptr(), xd.data_ptr()
        if cfg.with_grad:
            sv.gx, sv.gy, sv.g, sv.bre, sv.bim = (t.data_ptr() for t in feats)
        for i, h in enumerate(hs):
            sv.h[i] = h.data_ptr()
        sv.amax = words.data_ptr()
        gr = _hip.BlockGradsStruct()
        sk = ctx.sinks                # order: time, A_re, A_im, W0, b0, W1, b1, ...
        d_x, d_time = torch.empty_like(x), _grad_out(sk[0], time)
        dA_re, dA_im = _grad_out(sk[1], A_re), _grad_out(sk[2], A_im)
        dWs = [_grad_out(sk[3 + 2 * i], w) for i, w in enumerate(Ws)]
        dbs = [_grad_out(sk[4 + 2 * i], b) for i, b in enumerate(bs)]
        gr.d_x, gr.d_time, gr.dA_re, gr.dA_im = d_x.data_ptr(), d_time.data_ptr(), _hip.ptr(dA_re), _hip.ptr(dA_im)
        dx_amax = torch.empty(1, dtype=torch.float32, device=d_x.device)      # (written only, and tagged only, when the call tracks magnitudes)
        gr.d_out_amax, gr.d_x_amax = _hip.ptr(_amax_of(d_out)), dx_amax.data_ptr()
        for i in range(cfg.n_mlp):
            gr.dW[i], gr.db[i] = dWs[i].data_ptr(), dbs[i].data_ptr()
        ws, n = _ws(mb, L.dn_block_bwd_workspace_bytes(mb.ref(), C.byref(p)))
        _hip.check(L.dn_block_bwd_f32(mb.ref(), C.byref(p), x.data_ptr(), C.byref(sv), d_out.data_ptr(), C.byref(gr),
                                      ws.data_ptr(), n, _hip.stream_of(d_out)), "dn_block_bwd_f32")
        wb = []
        for dw, db in zip(dWs, dbs):
            wb += [dw, db]
        d_time, dA_re, dA_im, *wb = _deliver(ctx.sinks, [d_time, dA_re, dA_im, *wb])
        hook = getattr(cfg, "grad_hook", None)
        if hook is not None:          # dist.FlatParams: this block's gradient range may go out now
            hook()
        if L.dn_block_tracks_amax(mb.ref(), C.byref(p), 2):
            _tag_amax(d_x, dx_amax)   # the block before this one finds it on the gradient it receives
        return (None, None, None, d_x, d_time, dA_re, dA_im, *wb)


# ----------------------------------------------------------------------------------------------
# output remaps (layers.py:379-397)
# ----------------------------------------------------------------------------------------------
class GatherMeanFn(torch.autograd.Function):
    """out[i] = mean_j x[index[i, j]]  (faces: 3 rows, edges: 2 rows)"""

    @staticmethod
    @_on_device
    def forward(ctx, x, pat: GatherPattern):
        _require(x)
        x = _f32c(x)
        out = torch.empty(pat.n_out, x.shape[1], dtype=torch.float32, device=x.device)
        _hip.check(_hip.lib().dn_csr_mean_f32(pat.rowptr.data_ptr(), pat.col.data_ptr(), pat.n_out, x.data_ptr(), x.shape[1],
                                              float(pat.n_per), out.data_ptr(), _hip.stream_of(x)), "dn_csr_mean_f32")
        ctx.pat = pat
        return out

    @staticmethod
    @_on_device
    def backward(ctx, d_out):
        pat = ctx.pat
        d_out = _f32c(d_out)
        d_x = torch.empty(pat.n_src, d_out.shape[1], dtype=torch.float32, device=d_out.device)
        _hip.check(_hip.lib().dn_csr_mean_f32(pat.t_rowptr.data_ptr(), pat.t_col.data_ptr(), pat.n_src, d_out.data_ptr(),
                                              d_out.shape[1], float(pat.n_per), d_x.data_ptr(), _hip.stream_of(d_out)),
                   "dn_csr_mean_f32")
        return d_x, None


class MassMeanFn(torch.autograd.Function):
    """out[m] = sum_v mass_v x_v / sum_v mass_v over the vertices of mesh m (layers.py:397)"""

    @staticmethod
    @_on_device
    def forward(ctx, x, mb):
        _require(x, mb)
        x = _f32c(x)
        out = torch.empty(mb.n_mesh, x.shape[1], dtype=torch.float32, device=x.device)
        msum = torch.empty(mb.n_mesh, dtype=torch.float32, device=x.device)
        _hip.check(_hip.lib().dn_mass_mean_fwd_f32(mb.ref(), x.data_ptr(), x.shape[1], out.data_ptr(), msum.data_ptr(),
                                                   _hip.stream_of(x)), "dn_mass_mean_fwd_f32")
        ctx.mb = mb
        ctx.save_for_backward(msum)
        return out

    @staticmethod
    @_on_device
    def backward(ctx, d_out):
        (msum,) = ctx.saved_tensors
        d_out = _f32c(d_out)
        d_x = torch.empty(ctx.mb.v_total, d_out.shape[1], dtype=torch.float32, device=d_out.device)
        _hip.check(_hip.lib().dn_mass_mean_bwd_f32(ctx.mb.ref(), msum.data_ptr(), d_out.data_ptr(), d_out.shape[1],
                                                   d_x.data_ptr(), _hip.stream_of(d_out)), "dn_mass_mean_bwd_f32")
        return d_x, None


# ----------------------------------------------------------------------------------------------
# the head on the far side of the path (dn_head.hip): [gather-mean] -> [log_softmax] -> log-probabilities -> [NLL / smoothed loss]
# ----------------------------------------------------------------------------------------------
class HeadFn(torch.autograd.Function):
    """forward(x, pat, labels, log_softmax, smoothing, want_logp) -> (logp or None, loss or None)

    x: [n_src, C] logits (log-probabilities when ``log_softmax`` is False); pat: GatherPattern of the faces/edges remap or None
    (output i = row i); labels: int64 [n_out] or None.  One kernel forward (+ a 1-block finish when there is a loss), one kernel
    backward for any combination of incoming d_logp / d_loss."""

    @staticmethod
    @_on_device
    def forward(ctx, x, pat, labels, log_softmax, smoothing, want_logp):
        _require(x)
        L = _hip.lib()
        x = _f32c(x)
        n_src, Cc = x.shape
        if Cc > _hip.HEAD_MAX_CLASSES:
            raise ValueError("the fused head handles up to %d classes (got %d); DiffusionNet.forward_packed composes the remap and the "
                             "activation from separate ops above that" % (_hip.HEAD_MAX_CLASSES, Cc))
        n_out = pat.n_out if pat is not None else n_src
        if labels is not None:
            if labels.dtype != torch.int64:
                raise TypeError("labels must be int64")
            labels = labels.reshape(-1).contiguous()
            if labels.numel() != n_out:
                raise ValueError("expected %d labels, got %d" % (n_out, labels.numel()))
        need_logp = want_logp or (log_softmax and ctx.needs_input_grad[0])   # the backward of log_softmax needs the probabilities
        logp = torch.empty(n_out, Cc, dtype=torch.float32, device=x.device) if need_logp else None
        loss = torch.empty((), dtype=torch.float32, device=x.device) if labels is not None else None
        count = torch.empty((), dtype=torch.float32, device=x.device) if labels is not None else None
        ws = _hip.workspace(x.device, L.dn_head_workspace_bytes())
        _hip.check(L.dn_head_fwd_f32(x.data_ptr(), n_src, Cc, _hip.ptr(pat.rowptr) if pat is not None else None,
                                     _hip.ptr(pat.col) if pat is not None else None, n_out, float(pat.n_per) if pat is not None else 1.0,
                                     int(bool(log_softmax)), _hip.ptr(labels), float(smoothing), _hip.ptr(logp), _hip.ptr(loss), _hip.ptr(count),
                                     ws.data_ptr(), ws.numel(), _hip.stream_of(x)), "dn_head_fwd_f32")
        ctx.pat, ctx.lsm, ctx.smoothing, ctx.shape = pat, bool(log_softmax), float(smoothing), (n_src, n_out, Cc)
        ctx.save_for_backward(*[t for t in (logp, labels, count) if t is not None])
        ctx.has = (logp is not None, labels is not None)
        return (logp if want_logp else None), loss

    @staticmethod
    @_on_device
    def backward(ctx, d_logp, d_loss):
        sav = list(ctx.saved_tensors)
        logp = sav.pop(0) if ctx.has[0] else None
        labels = sav.pop(0) if ctx.has[1] else None
        count = sav.pop(0) if ctx.has[1] else None
        n_src, n_out, Cc = ctx.shape
        pat = ctx.pat
        ref = d_logp if d_logp is not None else d_loss
        d_x = torch.empty(n_src, Cc, dtype=torch.float32, device=ref.device)
        d_logp = _f32c(d_logp) if d_logp is not None else None
        d_loss = _f32c(d_loss) if d_loss is not None else None
        _hip.check(_hip.lib().dn_head_bwd_f32(_hip.ptr(logp), n_out, Cc, _hip.ptr(pat.t_rowptr) if pat is not None else None,
                                              _hip.ptr(pat.t_col) if pat is not None else None, n_src, float(pat.n_per) if pat is not None else 1.0,
                                              int(ctx.lsm), _hip.ptr(labels) if d_loss is not None else None, ctx.smoothing, _hip.ptr(d_logp),
                                              _hip.ptr(d_loss), _hip.ptr(count), d_x.data_ptr(), _hip.stream_of(ref)), "dn_head_bwd_f32")
        return d_x, None, None, None, None, None
