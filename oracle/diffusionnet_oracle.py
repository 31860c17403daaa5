"""CPU oracle for the DiffusionNet forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement (torch-CPU tensor
algebra, fp32 or fp64) of the algorithm in the reference's
``src/diffusion_net/layers.py`` and the two basis transforms of
``src/diffusion_net/geometry.py``.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the
checker / the timed CPU baseline -- never from the product package
(``diffusion-net_amd/``), which must fail loudly when the HIP library is missing.

Parity pin: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, imported read-only in the dev container by
``tests/golden/make_golden.py`` and committed as ``tests/golden/*.npz``
(``tests/test_oracle_golden.py`` checks every one of them).

The restatement is *functional*: weights come in as a flat ``dict`` keyed by
the reference's ``state_dict`` names, every function is stateless, gradients
come from torch autograd exactly as in the reference (which has no hand-written
backward either).  Dropout is made reproducible by passing the keep-masks in.

Reference anchors (file:line under /root/reference/src/diffusion_net):
  to_basis / from_basis ............ geometry.py:572-583, 586-598
  learned-time spectral diffusion .. layers.py:44-67
  spatial gradient apply ........... layers.py:213-223
  gradient features ................ layers.py:117-130
  MiniMLP (dropout placement) ...... layers.py:137-164
  block wiring + residual .......... layers.py:200-241
  net: batch dim, head/tail, remap . layers.py:342-407
"""
from __future__ import annotations

import torch

MIN_DIFFUSION_TIME = 1e-8  # layers.py:49


# --------------------------------------------------------------------------
# basis transforms (geometry.py:572-598)
# --------------------------------------------------------------------------
def to_basis(values, basis, massvec):
    """(B,V,C),(B,V,K),(B,V) -> (B,K,C): mass-weighted projection, geometry.py:582-583."""
    weighted = values * massvec[..., None]
    return torch.einsum("bvk,bvc->bkc", basis, weighted)


def from_basis(coefs, basis):
    """(B,K,C),(B,V,K) -> (B,V,C), geometry.py:598 (real branch only; the complex
    branch at :595-596 calls helpers that do not exist in the reference)."""
    return torch.einsum("bvk,bkc->bvc", basis, coefs)


# --------------------------------------------------------------------------
# LearnedTimeDiffusion, method='spectral' (layers.py:44-67)
# --------------------------------------------------------------------------
def clamp_time(t):
    """layers.py:48-49: projection of the learned times onto [1e-8, inf)."""
    return torch.clamp(t, min=MIN_DIFFUSION_TIME)


def spectral_diffusion(x, mass, evals, evecs, time):
    """x:(B,V,C) mass:(B,V) evals:(B,K) evecs:(B,V,K) time:(C) -> (B,V,C).

    ``time`` must already be clamped (the reference clamps the Parameter in
    place before using it, layers.py:48-49,62)."""
    spec = to_basis(x, evecs, mass)                                   # layers.py:59
    decay = torch.exp(-evals[..., None] * time[None, None, :])        # layers.py:63
    return from_basis(decay * spec, evecs)                            # layers.py:64-67


# --------------------------------------------------------------------------
# gradient apply + SpatialGradientFeatures (layers.py:213-223, 117-130)
# --------------------------------------------------------------------------
def _sparse_item(sp, b):
    """b-th [V,V] operator of a batched sparse input (list, or 3-D sparse COO as the
    reference builds with unsqueeze(0), layers.py:355-356)."""
    if isinstance(sp, (list, tuple)):
        return sp[b]
    if sp.dim() == 3:
        return sp[b]
    return sp


def gradient_apply(xd, gradX, gradY):
    """xd:(B,V,C) -> (gx, gy) each (B,V,C): per-item sparse@dense, layers.py:217-220."""
    gxs, gys = [], []
    for b in range(xd.shape[0]):
        gxs.append(torch.mm(_sparse_item(gradX, b), xd[b]))
        gys.append(torch.mm(_sparse_item(gradY, b), xd[b]))
    return torch.stack(gxs, 0), torch.stack(gys, 0)


def gradient_features(gx, gy, A_re=None, A_im=None, A=None):
    """tanh of the real inner product <g, rot(g)>, layers.py:121-130.
    Linear layers are bias-free, y = x @ W.T (layers.py:110-113)."""
    if A is None:
        b_re = gx @ A_re.T - gy @ A_im.T        # layers.py:122
        b_im = gy @ A_re.T + gx @ A_im.T        # layers.py:123
    else:
        b_re = gx @ A.T                         # layers.py:125
        b_im = gy @ A.T                         # layers.py:126
    return torch.tanh(gx * b_re + gy * b_im)    # layers.py:128-130


# --------------------------------------------------------------------------
# MiniMLP (layers.py:137-164)
# --------------------------------------------------------------------------
def mini_mlp(h, weights, biases, keep_masks=None, act_pattern=None, capture=None):
    """Linear stack with ReLU between layers and dropout(p=.5) in front of every
    layer but the first (layers.py:143-147); nothing after the last (layers.py:160).

    keep_masks: None (eval / dropout off) or a list with one {0,1} mask per
    layer i>0, shaped like that layer's input; kept values are scaled by 2 as
    nn.Dropout(p=.5) does.
    Test diagnostics (no reference counterpart): ``capture`` -- a list that receives the pre-activations of every hidden layer;
    ``act_pattern`` -- one boolean tensor per hidden layer: the ReLU is replaced by "pass where True, zero where False", i.e. the
    network is evaluated with a GIVEN activation pattern (a ReLU net is piecewise linear: its gradient is only defined per pattern,
    and a unit whose pre-activation lies within rounding distance of zero may land on either side in any finite-precision
    evaluation, including the reference's own)."""
    n = len(weights)
    for i in range(n):
        if keep_masks is not None and i > 0:
            h = h * keep_masks[i - 1] * 2.0
        h = h @ weights[i].T + biases[i]
        if i + 1 < n:
            if capture is not None:
                capture.append(h.detach())
            h = torch.relu(h) if act_pattern is None else h * act_pattern[i].to(h.dtype)
    return h


# --------------------------------------------------------------------------
# parameter-dict helpers (state_dict naming pinned by the shipped .pth files)
# --------------------------------------------------------------------------
def mlp_params(params, prefix):
    ws, bs, i = [], [], 0
    while f"{prefix}.mlp.miniMLP_mlp_layer_{i:03d}.weight" in params:
        ws.append(params[f"{prefix}.mlp.miniMLP_mlp_layer_{i:03d}.weight"])
        bs.append(params[f"{prefix}.mlp.miniMLP_mlp_layer_{i:03d}.bias"])
        i += 1
    return ws, bs


def count_blocks(params):
    n = 0
    while f"block_{n}.diffusion.diffusion_time" in params:
        n += 1
    return n


# --------------------------------------------------------------------------
# DiffusionNetBlock (layers.py:200-241)
# --------------------------------------------------------------------------
def block_forward(params, prefix, x, mass, evals, evecs, gradX, gradY, keep_masks=None, act_pattern=None, capture=None):
    time = clamp_time(params[f"{prefix}.diffusion.diffusion_time"])
    xd = spectral_diffusion(x, mass, evals, evecs, time)                  # layers.py:210
    feats = [x, xd]
    if f"{prefix}.gradient_features.A_re.weight" in params:
        gx, gy = gradient_apply(xd, gradX, gradY)                         # layers.py:217-223
        feats.append(gradient_features(
            gx, gy,
            A_re=params[f"{prefix}.gradient_features.A_re.weight"],
            A_im=params[f"{prefix}.gradient_features.A_im.weight"]))
    elif f"{prefix}.gradient_features.A.weight" in params:
        gx, gy = gradient_apply(xd, gradX, gradY)
        feats.append(gradient_features(gx, gy, A=params[f"{prefix}.gradient_features.A.weight"]))
    h0 = torch.cat(feats, dim=-1)                                         # layers.py:229/232
    ws, bs = mlp_params(params, prefix)
    return mini_mlp(h0, ws, bs, keep_masks, act_pattern, capture) + x     # layers.py:236-239


# --------------------------------------------------------------------------
# DiffusionNet.forward (layers.py:342-407)
# --------------------------------------------------------------------------
def remap_outputs(x, outputs_at, mass=None, edges=None, faces=None):
    """layers.py:376-397.  x:(B,V,Co)."""
    if outputs_at == "vertices":
        return x
    if outputs_at in ("edges", "faces"):
        idx = edges if outputs_at == "edges" else faces               # (B,E,2) / (B,F,3)
        picked = torch.stack([x[b][idx[b]] for b in range(x.shape[0])], 0)   # (B,E,n,Co)
        return picked.mean(dim=2)
    if outputs_at == "global_mean":
        return (x * mass[..., None]).sum(dim=-2) / mass.sum(dim=-1, keepdim=True)   # layers.py:397
    raise ValueError("invalid setting for outputs_at")                # layers.py:278


def net_forward(params, x_in, mass, evals, evecs, gradX=None, gradY=None, edges=None, faces=None,
                outputs_at="vertices", last_activation=None, keep_masks=None, act_patterns=None, capture=None):
    """Whole-net forward.  ``params``: dict of tensors keyed like the reference's
    state_dict.  Inputs unbatched ([V,..]) or batched ([B,V,..]) as layers.py:346-363;
    gradX/gradY torch sparse COO ([V,V] / [B,V,V]) or a list of [V,V] sparse.
    keep_masks: None or list (one entry per block) of lists of dropout keep-masks."""
    squeeze = False
    if x_in.dim() == 2:
        squeeze = True
        x_in, mass = x_in[None], mass[None]
        evals = evals[None] if evals is not None else None
        evecs = evecs[None] if evecs is not None else None
        edges = edges[None] if edges is not None else None
        faces = faces[None] if faces is not None else None
        if gradX is not None and not isinstance(gradX, (list, tuple)) and gradX.dim() == 2:
            gradX, gradY = [gradX], [gradY]
    elif x_in.dim() != 3:
        raise ValueError("x_in should be tensor with shape [N,C] or [B,N,C]")   # layers.py:363

    C_in = params["first_lin.weight"].shape[1]
    if x_in.shape[-1] != C_in:
        raise ValueError("wrong number of input channels")                # layers.py:343-344

    x = x_in @ params["first_lin.weight"].T + params["first_lin.bias"]      # layers.py:366
    for i in range(count_blocks(params)):                                    # layers.py:369-370
        km = keep_masks[i] if keep_masks is not None else None
        cap = [] if capture is not None else None
        x = block_forward(params, f"block_{i}", x, mass, evals, evecs, gradX, gradY, km,
                          act_patterns[i] if act_patterns is not None else None, cap)
        if capture is not None:
            capture.append(cap)
    x = x @ params["last_lin.weight"].T + params["last_lin.bias"]           # layers.py:373
    out = remap_outputs(x, outputs_at, mass=mass, edges=edges, faces=faces)
    if last_activation is not None:                                          # layers.py:400-401
        out = last_activation(out)
    return out[0] if squeeze else out                                        # layers.py:404-405


def net_forward_backward(params, inputs, outputs_at="vertices", last_activation=None,
                         keep_masks=None, loss_weights=None, act_patterns=None, capture=None):
    """Forward + autograd backward of ``sum(out * loss_weights)``; returns
    (out, grads-dict incl. 'x_in').  Parameters are re-leafed so callers keep theirs."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    x_in = inputs["x_in"].detach().clone().requires_grad_(True)
    kw = {k: v for k, v in inputs.items() if k != "x_in"}
    out = net_forward(leaf, x_in, outputs_at=outputs_at, last_activation=last_activation,
                      keep_masks=keep_masks, act_patterns=act_patterns, capture=capture, **kw)
    w = loss_weights if loss_weights is not None else torch.ones_like(out)
    (out * w).sum().backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    grads["x_in"] = x_in.grad
    return out.detach(), grads


def hks(evals, evecs, scales):
    """geometry.py:600-628: heat kernel signature, out[v, s] = sum_k exp(-evals[k] * scales[s]) * evecs[v, k]^2
    (unbatched restatement; the reference broadcasts a (B,V,S,K) term tensor and sums over K)."""
    coefs = torch.exp(-evals[None, :] * scales[:, None])                 # (S,K)
    return (evecs * evecs) @ coefs.t()                                   # (V,S)


def label_smoothing_log_loss(pred, label, smoothing):
    """utils.py:18-24 for the 1-D prediction of its only caller: -(q . pred) with q the smoothed one-hot."""
    n_class = pred.shape[-1]
    q = torch.full_like(pred, smoothing / (n_class - 1))
    q[label] = 1.0 - smoothing
    return -(q * pred).sum()
