"""Drop-in ``diffusion_net.layers`` for MI355X: same classes, constructor arguments, forward
signatures, error behaviour and ``state_dict`` keys as the reference's
``src/diffusion_net/layers.py``, with every forward/backward op executed by hand-written HIP
kernels (libdiffnet_hip.so) instead of ATen.

Reference anchors: DiffusionNet (layers.py:244-407), DiffusionNetBlock (:167-241),
LearnedTimeDiffusion (:17-90), SpatialGradientFeatures (:93-130), MiniMLP (:133-164).

Beyond the reference API, ``DiffusionNet.forward_packed`` runs a *ragged* batch of meshes
(different vertex counts) packed in a :class:`~diffusion_net.batch.MeshBatch` in one set of
launches -- the unit of work the multi-GPU sharding distributes.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn

from . import _hip, autograph, ops
from .batch import GatherPattern, MeshBatch, operator_cache

_MIN_TIME = 1e-8


def _compiling():
    """True while torch.compile / torch.export is tracing: the ops then go through their torch.library registrations (torchlib.py),
    which a compiler keeps as opaque graph nodes; eager execution calls the autograd Functions of ops.py directly."""
    if torch.compiler.is_compiling():
        from . import torchlib  # noqa: F401  (registers torch.ops.diffusion_net.*)
        return True
    return False


def _linear_init_(weight: torch.Tensor, bias: Optional[torch.Tensor]):
    """Same distribution as ``nn.Linear.reset_parameters`` (U(+-1/sqrt(fan_in)))."""
    nn.init.kaiming_uniform_(weight, a=math.sqrt(5))
    if bias is not None:
        bound = 1.0 / math.sqrt(weight.shape[1]) if weight.shape[1] > 0 else 0.0
        nn.init.uniform_(bias, -bound, bound)


class _RowLinear(nn.Module):
    """Parameters of an ``nn.Linear`` (keys ``weight``/``bias``); applied by the HIP row-GEMM."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        _linear_init_(self.weight, self.bias)

    def apply_rows(self, x2d, mb):
        b = self.bias if self.bias is not None else torch.zeros(self.out_features, device=x2d.device)
        if _compiling():
            return torch.ops.diffusion_net.linear(x2d, self.weight, b, mb.handle)
        return ops.LinearFn.apply(x2d, self.weight, b, mb)

    def forward(self, x):
        lead = x.shape[:-1]
        x2d = x.reshape(-1, x.shape[-1])
        mb = MeshBatch.rows_only(x2d.shape[0], x2d.device)
        return self.apply_rows(x2d, mb).reshape(*lead, self.out_features)


def _batch_of(mass, evals, evecs, gradX=None, gradY=None):
    """MeshBatch from reference-style tensors (batched [B,..] or unbatched)."""
    if evecs.dim() == 2:
        mass, evals, evecs = mass[None], evals[None], evecs[None]
        if gradX is not None and gradX.dim() == 2:
            gradX, gradY = gradX.unsqueeze(0), gradY.unsqueeze(0)
    return operator_cache.lookup(mass, evals, evecs, gradX, gradY, None, ("ops", gradX is not None),
                                 lambda: (MeshBatch.from_reference_args(mass, evals, evecs, gradX, gradY), None))[0]


class LearnedTimeDiffusion(nn.Module):
    """Per-channel learned-time heat diffusion (reference layers.py:17-90)."""

    def __init__(self, C_inout, method="spectral"):
        super().__init__()
        self.C_inout = C_inout
        self.diffusion_time = nn.Parameter(torch.zeros(C_inout))      # layers.py:38,41
        self.method = method

    def clamp_time_(self):
        # same observable side effect as layers.py:48-49 (the Parameter holds its clamp afterwards, so the
        # optimizer sees the clamped value); done in place so flat parameter buckets keep their storage
        # (.data: the clamp is idempotent between optimizer steps and must not invalidate tensors another
        # in-flight forward saved for its backward)
        if torch.compiler.is_compiling():
            with torch.no_grad():
                self.diffusion_time.clamp_(min=_MIN_TIME)
            return
        self.diffusion_time.data.clamp_(min=_MIN_TIME)

    def forward(self, x, L, mass, evals, evecs):
        self.clamp_time_()
        if x.shape[-1] != self.C_inout:
            raise ValueError("Tensor has wrong shape = {}. Last dim shape should have number of channels = {}".format(
                x.shape, self.C_inout))
        if self.method == "spectral":
            mb = _batch_of(mass, evals, evecs)
            out = ops.DiffusionFn.apply(x.reshape(-1, self.C_inout), self.diffusion_time, mb)
            return out.reshape(x.shape)
        if self.method == "implicit_dense":
            # Toy-size dense solve (reference layers.py:69-84); not part of the accelerated path
            # (no experiment or benchmark config uses it) -- plain torch.linalg on the tensors' device.
            V = x.shape[-2]
            mat = L.to_dense().unsqueeze(1).expand(-1, self.C_inout, V, V).clone()
            mat *= self.diffusion_time[None, :, None, None]
            mat += torch.diag_embed(mass).unsqueeze(1)
            chol = torch.linalg.cholesky(mat)
            rhs = (x * mass.unsqueeze(-1)).transpose(1, 2).unsqueeze(-1)
            return torch.cholesky_solve(rhs, chol).squeeze(-1).transpose(1, 2)
        raise ValueError("unrecognized method")


class SpatialGradientFeatures(nn.Module):
    """tanh(<grad, learned-rotation(grad)>) features (reference layers.py:93-130)."""

    def __init__(self, C_inout, with_gradient_rotations=True):
        super().__init__()
        self.C_inout = C_inout
        self.with_gradient_rotations = with_gradient_rotations
        if with_gradient_rotations:
            self.A_re = _RowLinear(C_inout, C_inout, bias=False)
            self.A_im = _RowLinear(C_inout, C_inout, bias=False)
        else:
            self.A = _RowLinear(C_inout, C_inout, bias=False)

    def matrices(self):
        if self.with_gradient_rotations:
            return self.A_re.weight, self.A_im.weight
        return self.A.weight, None

    def forward(self, vectors):
        # vectors: (..., V, C, 2) as in the reference; planar copies feed the HIP op
        lead = vectors.shape[:-2]
        gx = vectors[..., 0].reshape(-1, self.C_inout).contiguous()
        gy = vectors[..., 1].reshape(-1, self.C_inout).contiguous()
        mb = MeshBatch.rows_only(gx.shape[0], gx.device)
        A_re, A_im = self.matrices()
        return ops.GradFeatFn.apply(gx, gy, A_re, A_im, mb).reshape(*lead, self.C_inout)


class MiniMLP(nn.Sequential):
    """Linear/ReLU stack with Dropout(p=.5) before every layer but the first; module names (and
    therefore state_dict keys) follow the reference (layers.py:133-164)."""

    def __init__(self, layer_sizes, dropout=False, activation=nn.ReLU, name="miniMLP"):
        super().__init__()
        self.layer_sizes = list(layer_sizes)
        self.uses_dropout = bool(dropout)
        self._hip_fusable = activation is nn.ReLU
        n = len(layer_sizes) - 1
        for i in range(n):
            if dropout and i > 0:
                self.add_module(name + "_mlp_layer_dropout_{:03d}".format(i), nn.Dropout(p=.5))
            self.add_module(name + "_mlp_layer_{:03d}".format(i), _RowLinear(layer_sizes[i], layer_sizes[i + 1]))
            if i + 1 < n:
                self.add_module(name + "_mlp_act_{:03d}".format(i), activation())

    def linears(self) -> List[_RowLinear]:
        return [m for m in self if isinstance(m, _RowLinear)]


class DiffusionNetBlock(nn.Module):
    """diffusion -> spatial gradient features -> MiniMLP -> residual (reference layers.py:167-241),
    executed as one fused sequence of HIP launches (``dn_block_fwd_f32`` / ``dn_block_bwd_f32``)."""

    def __init__(self, C_width, mlp_hidden_dims, dropout=True, diffusion_method="spectral",
                 with_gradient_features=True, with_gradient_rotations=True):
        super().__init__()
        self.C_width = C_width
        self.mlp_hidden_dims = mlp_hidden_dims
        self.dropout = dropout
        self.with_gradient_features = with_gradient_features
        self.with_gradient_rotations = with_gradient_rotations
        self.diffusion = LearnedTimeDiffusion(C_width, method=diffusion_method)
        self.MLP_C = 2 * C_width
        if with_gradient_features:
            self.gradient_features = SpatialGradientFeatures(C_width, with_gradient_rotations=with_gradient_rotations)
            self.MLP_C += C_width
        self.mlp = MiniMLP([self.MLP_C] + list(mlp_hidden_dims) + [C_width], dropout=dropout)
        self._cfg = ops.BlockConfig(C_width, self.mlp.layer_sizes, with_gradient_features, with_gradient_rotations)
        self._cfg.clamp_time = True       # forward_packed: the in-place clamp of diffusion_time rides in the block call (layers.py:48-49)
        self.mask_provider = None   # test hook: callable(layer_index, shape, device) -> uint8 keep mask
        self.drop_seed_provider = None   # test hook: callable() -> int seed of the in-kernel dropout

    def _dropout_masks(self, n_rows, device):
        """nn.Dropout(p=.5) on the inputs of MLP layers 1.. (layers.py:143-147) in train mode.  Default: a 64-bit seed drawn
        from torch's CPU generator (reproducible under torch.manual_seed); the HIP epilogues derive the keep bits from it, no
        mask tensor exists.  With a ``mask_provider`` (tests): explicit uint8 keep masks."""
        if not (self.training and self.dropout):
            return None
        if self.mask_provider is None:
            if torch.compiler.is_compiling():   # a traceable draw: constant host part + a device word from torch's device generator
                return (0x5DEECE66D | 1, torch.randint(1, 2 ** 62, (1,), dtype=torch.int64, device=device))
            gs = getattr(self, "_graph_seed", None)
            if gs is not None:           # graphs.GraphedTrainStep: (constant host part, device word advanced by the graph itself)
                return gs
            if self.drop_seed_provider is not None:
                return int(self.drop_seed_provider())
            seed = int(torch.randint(1, 2 ** 62, (1,), dtype=torch.int64).item())
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                # replicas are seeded identically; their dropout masks must not be
                seed = (seed + 0x9E3779B97F4A7C15 * (torch.distributed.get_rank() + 1)) % (2 ** 62) or 1
            return seed
        masks = [None]
        for i in range(1, self._cfg.n_mlp):
            shape = (n_rows, self._cfg.widths[i])
            masks.append(self.mask_provider(i, shape, device).to(device=device, dtype=torch.uint8).contiguous())
        return masks

    def forward_packed(self, x2d: torch.Tensor, mb: MeshBatch) -> torch.Tensor:
        """x2d: [v_total, C] on the concatenated vertex axis of ``mb``."""
        if x2d.shape[-1] != self.C_width:
            raise ValueError("Tensor has wrong shape = {}. Last dim shape should have number of channels = {}".format(
                x2d.shape, self.C_width))
        if self.diffusion.method != "spectral" or not self.mlp._hip_fusable:
            raise NotImplementedError("the HIP block implements diffusion_method='spectral' with ReLU MLPs")
        if self.with_gradient_features and not mb.has_grad:
            raise ValueError("gradient features need gradX/gradY")
        if _compiling():
            self.diffusion.clamp_time_()
        # (eager: the clamp of layers.py:48-49 happens in place inside the block call's first launch, BlockConfig.clamp_time)
        A_re = A_im = None
        if self.with_gradient_features:
            A_re, A_im = self.gradient_features.matrices()
        wb = []
        for lin in self.mlp.linears():
            wb += [lin.weight, lin.bias]
        masks = self._dropout_masks(x2d.shape[0], x2d.device)
        if _compiling():
            if isinstance(masks, list):
                raise NotImplementedError("explicit dropout masks (a test hook) are not available under torch.compile")
            seed, seed_dev = (masks if isinstance(masks, tuple) else (masks or 0, None))
            return torch.ops.diffusion_net.block(x2d, self.diffusion.diffusion_time, A_re, A_im, wb, mb.handle, self._cfg.handle, seed, seed_dev,
                                                 mb.n_mesh, mb.k_eig)[0]
        self._cfg._grad_enabled = torch.is_grad_enabled()      # (sampled here: inside the Function's forward grad mode is always off)
        return ops.BlockFn.apply(mb, self._cfg, masks, x2d, self.diffusion.diffusion_time, A_re, A_im, *wb)

    def forward(self, x_in, mass, L, evals, evecs, gradX, gradY):
        # reference signature (layers.py:200): batched [B,V,C] inputs
        if x_in.shape[-1] != self.C_width:
            raise ValueError("Tensor has wrong shape = {}. Last dim shape should have number of channels = {}".format(
                x_in.shape, self.C_width))
        if self.diffusion.method != "spectral":
            return self._forward_unfused(x_in, mass, L, evals, evecs, gradX, gradY)
        mb = _batch_of(mass, evals, evecs, gradX if self.with_gradient_features else None,
                       gradY if self.with_gradient_features else None)
        return self.forward_packed(x_in.reshape(-1, self.C_width), mb).reshape(x_in.shape)

    def _forward_unfused(self, x_in, mass, L, evals, evecs, gradX, gradY):
        # only reached for diffusion_method='implicit_dense' (toy sizes, out of the accelerated scope)
        xd = self.diffusion(x_in, L, mass, evals, evecs)
        feats = [x_in, xd]
        if self.with_gradient_features:
            B, V, Cw = xd.shape
            mbg = _batch_of(mass, mass.new_zeros(B, 1), mass.new_zeros(B, V, 1), gradX, gradY)
            gx, gy = ops.GradApplyFn.apply(xd.reshape(-1, Cw), mbg)
            A_re, A_im = self.gradient_features.matrices()
            feats.append(ops.GradFeatFn.apply(gx, gy, A_re, A_im, mbg).reshape(B, V, Cw))
        return self.mlp(torch.cat(feats, dim=-1)) + x_in


class DiffusionNet(nn.Module):
    """Same constructor and ``forward`` contract as the reference ``DiffusionNet`` (layers.py:244-407)."""

    def __init__(self, C_in, C_out, C_width=128, N_block=4, last_activation=None, outputs_at="vertices",
                 mlp_hidden_dims=None, dropout=True, with_gradient_features=True, with_gradient_rotations=True,
                 diffusion_method="spectral"):
        super().__init__()
        self.C_in, self.C_out, self.C_width, self.N_block = C_in, C_out, C_width, N_block
        self.last_activation = last_activation
        self.outputs_at = outputs_at
        if outputs_at not in ["vertices", "edges", "faces", "global_mean"]:
            raise ValueError("invalid setting for outputs_at")
        if mlp_hidden_dims is None:
            mlp_hidden_dims = [C_width, C_width]
        self.mlp_hidden_dims = mlp_hidden_dims
        self.dropout = dropout
        self.diffusion_method = diffusion_method
        if diffusion_method not in ["spectral", "implicit_dense"]:
            raise ValueError("invalid setting for diffusion_method")
        self.with_gradient_features = with_gradient_features
        self.with_gradient_rotations = with_gradient_rotations

        self.first_lin = _RowLinear(C_in, C_width)
        self.last_lin = _RowLinear(C_width, C_out)
        self.blocks = []
        for i in range(N_block):
            blk = DiffusionNetBlock(C_width=C_width, mlp_hidden_dims=mlp_hidden_dims, dropout=dropout,
                                    diffusion_method=diffusion_method,
                                    with_gradient_features=with_gradient_features,
                                    with_gradient_rotations=with_gradient_rotations)
            self.blocks.append(blk)
            self.add_module("block_" + str(i), blk)

    def _activation_is_log_softmax(self):
        """The scripts pass ``last_activation=lambda x: torch.nn.functional.log_softmax(x, dim=-1)`` (human_segmentation_original.py:75):
        an opaque callable.  It is recognised by what it does to a probe tensor, once per module and activation object, so that the
        per-face / per-vertex head can run as ONE kernel (gather-mean + log_softmax, dn_head.hip) instead of three."""
        act = self.last_activation
        if act is None:
            return False
        hit = getattr(self, "_lsm_probe", None)
        if hit is None or hit[0] is not act:
            g = torch.Generator().manual_seed(0)
            probe = torch.randn(3, max(self.C_out, 2), generator=g) * 3.0
            try:
                with torch.no_grad():
                    ok = bool(torch.equal(act(probe), torch.nn.functional.log_softmax(probe, dim=-1)))
            except Exception:      # noqa: BLE001  (an activation that does not take a [3, C] tensor is simply not log_softmax)
                ok = False
            hit = (act, ok)
            object.__setattr__(self, "_lsm_probe", hit)
        return hit[1]

    def forward_packed_loss(self, x2d, mb: MeshBatch, gather: Optional[GatherPattern], labels, smoothing: float = 0.0):
        """``forward_packed`` followed by the mean NLL (or label-smoothed log loss) against ``labels`` with the whole head -- remap,
        log_softmax, loss -- in one kernel each way.  Needs a log_softmax ``last_activation`` (as the segmentation scripts have) and
        outputs at vertices, edges or faces.  Returns (log-probabilities, loss)."""
        if not self._activation_is_log_softmax() or self.outputs_at == "global_mean" or self.C_out > _hip.HEAD_MAX_CLASSES:
            preds = self.forward_packed(x2d, mb, gather)
            from .utils import label_smoothing_log_loss, nll_loss
            return preds, (nll_loss(preds, labels) if smoothing == 0.0 else label_smoothing_log_loss(preds, labels, smoothing))
        x = self._trunk(x2d, mb)
        return ops.HeadFn.apply(x, gather if self.outputs_at in ("edges", "faces") else None, labels, True, float(smoothing), True)

    def _trunk(self, x2d, mb):
        if x2d.shape[-1] != self.C_in:
            raise ValueError("DiffusionNet was constructed with C_in={}, but x_in has last dim={}".format(
                self.C_in, x2d.shape[-1]))
        x = self.first_lin.apply_rows(x2d, mb)
        for blk in self.blocks:
            x = blk.forward_packed(x, mb)
        return self.last_lin.apply_rows(x, mb)

    # ------------------------------------------------------------------ ragged-batch entry point
    def forward_packed(self, x2d, mb: MeshBatch, gather: Optional[GatherPattern] = None):
        """x2d: [v_total, C_in] features on the concatenated vertex axis of ``mb``.

        Returns [v_total, C_out] ('vertices'), [n_faces_or_edges_total, C_out] ('faces'/'edges',
        ``gather`` built from global vertex ids) or [n_mesh, C_out] ('global_mean'); the last
        activation is applied as in ``forward``."""
        x = self._trunk(x2d, mb)
        comp = _compiling()
        if self.outputs_at != "global_mean" and self.C_out <= _hip.HEAD_MAX_CLASSES and self._activation_is_log_softmax():
            # remap + log_softmax as one kernel (the scripts' per-face / per-vertex log-probabilities)
            pat = gather if self.outputs_at in ("edges", "faces") else None
            if comp:
                return torch.ops.diffusion_net.head(x, pat.handle if pat is not None else 0, None, True, 0.0, pat.n_out if pat is not None else x.shape[0])[0]
            return ops.HeadFn.apply(x, pat, None, True, 0.0, True)[0]
        if self.outputs_at in ("edges", "faces"):
            x = torch.ops.diffusion_net.gather_mean(x, gather.handle, gather.n_out) if comp else ops.GatherMeanFn.apply(x, gather)
        elif self.outputs_at == "global_mean":
            x = torch.ops.diffusion_net.mass_mean(x, mb.handle, mb.n_mesh)[0] if comp else ops.MassMeanFn.apply(x, mb)
        if self.last_activation is not None:
            x = self.last_activation(x)
        return x

    # ------------------------------------------------------------------ reference entry point
    def forward(self, x_in, mass, L=None, evals=None, evecs=None, gradX=None, gradY=None, edges=None, faces=None):
        if x_in.shape[-1] != self.C_in:
            raise ValueError("DiffusionNet was constructed with C_in={}, but x_in has last dim={}".format(
                self.C_in, x_in.shape[-1]))
        # identity / content keys: the operands as handed over -- only those that enter the pack: the index tensor of the output remap
        # is keyed (and later read on the device) only when outputs_at uses it (a caller may leave unused faces on the host)
        idx_key = edges if self.outputs_at == "edges" else (faces if self.outputs_at == "faces" else None)
        key_ops = (mass, evals, evecs, gradX, gradY, idx_key)
        if x_in.dim() == 2:
            squeeze = True
        elif x_in.dim() == 3:
            squeeze = False
        else:
            raise ValueError("x_in should be tensor with shape [N,C] or [B,N,C]")

        def batched():
            """The operands with the batch dimension of layers.py:346-358.  Evaluated only when something is packed: ``unsqueeze`` of a
            sparse tensor copies it (and synchronises with the host on a ROCm device), and the steady state of a cached mesh needs none
            of them."""
            if not squeeze:
                return mass, L, evals, evecs, gradX, gradY, edges, faces
            u = lambda t: t.unsqueeze(0) if t is not None else None
            return tuple(u(t) for t in (mass, L, evals, evecs, gradX, gradY, edges, faces))

        if self.diffusion_method != "spectral":
            mass_b, L_b, evals_b, evecs_b, gX_b, gY_b, edges_b, faces_b = batched()
            return self._forward_unfused(x_in.unsqueeze(0) if squeeze else x_in, mass_b, L_b, evals_b, evecs_b, gX_b, gY_b, edges_b, faces_b, squeeze)

        if squeeze:
            x_in = x_in.unsqueeze(0)
        B, V, _ = x_in.shape
        use_grad = self.with_gradient_features
        idx = None
        if self.outputs_at in ("edges", "faces"):
            idx = edges if self.outputs_at == "edges" else faces
            n_per_mesh = idx.shape[-2]                                 # AttributeError on None, as the reference

        def pack():
            mass_b, _, evals_b, evecs_b, gX_b, gY_b, edges_b, faces_b = batched()
            mb_ = MeshBatch.from_reference_args(mass_b, evals_b, evecs_b, gX_b if use_grad else None, gY_b if use_grad else None)
            gather_ = None
            if idx is not None:
                idx_b = edges_b if self.outputs_at == "edges" else faces_b
                offs = (torch.arange(B, device=idx_b.device, dtype=idx_b.dtype) * V).view(B, 1, 1)
                gather_ = GatherPattern((idx_b + offs).reshape(-1, idx_b.shape[-1]), B * V)
            return mb_, gather_
        # the packed operators of a mesh are built once and found again on later calls (batch.OperatorCache)
        mb, gather = operator_cache.lookup(None, None, None, None, None, None, ("net", use_grad, self.outputs_at, squeeze), pack,
                                           key_ops[:3] + ((gradX, gradY) if use_grad else (None, None)) + key_ops[5:])
        x2d = x_in.reshape(B * V, self.C_in)
        out = autograph.run(self, x2d, mb, gather)     # replay of a captured graph once this mesh has been seen a few times
        if out is None:
            out = self.forward_packed(x2d, mb, gather)
        if self.outputs_at == "vertices":
            out = out.reshape(B, V, -1)
        elif self.outputs_at in ("edges", "faces"):
            out = out.reshape(B, n_per_mesh, -1)
        return out.squeeze(0) if squeeze else out

    def _forward_unfused(self, x_in, mass, L, evals, evecs, gradX, gradY, edges, faces, squeeze):
        # diffusion_method='implicit_dense' only (outside the accelerated scope; see LearnedTimeDiffusion)
        x = self.first_lin(x_in)
        for blk in self.blocks:
            x = blk(x, mass, L, evals, evecs, gradX, gradY)
        x = self.last_lin(x)
        if self.outputs_at in ("edges", "faces"):
            idx = edges if self.outputs_at == "edges" else faces
            x = torch.stack([x[b][idx[b]].mean(dim=1) for b in range(x.shape[0])], 0)
        elif self.outputs_at == "global_mean":
            x = torch.sum(x * mass.unsqueeze(-1), dim=-2) / torch.sum(mass, dim=-1, keepdim=True)
        if self.last_activation is not None:
            x = self.last_activation(x)
        return x.squeeze(0) if squeeze else x
