"""``diffusion_net.geometry`` -- the two basis transforms on the hot path, HIP-backed.

Signatures follow the reference (geometry.py:572-598): batched tensors, basis [B,V,K],
values [B,V,C] / [B,K,C], massvec [B,V] (an unbatched leading dimension is accepted too).
The host-side operator precompute keeps the reference's names here (``compute_operators``, ``get_operators``,
``get_all_operators``, ``normalize_positions``, ``compute_hks[_autoscale]``) and lives in ``precompute.py``
(numpy/scipy restatement, triangle meshes).
"""
import torch

from . import ops
from .batch import MeshBatch
from .precompute import compute_operators, get_all_operators, get_operators, normalize_positions  # noqa: F401
from . import precompute as _pre


def _spectral_batch(basis, massvec):
    if basis.dim() == 2:
        basis = basis[None]
        massvec = massvec[None] if massvec is not None else None
    B, V, K = basis.shape
    if massvec is None:
        massvec = torch.ones(B, V, dtype=basis.dtype, device=basis.device)
    return MeshBatch.from_reference_args(massvec, torch.zeros(B, K, dtype=basis.dtype, device=basis.device), basis)


def to_basis(values, basis, massvec):
    """(B,V,D),(B,V,K),(B,V) -> (B,K,D): basis^T (values * mass)."""
    mb = _spectral_batch(basis, massvec)
    D = values.shape[-1]
    spec = ops.ToBasisFn.apply(values.reshape(-1, D), mb)
    return spec if values.dim() == 3 else spec[0]


def from_basis(values, basis):
    """(B,K,D),(B,V,K) -> (B,V,D): basis @ values."""
    if values.is_complex() or basis.is_complex():
        raise NotImplementedError("complex basis transforms are dead code in the reference (geometry.py:595-596)")
    mb = _spectral_batch(basis, None)
    D = values.shape[-1]
    out = ops.FromBasisFn.apply(values.reshape(mb.n_mesh, -1, D), mb)
    return out.reshape(*basis.shape[:-1], D)


def compute_hks(evals, evecs, scales):
    """Heat kernel signature (geometry.py:600-628): (K),(V,K),(S) -> (V,S), or batched (B,K),(B,V,K),(B,S) -> (B,V,S).
    On a ROCm device and outside autograd this is one streaming HIP pass over the eigenbasis (``dn_hks_f32``); host
    tensors (the reference computes it in its CPU dataset loaders) and differentiable calls use the torch formula."""
    on_dev = evecs.is_cuda and evals.is_cuda and scales.is_cuda and evecs.dtype == torch.float32
    if not on_dev or evecs.requires_grad or evals.requires_grad or scales.requires_grad:
        return _pre.compute_hks(evals, evecs, scales)
    return ops.hks(evals, evecs, scales)


def compute_hks_autoscale(evals, evecs, count):
    scales = torch.logspace(-2, 0.0, steps=count, device=evals.device, dtype=evals.dtype)   # geometry.py:630-633
    return compute_hks(evals, evecs, scales)
