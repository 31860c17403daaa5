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
from .precompute import (compute_hks, compute_hks_autoscale, compute_operators, get_all_operators,  # noqa: F401
                         get_operators, normalize_positions)


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
