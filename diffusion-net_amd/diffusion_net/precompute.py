"""Host-side geometry precompute: the per-mesh operators the hot path consumes.

north_star keeps this stage on the host CPU, cached on disk "as the reference does"; the reference delegates the
Laplacian to potpourri3d (not installable here) and builds the gradient operator in a per-vertex Python loop
(geometry.py:222-263).  This module restates the pipeline of ``compute_operators`` (geometry.py:276-392) with
numpy/scipy only, fully vectorised:

    normals -> tangent frames -> cotan Laplacian + lumped mass -> generalized eigenproblem (shift-invert eigsh)
            -> per-vertex least-squares gradient operator (complex; split into gradX / gradY)

and the npz operator cache of ``get_operators`` (geometry.py:426-570) with the same file naming
(sha1(verts,faces)_<bucket>.npz) and the same keys, so caches written by either implementation are interchangeable.
Triangle meshes only: the point-cloud branch needs robust_laplacian's point-cloud Laplacian and is not restated.
"""
from __future__ import annotations

import os

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as sla
import torch

from .utils import hash_arrays  # noqa: F401  (the cache key; also reachable as precompute.hash_arrays)

EPS = 1e-8            # geometry.py:309
GRAD_REG = 1e-5       # geometry.py:239


# ----------------------------------------------------------------------------------------- Laplacian, mass
def cotan_laplacian(verts: np.ndarray, faces: np.ndarray, denom_eps: float = 1e-10) -> sp.csr_matrix:
    """Positive semi-definite weak cotan Laplacian: for every face corner k with opposite edge (i,j),
    w = 0.5*cot(angle_k) = 0.5 * <a,b> / (|a x b| + eps); L[i,i] += w, L[j,j] += w, L[i,j] -= w, L[j,i] -= w."""
    V = verts.shape[0]
    rows, cols, vals = [], [], []
    for k in range(3):
        i, j, o = faces[:, (k + 1) % 3], faces[:, (k + 2) % 3], faces[:, k]
        a, b = verts[i] - verts[o], verts[j] - verts[o]
        w = 0.5 * np.einsum("ij,ij->i", a, b) / (np.linalg.norm(np.cross(a, b), axis=1) + denom_eps)
        rows += [i, j, i, j]
        cols += [i, j, j, i]
        vals += [w, w, -w, -w]
    L = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(V, V))
    return L.tocsr()


def vertex_areas(verts: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """Barycentric lumped mass: one third of the area of every incident face."""
    e1, e2 = verts[faces[:, 1]] - verts[faces[:, 0]], verts[faces[:, 2]] - verts[faces[:, 0]]
    area = 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1)
    return np.bincount(faces.reshape(-1), weights=np.repeat(area / 3.0, 3), minlength=verts.shape[0])


# ----------------------------------------------------------------------------------------- normals, frames
def _unit_face_normal_sum(verts, faces):
    n = np.cross(verts[faces[:, 1]] - verts[faces[:, 0]], verts[faces[:, 2]] - verts[faces[:, 0]])
    n = n / (np.linalg.norm(n, axis=1, keepdims=True) + 1e-6)         # normalize() of geometry.py:37-47
    out = np.zeros_like(verts)
    for k in range(3):
        np.add.at(out, faces[:, k], n)
    with np.errstate(invalid="ignore", divide="ignore"):
        return out / np.linalg.norm(out, axis=1, keepdims=True)


def vertex_normals(verts: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """Sum of unit face normals per vertex, normalised; degenerate vertices are re-estimated on slightly jittered
    positions and, failing that, get a fixed pseudo-random direction (behaviour of geometry.py:114-148)."""
    normals = _unit_face_normal_sum(verts, faces)
    bad = np.isnan(normals).any(axis=1)
    if bad.any():
        rng = np.random.RandomState(seed=777)
        scale = np.linalg.norm(verts.max(0) - verts.min(0)) * 1e-4
        jitter = (rng.rand(*verts.shape) - 0.5) * scale
        normals = _unit_face_normal_sum(verts + bad[:, None] * jitter, faces)
        bad = np.isnan(normals).any(axis=1)
        if bad.any():
            rnd = np.random.RandomState(seed=777).rand(*verts.shape) - 0.5
            normals[bad] = rnd[bad] / np.linalg.norm(rnd[bad], axis=1, keepdims=True)
    if np.isnan(normals).any():
        raise ValueError("NaN normals :(")
    return normals


def tangent_frames(normals: np.ndarray) -> np.ndarray:
    """(V,3,3) rows = (basisX, basisY, normal): basisX is e_x (or e_y where the normal is within ~26 degrees of e_x)
    projected onto the tangent plane, basisY = n x basisX (geometry.py:151-177)."""
    V = normals.shape[0]
    cand = np.where((np.abs(normals[:, 0]) < 0.9)[:, None], np.array([1.0, 0, 0]), np.array([0, 1.0, 0]))
    bx = cand - normals * np.einsum("ij,ij->i", cand, normals)[:, None]
    bx = bx / (np.linalg.norm(bx, axis=1, keepdims=True) + 1e-6)
    by = np.cross(normals, bx)
    frames = np.stack([bx, by, normals], axis=1)
    if np.isnan(frames).any():
        raise ValueError("NaN coordinate frame! Must be very degenerate")
    return frames.reshape(V, 3, 3)


# ----------------------------------------------------------------------------------------- gradient operator
def gradient_operator(verts: np.ndarray, frames: np.ndarray, edges: np.ndarray) -> sp.csc_matrix:
    """Complex (V,V) operator: row v holds the least-squares gradient stencil of vertex v over its outgoing edges
    (tail v -> tip j), expressed in v's tangent frame (real = X, imag = Y).

    Per vertex, with t_e the 2-D tangent coordinates of edge e:  G = sum_e t_e t_e^T + 1e-5 I,
    coefficient of neighbour j_e = G^-1 t_e, coefficient of v itself = -sum_e G^-1 t_e.
    (Vectorised restatement of the per-vertex solve of geometry.py:222-263.)"""
    V = verts.shape[0]
    tail, tip = edges[0], edges[1]
    keep = tail != tip
    tail, tip = tail[keep], tip[keep]
    d = verts[tip] - verts[tail]
    tx = np.einsum("ij,ij->i", d, frames[tail, 0])
    ty = np.einsum("ij,ij->i", d, frames[tail, 1])
    gxx = np.bincount(tail, weights=tx * tx, minlength=V) + GRAD_REG
    gxy = np.bincount(tail, weights=tx * ty, minlength=V)
    gyy = np.bincount(tail, weights=ty * ty, minlength=V) + GRAD_REG
    det = gxx * gyy - gxy * gxy
    ixx, ixy, iyy = gyy / det, -gxy / det, gxx / det
    cx = ixx[tail] * tx + ixy[tail] * ty
    cy = ixy[tail] * tx + iyy[tail] * ty
    coef = cx + 1j * cy
    diag = -(np.bincount(tail, weights=cx, minlength=V) + 1j * np.bincount(tail, weights=cy, minlength=V))
    rows = np.concatenate([np.arange(V), tail])
    cols = np.concatenate([np.arange(V), tip])
    data = np.concatenate([diag, coef])
    return sp.coo_matrix((data, (rows, cols)), shape=(V, V)).tocsc()


# ----------------------------------------------------------------------------------------- eigenbasis
def laplacian_eigenbasis(L: sp.spmatrix, massvec: np.ndarray, k_eig: int):
    """k smallest generalized eigenpairs of (L + eps I) phi = lambda M phi by shift-invert Lanczos; up to 4 retries with a
    growing diagonal shift if the factorisation fails (geometry.py:337-361).  Eigenvalues clipped at 0."""
    if k_eig == 0:
        return np.zeros((0,)), np.zeros((L.shape[0], 0))
    A = (L + sp.identity(L.shape[0]) * EPS).tocsc()
    M = sp.diags(massvec)
    fails = 0
    while True:
        try:
            evals, evecs = sla.eigsh(A, k=k_eig, M=M, sigma=EPS)
            return np.clip(evals, 0.0, np.inf), evecs
        except Exception as err:   # noqa: BLE001 -- ARPACK / SuperLU raise assorted types
            if fails > 3:
                raise ValueError("failed to compute eigendecomp") from err
            fails += 1
            A = A + sp.identity(L.shape[0]) * (EPS * 10 ** fails)


# ----------------------------------------------------------------------------------------- torch-facing API
def _to_torch_sparse(mat, device, dtype):
    coo = mat.tocoo()
    idx = torch.from_numpy(np.vstack((coo.row, coo.col)).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(coo.data.astype(np.float64)), coo.shape).coalesce().to(device=device, dtype=dtype)


def compute_operators(verts, faces, k_eig, normals=None):
    """Same contract as the reference's ``compute_operators`` (geometry.py:276-392), torch in / torch out:
    returns (frames [V,3,3], massvec [V], L sparse, evals [k], evecs [V,k], gradX sparse, gradY sparse)."""
    device, dtype = verts.device, verts.dtype
    if faces.numel() == 0:
        raise NotImplementedError("point clouds need robust_laplacian's point-cloud Laplacian (not restated); triangle meshes only")
    v = verts.detach().cpu().numpy().astype(np.float64)
    f = faces.detach().cpu().numpy().astype(np.int64)
    n = vertex_normals(v, f) if normals is None else normals.detach().cpu().numpy().astype(np.float64)
    frames = tangent_frames(n)
    L = cotan_laplacian(v, f, denom_eps=1e-10)
    mass = vertex_areas(v, f)
    mass = mass + EPS * mass.mean()
    if np.isnan(L.data).any():
        raise RuntimeError("NaN Laplace matrix")
    if np.isnan(mass).any():
        raise RuntimeError("NaN mass matrix")
    evals, evecs = laplacian_eigenbasis(L, mass, k_eig)
    Lc = L.tocoo()
    grad = gradient_operator(v, frames, np.stack([Lc.row, Lc.col]))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)
    return (t(frames), t(mass), _to_torch_sparse(L, device, dtype), t(evals), t(evecs),
            _to_torch_sparse(grad.real, device, dtype), _to_torch_sparse(grad.imag, device, dtype))


def _sparse_to_csc(t):
    idx, val = t.coalesce().indices().cpu().numpy(), t.coalesce().values().cpu().numpy()
    return sp.coo_matrix((val, (idx[0], idx[1])), shape=tuple(t.shape)).tocsc()


def get_operators(verts, faces, k_eig=128, op_cache_dir=None, normals=None, overwrite_cache=False):
    """``compute_operators`` behind the reference's on-disk npz cache (geometry.py:426-570): same key derivation, same
    file names, same array names (float32 payload, CSC triplets for L / gradX / gradY), so either implementation can
    read the other's cache.  A hit is verified against the stored verts/faces; an entry with too few eigenpairs is
    rebuilt."""
    device, dtype = verts.device, verts.dtype
    v_np, f_np = verts.detach().cpu().numpy(), faces.detach().cpu().numpy()
    if np.isnan(v_np).any():
        raise RuntimeError("tried to construct operators from NaN verts")
    path = None
    if op_cache_dir is not None:
        os.makedirs(op_cache_dir, exist_ok=True)
        key = hash_arrays((v_np, f_np))
        bucket = 0
        while True:
            path = os.path.join(op_cache_dir, f"{key}_{bucket}.npz")
            if not os.path.exists(path):
                break
            try:
                z = np.load(path, allow_pickle=True)
                if not (np.array_equal(v_np, z["verts"]) and np.array_equal(f_np, z["faces"])):
                    bucket += 1            # hash collision: next bucket
                    continue
                if overwrite_cache or int(z["k_eig"].item()) < k_eig or "L_data" not in z:
                    os.remove(path)
                    break
                rd = lambda p: sp.csc_matrix((z[p + "_data"], z[p + "_indices"], z[p + "_indptr"]), shape=tuple(z[p + "_shape"]))
                t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)
                return (t(z["frames"]), t(z["mass"]), _to_torch_sparse(rd("L"), device, dtype), t(z["evals"][:k_eig]),
                        t(z["evecs"][:, :k_eig]), _to_torch_sparse(rd("gradX"), device, dtype),
                        _to_torch_sparse(rd("gradY"), device, dtype))
            except Exception:              # unreadable entry: rebuild in place  # noqa: BLE001
                break
    out = compute_operators(verts, faces, k_eig, normals=normals)
    if path is not None:
        frames, mass, L, evals, evecs, gX, gY = out
        f32 = lambda a: a.detach().cpu().numpy().astype(np.float32)
        Lc, gXc, gYc = (_sparse_to_csc(m).astype(np.float32) for m in (L, gX, gY))
        np.savez(path, verts=v_np.astype(np.float32), frames=f32(frames), faces=f_np, k_eig=k_eig, mass=f32(mass),
                 L_data=Lc.data, L_indices=Lc.indices, L_indptr=Lc.indptr, L_shape=Lc.shape,
                 evals=f32(evals), evecs=f32(evecs),
                 gradX_data=gXc.data, gradX_indices=gXc.indices, gradX_indptr=gXc.indptr, gradX_shape=gXc.shape,
                 gradY_data=gYc.data, gradY_indices=gYc.indices, gradY_indptr=gYc.indptr, gradY_shape=gYc.shape)
    return out


def get_all_operators(verts_list, faces_list, k_eig, op_cache_dir=None, normals=None):
    """Lists in, seven lists out (geometry.py:395-424)."""
    cols = [[] for _ in range(7)]
    for i, (v, f) in enumerate(zip(verts_list, faces_list)):
        res = get_operators(v, f, k_eig, op_cache_dir, normals=None if normals is None else normals[i])
        for c, r in zip(cols, res):
            c.append(r)
    return tuple(cols)


def normalize_positions(pos, faces=None, method="mean", scale_method="max_rad"):
    """Centre and unit-scale vertex positions (geometry.py:635-665)."""
    if method == "mean":
        pos = pos - pos.mean(dim=-2, keepdim=True)
    elif method == "bbox":
        lo, hi = pos.min(dim=-2).values, pos.max(dim=-2).values
        pos = pos - ((hi + lo) / 2.0).unsqueeze(-2)
    else:
        raise ValueError("unrecognized method")
    if scale_method == "max_rad":
        return pos / pos.norm(dim=-1).max(dim=-1, keepdim=True).values.unsqueeze(-1)
    if scale_method == "area":
        if faces is None:
            raise ValueError("must pass faces for area normalization")
        c = pos[faces]
        area = 0.5 * torch.cross(c[:, 1] - c[:, 0], c[:, 2] - c[:, 0], dim=-1).norm(dim=1).sum()
        return pos * (1.0 / torch.sqrt(area))
    raise ValueError("unrecognized scale method")


def compute_hks(evals, evecs, scales):
    """Heat-kernel signature sum_k exp(-lambda_k s) phi_k(v)^2 -> (V,S) or (B,V,S) (geometry.py:600-628)."""
    squeeze = evals.dim() == 1
    if squeeze:
        evals, evecs, scales = evals[None], evecs[None], scales[None]
    coefs = torch.exp(-evals.unsqueeze(1) * scales.unsqueeze(-1))          # (B,S,K)
    out = torch.einsum("bsk,bvk->bvs", coefs, evecs * evecs)
    return out[0] if squeeze else out


def compute_hks_autoscale(evals, evecs, count):
    scales = torch.logspace(-2, 0.0, steps=count, device=evals.device, dtype=evals.dtype)   # geometry.py:630-633
    return compute_hks(evals, evecs, scales)
