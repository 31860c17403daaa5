"""Seeded synthetic meshes + spectral operators for tests and benchmarks.

Host-side input generation only (numpy/torch CPU).  It produces the same tuple
of per-mesh operators that the reference's ``get_operators`` returns
(geometry.py:426-570: mass, evals, evecs, gradX, gradY as fp32 dense / sparse
COO) without the Laplacian eigen-solve, so that large benchmark inputs are
cheap (SURVEY.md 8d "fast synthetic operator set"):

* connectivity: a twisted-torus triangulation on V vertices (every vertex has
  valence 6 -> 7 non-zeros per gradient row incl. the diagonal, F = 2V faces),
* mass: positive lumped areas summing to ~4*pi,
* evecs: mass-orthonormal basis (Phi^T M Phi = I),
* evals: sorted, lambda_0 = 0, range like a unit sphere's first K eigenvalues,
* gradX/gradY: identical sparsity pattern, zero row sums, |entries| ~ 1/edge.

Throughput of the hot path does not depend on the operator values, only on
V, K, nnz; parity tests use the same generator at small sizes.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def torus_connectivity(V: int):
    """Twisted-torus triangulation on V >= 16 vertices.

    Returns (faces [2V,3] int64, nbr [V,6] int64) where nbr lists the six ring
    neighbours of every vertex (distinct, none equal to the vertex itself)."""
    if V < 16:
        raise ValueError("need V >= 16")
    s = max(2, int(round(math.sqrt(V))))
    while (s + 1) * 2 >= V:
        s -= 1
    v = np.arange(V, dtype=np.int64)
    a, b, c = (v + 1) % V, (v + s + 1) % V, (v + s) % V
    faces = np.concatenate([np.stack([v, a, b], 1), np.stack([v, b, c], 1)], 0)
    nbr = np.stack([(v + 1) % V, (v - 1) % V, (v + s) % V, (v - s) % V,
                    (v + s + 1) % V, (v - s - 1) % V], 1)
    return faces, nbr


def make_mesh_operators(V: int, K: int, seed: int = 0, dtype=torch.float32):
    """One synthetic mesh.  Returns a dict of CPU tensors:
    verts [V,3], faces [F,3] i64, edges [E,2] i64, mass [V], evals [K], evecs [V,K],
    gradX / gradY sparse COO [V,V] (coalesced, identical index sets)."""
    rng = np.random.RandomState(seed)
    faces, nbr = torus_connectivity(V)

    mass = (0.5 + rng.rand(V)) * (4.0 * math.pi / V)
    q, _ = np.linalg.qr(rng.randn(V, K))
    evecs = q / np.sqrt(mass)[:, None]
    evals = np.sort(rng.rand(K)) * (0.55 * K)
    evals[0] = 0.0

    # complex gradient operator: off-diagonals random, diagonal = -row sum
    scale = 0.1 * math.sqrt(V)
    off = (rng.randn(V, 6) + 1j * rng.randn(V, 6)) * scale
    diag = -off.sum(1)
    rows = np.repeat(np.arange(V, dtype=np.int64), 7)
    cols = np.concatenate([np.arange(V, dtype=np.int64)[:, None], nbr], 1).reshape(-1)
    vals = np.concatenate([diag[:, None], off], 1).reshape(-1)
    idx = torch.from_numpy(np.stack([rows, cols], 0))
    gradX = torch.sparse_coo_tensor(idx, torch.from_numpy(vals.real.copy()).to(dtype), (V, V)).coalesce()
    gradY = torch.sparse_coo_tensor(idx, torch.from_numpy(vals.imag.copy()).to(dtype), (V, V)).coalesce()

    verts = rng.randn(V, 3)
    verts /= np.linalg.norm(verts, axis=1, keepdims=True)
    verts *= 1.0 + 0.1 * rng.randn(V, 1)
    edges = np.concatenate([np.stack([np.arange(V), nbr[:, 0]], 1),
                            np.stack([np.arange(V), nbr[:, 2]], 1),
                            np.stack([np.arange(V), nbr[:, 4]], 1)], 0).astype(np.int64)
    return {
        "verts": torch.from_numpy(verts).to(dtype),
        "faces": torch.from_numpy(faces),
        "edges": torch.from_numpy(edges),
        "mass": torch.from_numpy(mass).to(dtype),
        "evals": torch.from_numpy(evals).to(dtype),
        "evecs": torch.from_numpy(evecs).to(dtype),
        "gradX": gradX,
        "gradY": gradY,
    }


def randomize_times(state_dict, seed: int = 0, lo: float = 1e-3, hi: float = 0.3):
    """Replace every ``diffusion_time`` (initialised to 0 -> clamped to 1e-8, i.e. an
    identity projection, layers.py:41,49) by U(lo,hi), the range trained models
    occupy.  In place; returns the dict."""
    g = torch.Generator().manual_seed(seed)
    for k, v in state_dict.items():
        if k.endswith("diffusion_time"):
            v.copy_(lo + (hi - lo) * torch.rand(v.shape, generator=g, dtype=v.dtype))
    return state_dict


def sphere_mesh(V: int, seed: int = 0, bump: float = 0.1):
    """A real closed triangle mesh: V-point Fibonacci lattice on the unit sphere triangulated by its convex hull
    (valence ~6), then radially perturbed r = 1 + bump*N(0,1).  Returns (verts [V,3] f64, faces [F,3] i64)."""
    from scipy.spatial import ConvexHull
    i = np.arange(V) + 0.5
    phi = np.arccos(1.0 - 2.0 * i / V)
    theta = math.pi * (1.0 + 5.0 ** 0.5) * i
    pts = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], 1)
    faces = ConvexHull(pts).simplices.astype(np.int64)
    # consistent outward orientation
    c = pts[faces]
    flip = np.einsum("ij,ij->i", np.cross(c[:, 1] - c[:, 0], c[:, 2] - c[:, 0]), c.mean(1)) < 0
    faces[flip] = faces[flip][:, ::-1]
    r = 1.0 + bump * np.random.RandomState(seed).randn(V, 1)
    return pts * r, faces
