"""MeshBatch: device-resident, packed geometry for a ragged batch of meshes.

The reference passes the per-mesh operators of ``get_operators`` (geometry.py:426-570) to every
forward as loose tensors and re-slices the sparse ones inside the block (layers.py:217-220).
Here they are packed once: vertex axes concatenated, gradX/gradY converted from COO to one shared
int32 CSR pattern (plus the CSR of the transposes for backward), and the row-tile / split-V chunk
tables the HIP kernels walk are built on the host.  The struct handed to the C ABI
(``dn_mesh_batch_t``) only holds device pointers into tensors owned by this object.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _hip

_table_cache = {}


def default_chunk_rows(v_total: int) -> int:
    """Rows per split-V chunk: aim for >= ~512 chunks, 128 <= rows <= 1024, multiple of 32."""
    rows = 32 * max(1, -(-v_total // (512 * 32)))
    return int(min(1024, max(128, rows)))


def build_tables(sizes: Sequence[int], chunk_rows: int, tile_rows: int):
    """Host-side tile / chunk tables for meshes of the given vertex counts."""
    tiles, chunks, mco, mrows = [], [], [0], []
    row0 = 0
    for m, v in enumerate(sizes):
        mrows.append((row0, v, m, 0))
        for r in range(0, v, tile_rows):
            tiles.append((row0 + r, min(tile_rows, v - r), m, 0))
        for i, r in enumerate(range(0, v, chunk_rows)):
            chunks.append((row0 + r, min(chunk_rows, v - r), m, i))
        mco.append(len(chunks))
        row0 += v
    as_t = lambda rows: np.array(rows, dtype=np.int32).reshape(-1, 4)
    return as_t(tiles), as_t(chunks), np.array(mco, dtype=np.int32), as_t(mrows)


def _tables_on(device, sizes, chunk_rows):
    tile_rows = _hip.lib().dn_tile_rows()
    key = (str(device), tuple(sizes), chunk_rows, tile_rows)
    hit = _table_cache.get(key)
    if hit is None:
        if len(_table_cache) > 256:
            _table_cache.clear()
        hit = tuple(torch.from_numpy(a).to(device) for a in build_tables(sizes, chunk_rows, tile_rows))
        _table_cache[key] = hit
    return hit


def _csr_from_sorted_rows(rows: torch.Tensor, n_rows: int) -> torch.Tensor:
    counts = torch.bincount(rows, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=rows.device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    return rowptr.to(torch.int32)


class MeshBatch:
    """Packed operators of ``n_mesh`` independent meshes (vertex axes concatenated).

    Build with :meth:`from_operators` (lists of per-mesh tensors as returned by
    ``get_operators``) or :meth:`from_reference_args` (the tensors a reference-style
    ``forward`` receives).  ``rows_only`` gives a geometry-free batch for plain row ops."""

    def __init__(self):
        self.sizes: List[int] = []
        self.device = None
        self.k_eig = 0
        self.mass = self.evals = self.evecs = None
        self.g_rowptr = self.g_col = self.g_vx = self.g_vy = None
        self.gt_rowptr = self.gt_col = self.gt_vx = self.gt_vy = None
        self.tiles = self.chunks = self.mesh_chunk_off = self.mesh_rows = None
        self._struct = None

    # ------------------------------------------------------------------ constructors
    @classmethod
    def rows_only(cls, n_rows: int, device, chunk_rows: Optional[int] = None):
        mb = cls()
        mb.sizes, mb.device = [int(n_rows)], torch.device(device)
        mb._finish(chunk_rows)
        return mb

    @classmethod
    def from_operators(cls, mass, evals, evecs, gradX=None, gradY=None, device=None, chunk_rows=None):
        """mass/evals/evecs/gradX/gradY: lists with one entry per mesh ([V], [K], [V,K], sparse [V,V])."""
        mb = cls()
        device = torch.device(device) if device is not None else evecs[0].device
        mb.device = device
        mb.sizes = [int(e.shape[0]) for e in evecs]
        f32 = lambda t: t.to(device=device, dtype=torch.float32)
        mb.mass = torch.cat([f32(m).reshape(-1) for m in mass]).contiguous()
        mb.evecs = torch.cat([f32(e) for e in evecs], 0).contiguous()
        mb.evals = torch.stack([f32(e) for e in evals], 0).contiguous()
        mb.k_eig = int(mb.evecs.shape[1])
        if gradX is not None:
            offs = np.concatenate([[0], np.cumsum(mb.sizes)])
            rows, cols, vx, vy = [], [], [], []
            for i, (gx, gy) in enumerate(zip(gradX, gradY)):
                r, c, a, b = _shared_pattern(gx.to(device), gy.to(device))
                rows.append(r + int(offs[i])); cols.append(c + int(offs[i])); vx.append(a); vy.append(b)
            mb._set_grad(torch.cat(rows), torch.cat(cols), torch.cat(vx), torch.cat(vy))
        mb._finish(chunk_rows)
        return mb

    @classmethod
    def from_reference_args(cls, mass, evals, evecs, gradX=None, gradY=None, chunk_rows=None):
        """Batched tensors exactly as ``DiffusionNet.forward`` holds them after adding the batch
        dimension (layers.py:346-358): mass [B,V], evals [B,K], evecs [B,V,K], gradX/gradY sparse [B,V,V]."""
        mb = cls()
        B, V, K = evecs.shape
        device = evecs.device
        mb.device, mb.sizes, mb.k_eig = device, [int(V)] * int(B), int(K)
        mb.mass = mass.to(torch.float32).reshape(-1).contiguous()
        mb.evecs = evecs.to(torch.float32).reshape(B * V, K).contiguous()
        mb.evals = evals.to(torch.float32).reshape(B, K).contiguous()
        if gradX is not None:
            r, c, a, b = _shared_pattern(gradX, gradY)
            mb._set_grad(r, c, a, b)
        mb._finish(chunk_rows)
        return mb

    # ------------------------------------------------------------------ internals
    def _set_grad(self, rows, cols, vx, vy):
        """rows/cols: global int64 COO (row-sorted, coalesced); builds CSR and CSR of the transpose."""
        vt = sum(self.sizes)
        self.g_rowptr = _csr_from_sorted_rows(rows, vt)
        self.g_col = cols.to(torch.int32).contiguous()
        self.g_vx, self.g_vy = vx.to(torch.float32).contiguous(), vy.to(torch.float32).contiguous()
        perm = torch.argsort(cols, stable=True)
        self.gt_rowptr = _csr_from_sorted_rows(cols[perm], vt)
        self.gt_col = rows[perm].to(torch.int32).contiguous()
        self.gt_vx, self.gt_vy = self.g_vx[perm].contiguous(), self.g_vy[perm].contiguous()

    def _finish(self, chunk_rows):
        vt = sum(self.sizes)
        self.chunk_rows = int(chunk_rows or default_chunk_rows(vt))
        self.tiles, self.chunks, self.mesh_chunk_off, self.mesh_rows = _tables_on(self.device, self.sizes, self.chunk_rows)
        s = _hip.MeshBatchStruct()
        s.n_mesh, s.v_total, s.k_eig = len(self.sizes), vt, self.k_eig
        s.n_tiles, s.n_chunks = int(self.tiles.shape[0]), int(self.chunks.shape[0])
        s.g_nnz = int(self.g_col.shape[0]) if self.g_col is not None else 0
        for name in ("tiles", "chunks", "mesh_chunk_off", "mesh_rows", "mass", "evals", "evecs",
                     "g_rowptr", "g_col", "g_vx", "g_vy", "gt_rowptr", "gt_col", "gt_vx", "gt_vy"):
            setattr(s, name, _hip.ptr(getattr(self, name)))
        self._struct = s

    # ------------------------------------------------------------------ accessors
    @property
    def v_total(self):
        return sum(self.sizes)

    @property
    def n_mesh(self):
        return len(self.sizes)

    @property
    def has_grad(self):
        return self.g_rowptr is not None

    def ref(self):
        return C.byref(self._struct)


def _shared_pattern(gx: torch.Tensor, gy: torch.Tensor):
    """COO (global, row-sorted) index + value arrays of gradX/gradY on ONE pattern.

    Accepts 2-D [V,V] or 3-D [B,V,V] sparse COO tensors.  If the two index sets differ (never the
    case for operators built by geometry.py:375-382, but the API allows it) the union pattern is
    used with explicit zeros."""
    gx = gx.coalesce() if not gx.is_coalesced() else gx
    gy = gy.coalesce() if not gy.is_coalesced() else gy
    ix, iy = gx.indices(), gy.indices()
    same = ix.shape == iy.shape and bool(torch.equal(ix, iy))
    if same:
        idx, vx, vy = ix, gx.values(), gy.values()
    else:
        zx = torch.sparse_coo_tensor(iy, torch.zeros_like(gy.values()), gy.shape)
        zy = torch.sparse_coo_tensor(ix, torch.zeros_like(gx.values()), gx.shape)
        ux, uy = (gx + zx).coalesce(), (gy + zy).coalesce()
        idx, vx, vy = ux.indices(), ux.values(), uy.values()
    if idx.shape[0] == 3:
        V = gx.shape[-1]
        rows, cols = idx[0] * V + idx[1], idx[0] * V + idx[2]
    else:
        rows, cols = idx[0], idx[1]
    return rows, cols, vx, vy


class GatherPattern:
    """CSR of a row-gather matrix (faces / edges -> vertices) and of its transpose, for the
    gather-mean output remaps of layers.py:379-391."""

    def __init__(self, index: torch.Tensor, n_src_rows: int):
        # index: [n_out, n_per] int64, global row ids into the [n_src_rows, C] source
        n_out, n_per = index.shape
        dev = index.device
        flat = index.reshape(-1)
        self.n_out, self.n_per, self.n_src = int(n_out), int(n_per), int(n_src_rows)
        self.rowptr = torch.arange(0, n_out * n_per + 1, n_per, dtype=torch.int32, device=dev)
        self.col = flat.to(torch.int32).contiguous()
        perm = torch.argsort(flat, stable=True)
        self.t_rowptr = _csr_from_sorted_rows(flat[perm], n_src_rows)
        self.t_col = (perm // n_per).to(torch.int32).contiguous()
