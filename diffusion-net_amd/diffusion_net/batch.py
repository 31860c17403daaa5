"""MeshBatch: device-resident, packed geometry for a ragged batch of meshes.

The reference passes the per-mesh operators of ``get_operators`` (geometry.py:426-570) to every
forward as loose tensors and re-slices the sparse ones inside the block (layers.py:217-220).
Here they are packed once: vertex axes concatenated, gradX/gradY converted from COO to one shared
int32 CSR pattern (plus the CSR of the transposes for backward), and the row-tile / split-V chunk
tables the HIP kernels walk are built on the host.  The struct handed to the C ABI
(``dn_mesh_batch_t``) only holds device pointers into tensors owned by this object.
"""
from __future__ import annotations

import collections
import ctypes as C
import weakref
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _hip

_table_cache = {}

# integer handles of the packed objects, for the torch.library ops (torchlib.py): custom operators take tensors and plain numbers only
_handles = weakref.WeakValueDictionary()
_next_handle = [1]


def register_handle(obj) -> int:
    k = _next_handle[0]
    _next_handle[0] += 1
    _handles[k] = obj
    return k


def handle_object(k: int):
    if not k:
        return None
    try:
        return _handles[int(k)]
    except KeyError:
        raise RuntimeError(f"diffusion_net: handle {int(k)} names a packed object (MeshBatch / BlockConfig / GatherPattern) that has been "
                           "garbage-collected; the caller owns these objects and must keep them alive while an op that received the handle "
                           "-- or the backward of one -- can still run") from None


# strong references for the lifetime of an autograd node (torchlib.py: the eager autograd formulas of the custom ops pin the objects their
# backward will look up, so that dropping ``mb`` between forward and backward is harmless, as it is on the direct path)
_pins = {}


def pin_handle(k: int) -> None:
    if k:
        ent = _pins.get(int(k))
        if ent is None:
            _pins[int(k)] = [handle_object(k), 1]
        else:
            ent[1] += 1


def unpin_handle(k: int) -> None:
    ent = _pins.get(int(k)) if k else None
    if ent is not None:
        ent[1] -= 1
        if ent[1] <= 0:
            del _pins[int(k)]


def default_chunk_rows(v_total: int) -> int:
    """Rows per split-V chunk when a fixed size is asked for: aim for >= ~512 chunks, 128 <= rows <= 1024, multiple of 32."""
    rows = 32 * max(1, -(-v_total // (512 * 32)))
    return int(min(1024, max(128, rows)))


def balanced_chunk_rows(sizes: Sequence[int], target: int, min_rows: int = 128) -> List[int]:
    """Rows per chunk FOR EVERY MESH such that the batch has (at most, and if the meshes allow exactly) ``target`` chunks of nearly equal
    size: the split-V kernels run one workgroup per chunk (per-mesh sums) on ``target`` = one workgroup slot per CU, so 514 chunks on
    256 slots is three rounds of which the last holds two workgroups (+50 % on that launch), 256 chunks is one round.  Chunk counts
    are dealt to the meshes in proportion to their vertex counts (largest remainders first)."""
    n, vt = len(sizes), sum(sizes)
    if n == 0 or vt == 0:
        return [min_rows] * n
    caps = [max(1, v // min_rows) for v in sizes]
    quotas = [target * v / vt for v in sizes]
    counts = [min(cap, max(1, int(q))) for q, cap in zip(quotas, caps)]
    rem = target - sum(counts)
    order = sorted(range(n), key=lambda i: -(quotas[i] - int(quotas[i])))
    progressed = True
    while rem > 0 and progressed:
        progressed = False
        for j in order:
            if rem > 0 and counts[j] < caps[j]:
                counts[j] += 1
                rem -= 1
                progressed = True
    return [32 * max(1, -(-v // (c * 32))) for v, c in zip(sizes, counts)]


def build_tables(sizes: Sequence[int], chunk_rows, tile_rows: int):
    """Host-side tile / chunk tables for meshes of the given vertex counts.  chunk_rows: one int, or one per mesh."""
    per_mesh = list(chunk_rows) if isinstance(chunk_rows, (list, tuple)) else [int(chunk_rows)] * len(sizes)
    tiles, chunks, mco, mrows = [], [], [0], []
    row0 = 0
    for m, v in enumerate(sizes):
        mrows.append((row0, v, m, 0))
        for r in range(0, v, tile_rows):
            tiles.append((row0 + r, min(tile_rows, v - r), m, 0))
        for i, r in enumerate(range(0, v, per_mesh[m])):
            chunks.append((row0 + r, min(per_mesh[m], v - r), m, i))
        mco.append(len(chunks))
        row0 += v
    as_t = lambda rows: np.array(rows, dtype=np.int32).reshape(-1, 4)
    return as_t(tiles), as_t(chunks), np.array(mco, dtype=np.int32), as_t(mrows)


def diffusion_plan(sizes: Sequence[int], n_groups: int = 0):
    """Work plan of the one-launch diffusion operator (dn_diffuse.hip) for meshes of the given vertex counts: which rows of which mesh every
    workgroup owns in each mesh group.  The arithmetic is the library's (``dn_diffusion_plan``, host side).  Returns (plan [groups * n_wg, 4]
    int32, n_wg, groups) or None when the batch is not plannable."""
    L = _hip.lib()
    n_wg = int(L.dn_diffusion_plan_wgs())
    arr = np.ascontiguousarray(np.asarray(sizes, dtype=np.int32))
    plan = np.zeros((_hip.DIFFUSION_MAX_GROUPS * n_wg, 4), dtype=np.int32)
    g = int(L.dn_diffusion_plan(arr.ctypes.data, len(arr), n_wg, int(n_groups), plan.ctypes.data))
    if g <= 0:
        return None
    return plan[: g * n_wg].copy(), n_wg, g


def _plan_on(device, sizes, n_groups):
    key = ("df_plan", str(device), tuple(sizes), int(n_groups), _hip.get_option("diffuse_groups") if not n_groups else 0)
    hit = _table_cache.get(key)
    if hit is None:
        if len(_table_cache) > 256:
            _table_cache.clear()
        p = diffusion_plan(sizes, n_groups)
        hit = (torch.from_numpy(p[0]).to(device), p[1], p[2]) if p is not None else (None, 0, 0)
        _table_cache[key] = hit
    return hit


def _sg_units_on(device, sizes, k_eig):
    """Unit table of the spectral-gradient operands (``dn_spectral_units``, host arithmetic in the library): runs of <= 64 (128 at K = 256) rows of one mesh."""
    key = ("sg_units", str(device), tuple(sizes), int(k_eig))
    hit = _table_cache.get(key)
    if hit is None:
        if len(_table_cache) > 256:
            _table_cache.clear()
        L = _hip.lib()
        arr = np.ascontiguousarray(np.asarray(sizes, dtype=np.int32))
        n = int(L.dn_spectral_units(arr.ctypes.data, len(arr), int(k_eig), None))
        units = np.zeros((max(n, 1), 4), dtype=np.int32)
        L.dn_spectral_units(arr.ctypes.data, len(arr), int(k_eig), units.ctypes.data)
        hit = (torch.from_numpy(units[:n].copy()).to(device), n)
        _table_cache[key] = hit
    return hit


# False: batches are packed without the spectral-gradient operands (3 x 4 V K bytes per mesh) and the block forward keeps the back-projection
# launch + CSR gather (the library option "spectral_grad" = 0 selects that at run time for batches that do carry them)
spectral_grad = True
# True: also for k_eig = 256 batches (the two-launch form of C = K = 256; 3 x 4 V K bytes = 614 MB for one 200k-vertex mesh).  Off by default: that
# form measures slower than back-projection + gather (BASELINE config 4: 27.1 against 29.1 M vertices/s) and is only taken with the library
# option "spectral_grad" = 2; the tests switch this on.
spectral_grad_wide = False


def _tables_on(device, sizes, chunk_rows):
    tile_rows = _hip.lib().dn_tile_rows()
    key = (str(device), tuple(sizes), tuple(chunk_rows) if isinstance(chunk_rows, (list, tuple)) else chunk_rows, tile_rows)
    hit = _table_cache.get(key)
    if hit is None:
        if len(_table_cache) > 256:
            _table_cache.clear()
        hit = tuple(torch.from_numpy(a).to(device) for a in build_tables(sizes, chunk_rows, tile_rows))
        _table_cache[key] = hit
    return hit


def coo_to_csr(rows, row_div, cols, vx, vy, n_rows, n_cols):
    """int64 COO -> int32 CSR of the pattern + CSR of its transpose, on the device (``dn_coo_to_csr_i64``, dn_pack.hip).
    rows: [nnz] non-decreasing int64 row ids, or None: entry j belongs to row j // row_div.  Raises ValueError on an index outside
    the operator (the one host synchronisation of a pack; the reference's ``torch.gather`` / sparse mm would fault there too)."""
    if n_rows >= 2 ** 31 - 1 or n_cols >= 2 ** 31 - 1 or cols.numel() >= 2 ** 31 - 1:
        raise ValueError("operator too large for int32 CSR indices")
    _hip.require_device(cols)
    L, dev, nnz = _hip.lib(), cols.device, int(cols.numel())
    cols = cols.to(torch.int64).contiguous()
    rows = rows.to(torch.int64).contiguous() if rows is not None else None
    i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)
    f32 = lambda v: torch.empty(nnz, dtype=torch.float32, device=dev) if v is not None else None
    vx = vx.to(torch.float32).contiguous() if vx is not None else None
    vy = vy.to(torch.float32).contiguous() if vy is not None else None
    rowptr, col, t_rowptr, t_col, t_vx, t_vy = i32(n_rows + 1), i32(nnz), i32(n_cols + 1), i32(nnz), f32(vx), f32(vy)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    ws = _hip.workspace(dev, L.dn_coo_to_csr_workspace_bytes(nnz, n_cols))
    _hip.check(L.dn_coo_to_csr_i64(_hip.ptr(rows), int(row_div), cols.data_ptr(), _hip.ptr(vx), _hip.ptr(vy), nnz, n_rows, n_cols,
                                   rowptr.data_ptr(), col.data_ptr(), t_rowptr.data_ptr(), t_col.data_ptr(), _hip.ptr(t_vx), _hip.ptr(t_vy),
                                   status.data_ptr(), ws.data_ptr(), ws.numel(), _hip.stream_of(cols)), "dn_coo_to_csr_i64")
    if int(status.item()) != 0:
        raise ValueError("sparse operator / gather index out of range (rows must be sorted, 0 <= index < %d x %d)" % (n_rows, n_cols))
    return rowptr, col, vx, vy, t_rowptr, t_col, t_vx, t_vy


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
        self.handle = register_handle(self)
        self.amax = None          # [3] device floats: max |evecs|, max |mass|, ||[gradX; gradY]||_inf (magnitudes for the split-fp16 engine, dn_api.hip)
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
        (self.g_rowptr, self.g_col, self.g_vx, self.g_vy,
         self.gt_rowptr, self.gt_col, self.gt_vx, self.gt_vy) = coo_to_csr(rows, 1, cols, vx, vy, vt, vt)

    def _finish(self, chunk_rows):
        vt = sum(self.sizes)
        if vt >= 2 ** 31 - 1:
            raise ValueError("more than 2^31 vertices in one batch")
        # default: as many chunks as the split-V kernels have workgroup slots, nearly equal in size (balanced_chunk_rows); an explicit
        # chunk_rows (tests) gives fixed-size chunks
        self.chunk_rows = int(chunk_rows) if chunk_rows else balanced_chunk_rows(self.sizes, _hip.lib().dn_tn_target_chunks_k(int(self.k_eig)))
        self.tiles, self.chunks, self.mesh_chunk_off, self.mesh_rows = _tables_on(self.device, self.sizes, self.chunk_rows)
        s = _hip.MeshBatchStruct()
        s.n_mesh, s.v_total, s.k_eig = len(self.sizes), vt, self.k_eig
        s.n_tiles, s.n_chunks = int(self.tiles.shape[0]), int(self.chunks.shape[0])
        s.g_nnz = int(self.g_col.shape[0]) if self.g_col is not None else 0
        for name in ("tiles", "chunks", "mesh_chunk_off", "mesh_rows", "mass", "evals", "evecs",
                     "g_rowptr", "g_col", "g_vx", "g_vy", "gt_rowptr", "gt_col", "gt_vx", "gt_vy"):
            setattr(s, name, _hip.ptr(getattr(self, name)))
        # the one-launch diffusion operator's work plan (K = C = 128 batches; the library ignores it otherwise)
        self.df_plan = None
        if self.k_eig == 128 and self.evecs is not None and vt > 0:
            self.df_plan, s.df_n_wg, s.df_n_groups = _plan_on(self.device, self.sizes, 0)
            s.df_plan = _hip.ptr(self.df_plan)
            s.df_v_total = vt            # the plan's stamp: the library ignores a plan made for another batch
        if self.evecs is not None and self.evecs.numel() > 0 and self.mass is not None:
            # once per packed batch (two small reductions, no host synchronisation)
            words = [self.evecs.detach().abs().amax(), self.mass.detach().abs().amax()]
            if self.g_rowptr is not None and self.g_vx is not None and self.g_vx.numel() > 0:
                # infinity norm of the stacked gradient operators: max |gradX x|, |gradY x| <= it * max |x|
                # row sums of |gradX|, |gradY| in entry order with the library's CSR gather (one lane group per row, fixed order: the word
                # is bitwise reproducible -- index_add_'s float atomics were the one non-deterministic reduction left; the word
                # feeds a power-of-two scale, so a sum straddling 2^k could change the rounding of a product from run to run)
                nnz = int(self.g_vx.numel())
                ident = torch.arange(nnz, dtype=torch.int32, device=self.device)
                rs = torch.empty(2, vt, dtype=torch.float32, device=self.device)
                for k, v in enumerate((self.g_vx, self.g_vy)):
                    a = v.detach().abs().contiguous()
                    _hip.check(_hip.lib().dn_csr_mean_f32(self.g_rowptr.data_ptr(), ident.data_ptr(), vt, a.data_ptr(), 1, 1.0, rs[k].data_ptr(),
                                                          _hip.stream_of(a)), "dn_csr_mean_f32")
                words.append(rs.amax())
            else:
                words.append(torch.zeros((), dtype=torch.float32, device=self.device))
            self.amax = torch.stack(words).to(torch.float32).contiguous()
            s.evecs_amax, s.mass_amax = self.amax.data_ptr(), self.amax.data_ptr() + 4
            if self.g_rowptr is not None:
                s.grad_norm = self.amax.data_ptr() + 8
        # spectral-gradient operands (dn_spectral.hip): [evecs | gradX evecs | gradY evecs] as pre-split operand fragments -- with them the block
        # forward computes xd, gx, gy inside its chained row kernel.  Once per packed batch: two launches, no host synchronisation.
        self.sg_pack = self.sg_units = self.sg_amax = None
        L = _hip.lib()
        if spectral_grad and self.g_rowptr is not None and self.evecs is not None and vt > 0 and s.g_nnz > 0 \
                and (L.dn_spectral_grad_supported(self.k_eig, 128) or (spectral_grad_wide and L.dn_spectral_grad_supported(self.k_eig, 256))):
            self.sg_units, n_units = _sg_units_on(self.device, self.sizes, self.k_eig)
            self.sg_pack = torch.empty(int(L.dn_spectral_pack_bytes(n_units, self.k_eig)), dtype=torch.uint8, device=self.device)
            self.sg_amax = torch.empty(4 * len(self.sizes), dtype=torch.float32, device=self.device)
            ws = _hip.workspace(self.device, L.dn_spectral_pack_workspace_bytes(C.byref(s)))
            _hip.check(L.dn_spectral_pack_f32(C.byref(s), self.sg_units.data_ptr(), n_units, self.sg_pack.data_ptr(), self.sg_amax.data_ptr(),
                                              ws.data_ptr(), ws.numel(), _hip.stream_of(self.evecs)), "dn_spectral_pack_f32")
            s.sg_pack, s.sg_units, s.sg_amax, s.sg_n_units = self.sg_pack.data_ptr(), self.sg_units.data_ptr(), self.sg_amax.data_ptr(), n_units
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
        self.handle = register_handle(self)
        self.n_out, self.n_per, self.n_src = int(n_out), int(n_per), int(n_src_rows)
        self.rowptr, self.col, _, _, self.t_rowptr, self.t_col, _, _ = coo_to_csr(None, n_per, index.reshape(-1), None, None,
                                                                                  self.n_out, self.n_src)


def _storage_bytes(tensors, seen):
    """Bytes of the distinct storages behind ``tensors`` that are not in ``seen`` yet (views share their storage)."""
    total = 0
    for t in tensors:
        if t is None:
            continue
        parts = (t._indices(), t._values()) if t.is_sparse else (t,)
        for u in parts:
            st = u.untyped_storage()
            key = (u.device, st.data_ptr())
            if key not in seen:
                seen.add(key)
                total += int(st.nbytes())
    return total


def _entry_tensors(value):
    out = []
    for obj in value:
        if obj is None:
            continue
        for v in vars(obj).values():
            if isinstance(v, torch.Tensor):
                out.append(v)
    return out


def content_checksum(operands) -> bytes:
    """128-bit checksum over EVERY byte of every operand (``dn_checksum128``, dn_pack.hip): one streaming kernel per buffer, all
    accumulated on the device, then ONE 16-byte device-to-host copy -- the only host synchronisation of a content lookup."""
    L = _hip.lib()
    dev = next(t.device for t in operands if t is not None)
    for t in operands:        # the kernel dereferences every operand's pointer on `dev`: a host tensor among device operands would fault
        if t is not None and t.device != dev:
            raise RuntimeError("operator cache: operands on different devices (%s and %s); move every operator the forward uses to the "
                               "model's device" % (dev, t.device))
    acc = torch.zeros(2, dtype=torch.int64, device=dev)
    keep, ptrs, sizes, salts = [], [], [], []
    for slot, t in enumerate(operands):
        if t is None:
            continue
        parts = (t._indices(), t._values()) if t.is_sparse else (t,)
        for k, u in enumerate(parts):
            u = u.contiguous()
            if u.element_size() % 4 != 0:                 # bool / uint8 / fp16 operands: widen (never the case for the reference's operands)
                u = u.to(torch.int32 if not u.is_floating_point() else torch.float32)
            keep.append(u)
            ptrs.append(u.data_ptr()); sizes.append(u.numel() * u.element_size()); salts.append(4 * slot + k + 1)
    stream = _hip.stream_of(keep[0]) if keep else None
    for lo in range(0, len(keep), 16):                    # one launch per 16 buffers (a mesh has eight)
        n = min(16, len(keep) - lo)
        _hip.check(L.dn_checksum128_multi(n, (C.c_void_p * n)(*ptrs[lo:lo + n]), (C.c_size_t * n)(*sizes[lo:lo + n]),
                                          (C.c_uint64 * n)(*salts[lo:lo + n]), acc.data_ptr(), stream), "dn_checksum128_multi")
    return acc.cpu().numpy().tobytes()


class _CacheEntry:
    __slots__ = ("value", "operands", "versions", "nbytes", "device", "kid", "kfp", "aliases")


class OperatorCache:
    """Device-resident operator cache behind the REFERENCE signature.  The experiment scripts hand the same meshes'
    operators to ``forward`` again and again -- and move them to the device anew every step
    (human_segmentation_original.py:111-120) -- so re-packing them per call (COO -> CSR, transpose, tile tables) made the
    unmodified loop host-bound.  Two levels, both EXACT:
      1. identity: (device, dtype, data_ptr, version counter, shape) of every operand.  Hits when the caller keeps its device
         tensors; no kernel, no host synchronisation.  The entry holds references to the keyed tensors, so an address cannot be
         recycled under a live key, and an in-place edit bumps the version counter and misses.
      2. content: shapes / dtypes / device plus a 128-bit checksum over every byte of every operand (mass, evals, evecs, the index
         AND value arrays of gradX and gradY, faces / edges), computed on the device (``content_checksum``).  Hits when the caller
         re-uploads the same mesh.  Costs one 16-byte device-to-host copy -- the one host synchronisation of this path, taken only
         when identity missed, i.e. after the caller's own ten synchronous host-to-device uploads.  An entry found by content is
         only served while the tensors it aliases still carry the version counters they had when it was built.
         ``operator_cache.fingerprint = False`` relies on identity only; ``operator_cache.enabled = False`` packs on every call.
    One LRU over both levels, bounded by entries and by the bytes of device memory the entries keep alive (every tensor of the packed
    operators and every caller tensor pinned by the key), by default half of the device's memory; cleared and retried once if packing
    runs out of memory."""

    def __init__(self, max_entries=8192, max_bytes=None, mem_fraction=0.5):
        self.enabled, self.fingerprint = True, True
        self.max_entries, self.max_bytes, self.mem_fraction = max_entries, max_bytes, mem_fraction
        self._by_id = {}
        self._by_fp = {}
        self._alias = {}                             # identity key of a re-uploaded copy -> (entry, weak references to its tensors)
        self._lru = collections.OrderedDict()        # id(entry) -> entry, least recently used first
        self._bytes = collections.defaultdict(int)   # per device
        self._budget = {}
        self.hits_id = self.hits_fp = self.misses = 0

    @staticmethod
    def _ident(t):
        if t is None:
            return None
        if t.is_sparse:
            i, v = t._indices(), t._values()
            return ("coo", str(t.device), v.dtype, i.data_ptr(), i._version, v.data_ptr(), v._version, tuple(t.shape), int(v.shape[0]))
        return (str(t.device), t.dtype, t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()))

    @staticmethod
    def _versions(operands):
        out = []
        for t in operands:
            if t is None:
                out.append(None)
            elif t.is_sparse:
                out.append((t._indices()._version, t._values()._version))
            else:
                out.append(t._version)
        return tuple(out)

    @staticmethod
    def _shape_key(t):
        if t is None:
            return None
        if t.is_sparse:
            return ("coo", tuple(t.shape), t._values().dtype, int(t._values().shape[0]))
        return (tuple(t.shape), t.dtype)

    def budget(self, device):
        if self.max_bytes is not None:
            return self.max_bytes
        key = str(device)
        b = self._budget.get(key)
        if b is None:
            b = 64 << 30
            if torch.device(device).type == "cuda":
                try:
                    b = int(self.mem_fraction * torch.cuda.mem_get_info(device)[1])
                except Exception:      # noqa: BLE001
                    pass
            self._budget[key] = b
        return b

    def bytes_held(self, device=None):
        return sum(self._bytes.values()) if device is None else self._bytes[str(device)]

    def __len__(self):
        return len(self._lru)

    def clear(self):
        self._by_id.clear()
        self._by_fp.clear()
        self._alias.clear()
        self._lru.clear()
        self._bytes.clear()

    def _drop(self, e):
        self._lru.pop(id(e), None)
        if e.kid is not None and self._by_id.get(e.kid) is e:
            del self._by_id[e.kid]
        if e.kfp is not None and self._by_fp.get(e.kfp) is e:
            del self._by_fp[e.kfp]
        for k in list(e.aliases):
            self._alias.pop(k, None)
        e.aliases.clear()
        self._bytes[e.device] -= e.nbytes

    def _trim(self, device):
        dev, budget = str(device), self.budget(device)
        while len(self._lru) > max(1, self.max_entries):
            self._drop(next(iter(self._lru.values())))
        if self._bytes[dev] > budget:
            newest = next(reversed(self._lru.values()))
            for e in list(self._lru.values()):           # least recently used first; the entry just built always stays
                if self._bytes[dev] <= budget or e is newest:
                    break
                if e.device == dev:
                    self._drop(e)

    def _remember_alias(self, kid, e, operands):
        """A re-uploaded copy that was found by content: remember its IDENTITY as well, so that a caller who keeps these tensors gets the
        synchronisation-free path from the second call on.  Only weak references are held (the scripts upload fresh copies every step:
        pinning them would leak a mesh per step); the alias disappears with the first of its tensors -- a recycled address can
        therefore never match a stale key -- and with its entry."""
        if len(self._alias) > 4 * self.max_entries:
            return
        def gone(_ref, kid=kid, cache=weakref.ref(self)):
            c = cache()
            if c is not None:
                hit = c._alias.pop(kid, None)
                if hit is not None:
                    hit[0].aliases.discard(kid)
        try:
            refs = [weakref.ref(t, gone) for t in operands if t is not None]
        except TypeError:
            return
        self._alias[kid] = (e, refs)
        e.aliases.add(kid)

    def lookup(self, mass, evals, evecs, gradX, gradY, index, tag, build, key_operands=None):
        """build() -> (MeshBatch, GatherPattern or None).  Tensors are the batched reference operands; ``tag`` the static part;
        ``key_operands``: the tensors as the caller handed them over (``unsqueeze`` of a sparse tensor copies it), for both keys."""
        if not self.enabled:
            return build()
        operands = tuple(key_operands) if key_operands is not None else (mass, evals, evecs, gradX, gradY, index)
        kid = (tag,) + tuple(self._ident(t) for t in operands)
        e = self._by_id.get(kid)
        if e is None:
            hit = self._alias.get(kid)
            e = hit[0] if hit is not None and self._versions(hit[0].operands) == hit[0].versions else None
        if e is not None:
            self._lru.move_to_end(id(e))
            self.hits_id += 1
            return e.value
        kfp = None
        device = next((t.device for t in operands if t is not None), None)
        if self.fingerprint and device is not None and (device.type == "cuda" or _hip._allow_host_tensors):
            kfp = (tag, str(device)) + tuple(self._shape_key(t) for t in operands) + (content_checksum(operands),)
            e = self._by_fp.get(kfp)
            if e is not None:
                if self._versions(e.operands) == e.versions:
                    self._lru.move_to_end(id(e))
                    self.hits_fp += 1
                    self._remember_alias(kid, e, operands)
                    return e.value
                self._drop(e)       # the tensors it aliases were edited in place since: its contents no longer match its key
        self.misses += 1
        try:
            value = build()
        except torch.cuda.OutOfMemoryError:
            self.clear()
            torch.cuda.empty_cache()
            value = build()
        e = _CacheEntry()
        e.value, e.operands, e.versions, e.kid, e.kfp = value, operands, self._versions(operands), kid, kfp
        e.device, e.aliases = str(device), set()
        e.nbytes = _storage_bytes(_entry_tensors(value) + [t for t in operands if t is not None], set())
        old = self._by_id.get(kid)
        if old is not None:
            self._drop(old)
        self._by_id[kid] = e
        if kfp is not None:
            self._by_fp[kfp] = e
        self._lru[id(e)] = e
        self._bytes[e.device] += e.nbytes
        self._trim(device)
        return value


operator_cache = OperatorCache()
