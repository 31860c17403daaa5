"""ctypes binding of libdiffnet_hip.so (C ABI: include/diffnet_hip.h).

The library is the product: there is no Python/torch fallback for the hot path.
If the shared object is missing or the tensors are not on a ROCm device every
op raises -- loudly -- instead of silently computing somewhere else.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

MAX_MLP = 8
BLOCK_AMAX_WORDS = 16       # include/diffnet_hip.h: DN_BLOCK_AMAX_WORDS
HEAD_MAX_CLASSES = 2048   # dn_head.hip: 64 lanes x DN_HEAD_CPL classes per row
_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# DN_LIB_VARIANT=<tag> selects libdiffnet_hip_<tag>.so (same sources, other build flags) for A/B experiments
_VARIANT = os.environ.get("DN_LIB_VARIANT", "")
LIB_PATH = os.path.join(_PKG_DIR, "libdiffnet_hip%s.so" % (("_" + _VARIANT) if _VARIANT else ""))

TILE_DTYPE = np.dtype([("row0", "<i4"), ("nrows", "<i4"), ("mesh", "<i4"), ("aux", "<i4")])

_vp = C.c_void_p


class MeshBatchStruct(C.Structure):
    _fields_ = [
        ("n_mesh", C.c_int32), ("v_total", C.c_int32), ("k_eig", C.c_int32),
        ("n_tiles", C.c_int32), ("n_chunks", C.c_int32), ("g_nnz", C.c_int32),
        ("tiles", _vp), ("chunks", _vp), ("mesh_chunk_off", _vp), ("mesh_rows", _vp),
        ("mass", _vp), ("evals", _vp), ("evecs", _vp),
        ("g_rowptr", _vp), ("g_col", _vp), ("g_vx", _vp), ("g_vy", _vp),
        ("gt_rowptr", _vp), ("gt_col", _vp), ("gt_vx", _vp), ("gt_vy", _vp),
        ("evecs_amax", _vp), ("mass_amax", _vp), ("grad_norm", _vp),
        ("df_plan", _vp), ("df_n_wg", C.c_int32), ("df_n_groups", C.c_int32),
        ("sg_pack", _vp), ("sg_units", _vp), ("sg_amax", _vp), ("sg_n_units", C.c_int32),
        ("df_v_total", C.c_int32),
    ]


class BlockParamsStruct(C.Structure):
    _fields_ = [
        ("C", C.c_int32), ("n_mlp", C.c_int32), ("with_grad", C.c_int32), ("with_rot", C.c_int32),
        ("widths", C.c_int32 * (MAX_MLP + 1)),
        ("time", _vp), ("A_re", _vp), ("A_im", _vp),
        ("W", _vp * MAX_MLP), ("b", _vp * MAX_MLP), ("mask", _vp * MAX_MLP), ("drop_seed", C.c_uint64), ("drop_seed_dev", _vp),
        ("x_amax", _vp), ("out_amax", _vp), ("clamp_time", C.c_int32), ("flags", C.c_uint32),
    ]


class BlockSavedStruct(C.Structure):
    _fields_ = [
        ("xs", _vp), ("xd", _vp), ("gx", _vp), ("gy", _vp), ("g", _vp), ("bre", _vp), ("bim", _vp),
        ("h", _vp * MAX_MLP), ("amax", _vp),
    ]


class BlockGradsStruct(C.Structure):
    _fields_ = [
        ("d_x", _vp), ("d_time", _vp), ("dA_re", _vp), ("dA_im", _vp),
        ("dW", _vp * MAX_MLP), ("db", _vp * MAX_MLP), ("d_out_amax", _vp), ("d_x_amax", _vp),
    ]


_P = C.POINTER
_SIGNATURES = {
    # name: (restype, argtypes)
    "dn_version": (C.c_int, []),
    "dn_tile_rows": (C.c_int, []),
    "dn_tn_target_chunks": (C.c_int, []),
    "dn_tn_target_chunks_k": (C.c_int, [C.c_int]),
    "dn_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "dn_get_option": (C.c_int, [C.c_char_p, _P(C.c_int)]),
    "dn_diffusion_plan_wgs": (C.c_int, []),
    "dn_diffusion_plan": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    "dn_prof_enable": (C.c_int, [C.c_int]),
    "dn_prof_reset": (C.c_int, []),
    "dn_prof_read": (C.c_int, [C.c_int, _P(C.c_double)]),
    "dn_prof_kind_name": (C.c_char_p, [C.c_int]),
    "dn_to_basis_workspace_bytes": (C.c_size_t, [_P(MeshBatchStruct), C.c_int]),
    "dn_to_basis_f32": (C.c_int, [_P(MeshBatchStruct), _vp, C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "dn_from_basis_f32": (C.c_int, [_P(MeshBatchStruct), _vp, C.c_int, C.c_int, _vp, _vp]),
    "dn_diffusion_workspace_bytes": (C.c_size_t, [_P(MeshBatchStruct), C.c_int]),
    "dn_diffusion_fwd_f32": (C.c_int, [_P(MeshBatchStruct), _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp]),
    "dn_diffusion_bwd_f32": (C.c_int, [_P(MeshBatchStruct), _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "dn_spectral_grad_supported": (C.c_int, [C.c_int, C.c_int]),
    "dn_spectral_units": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "dn_spectral_pack_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "dn_spectral_pack_workspace_bytes": (C.c_size_t, [_P(MeshBatchStruct)]),
    "dn_spectral_pack_f32": (C.c_int, [_P(MeshBatchStruct), _vp, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp]),
    "dn_grad_apply_fwd_f32": (C.c_int, [_P(MeshBatchStruct), _vp, C.c_int, _vp, _vp, _vp]),
    "dn_grad_apply_bwd_f32": (C.c_int, [_P(MeshBatchStruct), _vp, _vp, _vp, C.c_int, _vp, _vp]),
    "dn_gradfeat_workspace_bytes": (C.c_size_t, [_P(MeshBatchStruct), C.c_int]),
    "dn_gradfeat_fwd_f32": (C.c_int, [_P(MeshBatchStruct), _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "dn_gradfeat_bwd_f32": (C.c_int, [_P(MeshBatchStruct)] + [_vp] * 8 + [C.c_int] + [_vp] * 5 + [C.c_size_t, _vp]),
    "dn_linear_workspace_bytes": (C.c_size_t, [_P(MeshBatchStruct), C.c_int, C.c_int]),
    "dn_linear_fwd_f32": (C.c_int, [_P(MeshBatchStruct), _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "dn_linear_bwd_f32": (C.c_int, [_P(MeshBatchStruct), _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "dn_linear_fwd_amax_f32": (C.c_int, [_P(MeshBatchStruct), _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "dn_linear_bwd_amax_f32": (C.c_int, [_P(MeshBatchStruct), _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "dn_block_fwd_workspace_bytes": (C.c_size_t, [_P(MeshBatchStruct), _P(BlockParamsStruct), C.c_int]),
    "dn_block_tracks_amax": (C.c_int, [_P(MeshBatchStruct), _P(BlockParamsStruct), C.c_int]),
    "dn_block_bwd_workspace_bytes": (C.c_size_t, [_P(MeshBatchStruct), _P(BlockParamsStruct)]),
    "dn_block_fwd_f32": (C.c_int, [_P(MeshBatchStruct), _P(BlockParamsStruct), _vp, _vp, _P(BlockSavedStruct), _vp, C.c_size_t, _vp]),
    "dn_block_bwd_f32": (C.c_int, [_P(MeshBatchStruct), _P(BlockParamsStruct), _vp, _P(BlockSavedStruct), _vp,
                                   _P(BlockGradsStruct), _vp, C.c_size_t, _vp]),
    "dn_head_workspace_bytes": (C.c_size_t, []),
    "dn_head_fwd_f32": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_float, C.c_int, _vp, C.c_float, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "dn_head_bwd_f32": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_float, C.c_int, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "dn_hks_f32": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "dn_coo_to_csr_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "dn_coo_to_csr_i64": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_int] + [_vp] * 7 + [_vp, C.c_size_t, _vp]),
    "dn_checksum128": (C.c_int, [_vp, C.c_size_t, C.c_uint64, _vp, _vp]),
    "dn_checksum128_multi": (C.c_int, [C.c_int, _P(_vp), _P(C.c_size_t), _P(C.c_uint64), _vp, _vp]),
    "dn_csr_mean_f32": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_float, _vp, _vp]),
    "dn_mass_mean_fwd_f32": (C.c_int, [_P(MeshBatchStruct), _vp, C.c_int, _vp, _vp, _vp]),
    "dn_mass_mean_bwd_f32": (C.c_int, [_P(MeshBatchStruct), _vp, _vp, C.c_int, _vp, _vp]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None
_allow_host_tensors = False   # flipped ONLY by the CPU-emulator test fixture (tests/emu)


default_options = {}   # tuning options applied to every library this module binds (the test suite: {"chain_min_rows": 0})


def _bind(path):
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    for k, v in default_options.items():
        if lib.dn_set_option(k.encode(), int(v)) != 0:
            raise RuntimeError("unknown library option %r" % k)
    return lib


def lib():
    """The loaded HIP library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libdiffnet_hip.so not found at {LIB_PATH}: build it with "
                "`make -C diffusion-net_amd/csrc` (or __graft_entry__.build()). "
                "There is no fallback path for the DiffusionNet hot ops.")
        _lib = _bind(LIB_PATH)
    return _lib


def set_option(name: str, value: int) -> int:
    """Set a tuning option of the library (include/diffnet_hip.h: dn_set_option); returns the previous value."""
    L = lib()
    old = C.c_int(0)
    check(L.dn_get_option(name.encode(), C.byref(old)), "dn_get_option(%s)" % name)
    check(L.dn_set_option(name.encode(), int(value)), "dn_set_option(%s)" % name)
    return old.value


def get_option(name: str) -> int:
    v = C.c_int(0)
    check(lib().dn_get_option(name.encode(), C.byref(v)), "dn_get_option(%s)" % name)
    return v.value


DIFFUSION_MAX_GROUPS = 4    # include/diffnet_hip.h: DN_DIFFUSION_MAX_GROUPS


def _use_library_for_tests(path, allow_host_tensors):
    """Test hook (tests/emu only): bind another build of the same C ABI."""
    global _lib, _allow_host_tensors
    _lib = _bind(path) if path else None
    _allow_host_tensors = bool(allow_host_tensors)


def require_device(t: torch.Tensor):
    if not t.is_cuda and not _allow_host_tensors:
        raise RuntimeError(
            "diffusion_net HIP ops need tensors on a ROCm device (got %s); there is no CPU path" % t.device)


def ptr(t):
    return None if t is None else t.data_ptr()


def stream_of(t: torch.Tensor):
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return None


def check(err, what):
    if err != 0:
        raise RuntimeError(f"{what} failed with hipError {err}")


_workspaces = {}


def workspace(device, nbytes):
    """Grow-only scratch buffer per (device, current stream): reuse is stream-ordered, and concurrent streams
    (two half-batches in flight) never share scratch."""
    dev = torch.device(device)
    if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
        # A graph owns its scratch: the buffer comes from the capture's memory pool and goes back to it when the caller drops it (later
        # allocations of the same capture may reuse it -- stream order makes that safe, as for every temporary under capture).  A cached
        # buffer must not be handed to a capture: it may live in the pool of graphs that are gone (unmapped with them: seen as a GPU
        # memory fault in a replay), and growing it mid-capture would free memory the nodes already captured still use.
        return torch.empty(int(nbytes) + 4096, dtype=torch.uint8, device=dev)
    sid = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
    key = (str(dev), sid)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=dev)
        _workspaces[key] = buf
    return buf
