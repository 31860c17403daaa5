#!/usr/bin/env python
"""Phase timeline of the persistent row-GEMM (development aid, GPU only).

Needs the trace build:  make -C diffusion-net_amd/csrc BUILD=build_trace EXTRA=-DDN_PT_TRACE OUT=../diffusion_net/libdiffnet_hip_trace.so
Run as                  DN_LIB_VARIANT=trace python tools/trace_pt.py
Prints, for waves 0 and 7 of workgroup 0, the average shader cycles spent in each phase of a slice iteration."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-net_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from diffusion_net import _hip, ops

PHASES_MFMA = ["reads+MFMA k16 #0", "MFMA k16 #1", "park", "barrier wait"]
PHASES_LOAD = ["wait+split+LDS writes", "pieces out", "piece operands+cursor+prefetch issue", "barrier"]


def dump(name):
    torch.cuda.synchronize()
    lib = _hip.lib()
    buf = (ctypes.c_ulonglong * 8192)()
    lib.dn_debug_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rc = lib.dn_debug_trace_read(buf, 8192)
    assert rc == 0, rc
    t = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(2, 4096)
    print("== %s" % name)
    for w, lab, ph in ((0, "MFMA wave 0", PHASES_MFMA), (1, "loader wave 4", PHASES_LOAD)):
        tt = t[w]
        good = tt > 0
        n = int(np.argmin(good)) if not good.all() else 4096
        n = (n // 4) * 4
        it = tt[:n].reshape(-1, 4)
        d = np.concatenate([np.diff(it, axis=1), (it[1:, 0] - it[:-1, 3])[:, None].tolist() + [[0]]], axis=1)
        whole = np.diff(it[:, 0])
        print(" %s: %d iterations, %.0f cycles/iteration" % (lab, it.shape[0], whole.mean() if len(whole) else 0))
        for k, name in enumerate(ph):
            print("   %-26s mean %7.0f  min %6d  max %6d" % (name, d[:-1, k].mean(), d[:-1, k].min(), d[:-1, k].max()))
        print("   first 12 iterations:")
        for r in d[:12]:
            print("    ", " ".join("%6d" % v for v in r))


def main():
    dev = torch.device("cuda:0")
    sizes = bench.mesh_sizes(16, 10000, 0)
    meshes, mb, gather, x3 = bench.build_batch(sizes, 128, dev, 0)
    V, C, K = sum(sizes), 128, 128
    g = torch.Generator(device="cpu").manual_seed(0)
    R = lambda *s: torch.randn(*s, generator=g).to(dev)
    x = R(V, C)
    W = R(C, C) / C ** 0.5
    b = R(C)
    spec = R(len(sizes), K, C)
    with torch.no_grad():
        for _ in range(3):
            ops._from_basis_raw(mb, spec)
        dump("from_basis (MODE 0, B = per-mesh spectrum)")
        for _ in range(3):
            ops.LinearFn.apply(x, W, b, mb)
        dump("linear C->C (MODE 0, bias)")


if __name__ == "__main__":
    main()
