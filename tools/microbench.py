#!/usr/bin/env python
"""Per-op timings of the HIP path at the benchmark shape (development aid, GPU only).
Prints one line per op: avg ms, TFLOP/s (exact-f32 MFMA peak 157.3) and algorithmic GB/s."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-net_amd"))
sys.path.insert(0, ROOT)
import torch

import bench
from diffusion_net import ops


REPS = [20]


def timeit(fn, reps=None, warm=2):
    reps = reps or REPS[0]
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--meshes", type=int, default=16)
    ap.add_argument("--verts", type=int, default=10000)
    ap.add_argument("--cwidth", type=int, default=128)
    ap.add_argument("--keig", type=int, default=128)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--zeros", action="store_true", help="all-zero operands (DVFS probe: same instruction stream, least switching power)")
    a = ap.parse_args()
    REPS[0] = a.reps
    dev = torch.device("cuda:0")
    sizes = bench.mesh_sizes(a.meshes, a.verts, 0)
    meshes, mb, gather, x3 = bench.build_batch(sizes, a.keig, dev, 0)
    V, C, K = sum(sizes), a.cwidth, a.keig
    g = torch.Generator(device="cpu").manual_seed(0)
    R = (lambda *s: torch.zeros(*s, device=dev)) if a.zeros else (lambda *s: torch.randn(*s, generator=g).to(dev))
    x, y = R(V, C), R(V, C)
    W = R(C, C) / C ** 0.5
    W3 = R(C, 3 * C) / C ** 0.5
    b = R(C)
    t = torch.full((C,), 0.05, device=dev)
    spec = R(len(sizes), K, C)
    nnz = int(mb.g_col.shape[0])
    rows = []

    def rec(name, ms, flops, byts):
        rows.append((name, ms, flops / ms / 1e9, byts / ms / 1e6))
        print(f"{name:34s} {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TFLOP/s ({flops/ms/1e9/157.3*100:5.1f}%)  {byts/ms/1e6:8.0f} GB/s", flush=True)

    with torch.no_grad():
        z = torch.empty_like(x)
        rec("[calib] torch fill 81MB", timeit(lambda: z.fill_(1.0)), 0.0, 4.0 * V * C)
        rec("[calib] torch copy 81MB->81MB", timeit(lambda: z.copy_(x)), 0.0, 8.0 * V * C)
        rec("[calib] torch add 2x81MB->81MB", timeit(lambda: torch.add(x, y, out=z)), 0.0, 12.0 * V * C)
        # the three lines above re-touch the same <=243 MB: they sit in the 256 MiB Infinity Cache and are NOT an HBM figure.
        # HBM calibration: copies cycling through 8 source/destination pairs (1.3 GB), so that nothing is re-read from the cache
        ring = [(torch.randn(V, C, device=dev), torch.empty(V, C, device=dev)) for _ in range(8)]
        cnt = [0]

        def rot_copy():
            s_, d_ = ring[cnt[0] % 8]
            cnt[0] += 1
            d_.copy_(s_)
        rec("[calib] HBM copy, 8 rotating 81MB pairs", timeit(rot_copy), 0.0, 8.0 * V * C)
        del ring
        rec("to_basis (tngemm+reduce)", timeit(lambda: ops._to_basis_raw(mb, x, True)), 2.0 * V * K * C, 4.0 * V * (K + C + 1))
        rec("from_basis (rowgemm NN)", timeit(lambda: ops._from_basis_raw(mb, spec)), 2.0 * V * K * C, 4.0 * V * (K + C))
        rec("diffusion fwd (3 launches)", timeit(lambda: ops.DiffusionFn.apply(x, t, mb)), 4.0 * V * K * C, 4.0 * V * (2 * C + 2 * K + 1))
        rec("grad_apply fwd (spmm x2)", timeit(lambda: ops.GradApplyFn.apply(x, mb)), 4.0 * nnz * C, 4.0 * (V + 3 * nnz + 3 * V * C))
        rec("gradfeat fwd (dual rowgemm NT)", timeit(lambda: ops.GradFeatFn.apply(x, y, W, W, mb)), 8.0 * V * C * C, 4.0 * 5 * V * C)
        rec("linear C->C (rowgemm NT)", timeit(lambda: ops.LinearFn.apply(x, W, b, mb)), 2.0 * V * C * C, 4.0 * 2 * V * C)
        x3c = torch.cat([x, y, x], 1).contiguous()
        rec("linear 3C->C (rowgemm NT)", timeit(lambda: ops.LinearFn.apply(x3c, W3, b, mb)), 6.0 * V * C * C, 4.0 * 4 * V * C)
        rec("linear 3->C (first_lin)", timeit(lambda: ops.LinearFn.apply(x3, R(C, 3), b, mb)), 6.0 * V * C, 4.0 * V * (C + 3))
    # block fwd / bwd through the fused entry points
    import diffusion_net
    torch.manual_seed(0)
    blk = diffusion_net.layers.DiffusionNetBlock(C, [C, C], dropout=False).to(dev)
    with torch.no_grad():
        blk.diffusion.diffusion_time.fill_(0.05)
        fl_f = 2.0 * V * C * C * (2 * K / C + 4 + 3 + 1 + 1) + 4.0 * nnz * C
        rec("block fwd (inference)", timeit(lambda: blk.forward_packed(x, mb)), fl_f, 4.0 * V * C * 12)
    xg = x.clone().requires_grad_(True)

    def fb():
        out = blk.forward_packed(xg, mb)
        out.backward(y)
    rec("block fwd+bwd (train)", timeit(fb, reps=max(2, a.reps // 2)), 3.1 * fl_f, 4.0 * V * C * 60)


if __name__ == "__main__":
    main()
