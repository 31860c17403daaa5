#!/usr/bin/env python
"""Bitwise run-to-run determinism stress of the HIP path (development aid, GPU only).

Every op of the block is repeated --reps times on identical inputs at the benchmark shape; any tensor that is not bitwise
equal to the first repetition is reported with the rows / columns that differ.  (This is how the intermittent stale-register
stores behind packed converts were found: one wrong bf16 pair in ~3 % of launches.)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-net_amd")); sys.path.insert(0, ROOT)
import torch

import bench
import diffusion_net
from diffusion_net import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--meshes", type=int, default=16)
    ap.add_argument("--verts", type=int, default=10000)
    ap.add_argument("--cwidth", type=int, default=128)
    ap.add_argument("--keig", type=int, default=128)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sizes = bench.mesh_sizes(a.meshes, a.verts, 0)
    meshes, mb, gather, x3 = bench.build_batch(sizes, a.keig, dev, 0)
    V, C, K = sum(sizes), a.cwidth, a.keig
    g = torch.Generator().manual_seed(0)
    R = lambda *s: torch.randn(*s, generator=g).to(dev)
    x, y, w = R(V, C), R(V, C), R(V, C)
    W, W3, b = R(C, C) / C ** 0.5, R(C, 3 * C) / C ** 0.5, R(C)
    t = torch.full((C,), 0.05, device=dev)
    torch.manual_seed(0)
    blk = diffusion_net.layers.DiffusionNetBlock(C, [C, C], dropout=True).to(dev).train(True)
    blk.drop_seed_provider = lambda: 12345

    def leafs(*ts):
        return [v.clone().requires_grad_(True) for v in ts]

    def case_gradfeat():
        gx, gy, a_, b_ = leafs(x, y, W, W.t().contiguous())
        out = ops.GradFeatFn.apply(gx, gy, a_, b_, mb)
        (out * w).sum().backward()
        return [out.detach(), gx.grad, gy.grad, a_.grad, b_.grad]

    def case_linear(Wm, xin):
        xi, wi, bi = leafs(xin, Wm, b)
        out = ops.LinearFn.apply(xi, wi, bi, mb)
        (out * w).sum().backward()
        return [out.detach(), xi.grad, wi.grad, bi.grad]

    def case_diffusion():
        xi, ti = leafs(x, t)
        out = ops.DiffusionFn.apply(xi, ti, mb)
        (out * w).sum().backward()
        return [out.detach(), xi.grad, ti.grad]

    def case_gradapply():
        xi, = leafs(x)
        gx, gy = ops.GradApplyFn.apply(xi, mb)
        ((gx + 2 * gy) * w).sum().backward()
        return [gx.detach(), gy.detach(), xi.grad]

    def case_block():
        xi, = leafs(x)
        blk.zero_grad(set_to_none=True)
        out = blk.forward_packed(xi, mb)
        (out * w).sum().backward()
        return [out.detach(), xi.grad] + [p.grad for p in blk.parameters()]

    x3c = torch.cat([x, y, x], 1).contiguous()
    cases = [("gradient features", case_gradfeat), ("linear C->C", lambda: case_linear(W, x)),
             ("linear 3C->C", lambda: case_linear(W3, x3c)), ("diffusion", case_diffusion), ("gradient apply", case_gradapply),
             ("block fwd+bwd (seeded dropout)", case_block)]
    total_bad = 0
    for name, fn in cases:
        first, bad = None, 0
        for it in range(a.reps):
            cur = [v.clone() for v in fn()]
            if first is None:
                first = cur
                continue
            for i, (c0, c1) in enumerate(zip(first, cur)):
                if not torch.equal(c0, c1):
                    bad += 1
                    d = (c1 - c0).abs().reshape(c0.shape[0], -1) if c0.dim() > 1 else (c1 - c0).abs().reshape(1, -1)
                    nz = (d > 0).nonzero()
                    print("  %s: rep %d tensor %d: %d elements differ, max %.3g, rows %d..%d cols %s" % (
                        name, it, i, nz.shape[0], float(d.max()), int(nz[:, 0].min()), int(nz[:, 0].max()), nz[:, 1].unique().tolist()[:6]), flush=True)
        print("%-34s %d reps: %s" % (name, a.reps, "bitwise identical" if bad == 0 else "%d MISMATCHES" % bad), flush=True)
        total_bad += bad
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()
