#!/usr/bin/env python
"""Times the thin ends of the network at the bench shape: first_lin (3 -> C) and last_lin (C -> 8) forward/backward, and the fused head
(face gather-mean + log_softmax + NLL) forward/backward.  python tools/bench_thin.py"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "diffusion-net_amd"))
import bench  # noqa: E402
from diffusion_net import ops  # noqa: E402


def timeit(fn, reps=30, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    sizes = bench.mesh_sizes(16, 10000, 0)
    meshes, mb, gather, x3 = bench.build_batch(sizes, 128, dev, 0)
    V, C = sum(sizes), 128
    g = torch.Generator().manual_seed(0)
    R = lambda *s: torch.randn(*s, generator=g).to(dev)
    for name, cin, cout in (("first_lin 3->128", 3, C), ("last_lin 128->8", C, 8)):
        x = R(V, cin).requires_grad_(True)
        W, b = R(cout, cin).requires_grad_(True), R(cout).requires_grad_(True)
        d = R(V, cout)
        y = ops.LinearFn.apply(x, W, b, mb)
        print("%-22s fwd %7.1f us   bwd %7.1f us" % (name, timeit(lambda: ops.LinearFn.apply(x, W, b, mb)),
                                                     timeit(lambda: torch.autograd.grad(y, (x, W, b), d, retain_graph=True))), flush=True)
    logits = R(V, 8).requires_grad_(True)
    labels = torch.randint(0, 8, (gather.n_out,), device=dev)
    _, loss = ops.HeadFn.apply(logits, gather, labels, True, 0.0, True)
    print("%-22s fwd %7.1f us   bwd %7.1f us   (n_out %d)" % ("head faces+lsm+nll", timeit(lambda: ops.HeadFn.apply(logits, gather, labels, True, 0.0, True)),
                                                         timeit(lambda: torch.autograd.grad(loss, logits, retain_graph=True)), gather.n_out))


if __name__ == "__main__":
    main()
