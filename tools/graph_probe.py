#!/usr/bin/env python
"""Which forward-only HIP-graph capture of the packed forward survives?  Each case runs in its own process (a crash in one cannot
mask the others): python tools/graph_probe.py            -> runs every case and prints one line per case
                  python tools/graph_probe.py <case>     -> one case in this process"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-net_amd"))

CASES = ["c32_private", "c32_shared", "c32_nograd", "c32_dropout", "c128_private", "c128_nodrop_shared", "c32_vertices", "c32_first_lin", "c32_block",
         "c32_last_lin", "c32_head", "c32_fwd_bwd_one_graph", "c32_v3000", "c64_private", "ag_live_graph"]


def one(case):
    import torch
    import torch.nn.functional as F
    import diffusion_net
    from diffusion_net import ops, synthetic
    from diffusion_net.batch import GatherPattern, MeshBatch
    dev = torch.device("cuda:0")
    C = 128 if case.startswith("c128") else (64 if case.startswith("c64") else 32)
    V = 3000 if case.endswith("v3000") else 300
    K = 16
    drop = case in ("c32_dropout", "c128_private")
    torch.manual_seed(0)
    outputs_at = "vertices" if case == "c32_vertices" else "faces"
    model = diffusion_net.layers.DiffusionNet(3, 4, C_width=C, N_block=2, outputs_at=outputs_at, dropout=drop,
                                              last_activation=lambda t: F.log_softmax(t, dim=-1)).to(dev).train()
    m = synthetic.make_mesh_operators(V, K, seed=1)
    mb = MeshBatch.from_operators([m["mass"]], [m["evals"]], [m["evecs"]], [m["gradX"]], [m["gradY"]], device=dev)
    gather = GatherPattern(m["faces"].to(dev), V) if outputs_at == "faces" else None
    x = m["verts"].to(dev)
    grad = case != "c32_nograd"
    if drop:
        seed = torch.zeros(1, dtype=torch.int64, device=dev)
        for bi, blk in enumerate(model.blocks):
            blk._graph_seed = (12345 + 2 * bi + 1, seed)
    xc = torch.randn(V, C, device=dev)

    def fn():
        with torch.set_grad_enabled(grad):
            if case == "c32_first_lin":
                return model.first_lin.apply_rows(x, mb)
            if case == "c32_block":
                return model.blocks[0].forward_packed(xc, mb)
            if case == "c32_last_lin":
                return model.last_lin.apply_rows(xc, mb)
            if case == "c32_head":
                return ops.HeadFn.apply(torch.randn(V, 4, device=dev, requires_grad=True), gather, None, True, 0.0, True)[0]
            out = model.forward_packed(x, mb, gather)
            if case == "c32_fwd_bwd_one_graph":
                torch.autograd.grad(out, [p for p in model.parameters()], torch.ones_like(out), allow_unused=True)
            return out

    if case == "ag_live_graph":
        # the caller's previous loss (built on the default stream) is still alive when diffusion_net.autograph captures: the parameters'
        # gradient-accumulator nodes belong to the default stream (a captured backward into them crashed hipStreamEndCapture)
        from diffusion_net import autograph
        keep = model.forward_packed(x, mb, gather).sum()
        keep.backward()
        gf = autograph.GraphedForward(model, x, mb, gather, True)
        out = gf(x)
        out.sum().backward()
        torch.cuda.synchronize()
        print(case, "OK", float(out.detach().abs().sum()), float(keep), flush=True)
        return
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    kw = {"pool": torch.cuda.graph_pool_handle()} if "shared" in case else {}
    print(case, "capturing", flush=True)
    with torch.cuda.graph(g, capture_error_mode="thread_local", **kw):
        out = fn()
    print(case, "captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print(case, "OK", float(out.detach().abs().sum()), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=300)
            last = [l for l in r.stdout.strip().splitlines() if l.startswith(c)]
            print("%-24s rc=%-4d %s" % (c, r.returncode, last[-1] if last else r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else ""), flush=True)
