#!/usr/bin/env python
"""bench.py -- DiffusionNet training-step throughput on MI355X (BASELINE.json metric).

One "step" = forward + NLL loss + backward (+ RCCL gradient all-reduce for N>1) + Adam update over
one ragged batch of synthetic meshes per GPU (weak scaling: every rank owns its own batch).
Workload (config.workload): BASELINE configs[1] shape (human-segmentation net: C_in=3, C_out=8,
C_width=128, K=128, N_block=4, per-face outputs, dropout on) at the north-star size
(>=10k-vertex meshes, batch of meshes).

Prints ONE JSON line on rank 0 (see the contract in the task statement) with two extra objects:
  roofline     -- dominant kernel family, algorithmic flops (or bytes) / hipEvent-measured duration
  cpu_baseline -- the CPU oracle (torch-CPU restatement of the reference) on a bounded sample
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "diffusion-net_amd"))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist
import torch.nn.functional as F

PEAK_MFMA_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: exact-f32 MFMA = f32 vector peak
PEAK_HBM_GBPS = 8000.0            # HBM3E spec (6.29 TB/s measured copy ceiling)
PEAK_MFMA_BF16_TFLOPS = 2500.0    # dense bf16 MFMA (no sparsity)


def mesh_sizes(n_meshes, v_mean, rank):
    # ragged but deterministic: +-10 % around v_mean
    g = torch.Generator().manual_seed(1234 + rank)
    return [int(v_mean * (0.9 + 0.2 * torch.rand(1, generator=g).item())) for _ in range(n_meshes)]


def build_batch(sizes, K, device, seed0):
    from diffusion_net import synthetic
    from diffusion_net.batch import GatherPattern, MeshBatch
    meshes = [synthetic.make_mesh_operators(v, K, seed=seed0 + i) for i, v in enumerate(sizes)]
    mb = MeshBatch.from_operators([m["mass"] for m in meshes], [m["evals"] for m in meshes],
                                  [m["evecs"] for m in meshes], [m["gradX"] for m in meshes],
                                  [m["gradY"] for m in meshes], device=device)
    offs, faces = 0, []
    for m, v in zip(meshes, sizes):
        faces.append(m["faces"] + offs)
        offs += v
    faces = torch.cat(faces, 0).to(device)
    gather = GatherPattern(faces, sum(sizes))
    x = torch.cat([m["verts"] for m in meshes], 0).to(device)
    return meshes, mb, gather, x


def cpu_baseline(args, C_out):
    """CPU oracle (kind 'port'): fwd + loss + bwd + Adam on ONE mesh of the workload per step."""
    import diffusion_net
    from diffusion_net import synthetic
    from oracle import diffusionnet_oracle as orc
    torch.manual_seed(0)
    V = args.verts
    m = synthetic.make_mesh_operators(V, args.keig, seed=99)
    model = diffusion_net.layers.DiffusionNet(3, C_out, C_width=args.cwidth, N_block=args.blocks, outputs_at="faces")
    params = {k: v.clone().requires_grad_(True) for k, v in synthetic.randomize_times(model.state_dict(), seed=0).items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-3)
    labels = torch.randint(0, C_out, (m["faces"].shape[0],))
    n_mask = 2

    def step():
        opt.zero_grad()
        masks = [[torch.bernoulli(torch.full((V, args.cwidth), 0.5)) for _ in range(n_mask)] for _ in range(args.blocks)]
        p = dict(params)
        for k in p:
            if k.endswith("diffusion_time"):
                p[k] = orc.clamp_time(p[k])
        out = orc.net_forward(p, m["verts"], m["mass"], m["evals"], m["evecs"], m["gradX"], m["gradY"], faces=m["faces"],
                              outputs_at="faces", last_activation=lambda t: F.log_softmax(t, dim=-1), keep_masks=masks)
        F.nll_loss(out, labels).backward()
        opt.step()

    for _ in range(2):
        step()
    t0, n = time.perf_counter(), 0
    while n < 40 and (time.perf_counter() - t0 < 15.0 or n < 3):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": V / dt, "unit": "vertices/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} train steps of one {V}-vertex mesh (same net/config, torch-CPU oracle, fp32)",
            "ms_per_step": dt * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--meshes", type=int, default=16, help="meshes per GPU per step")
    ap.add_argument("--verts", type=int, default=10000, help="mean vertices per mesh")
    ap.add_argument("--cwidth", type=int, default=128)
    ap.add_argument("--keig", type=int, default=128)
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--streams", type=int, default=1, help="split the per-GPU batch into this many sub-batches run on separate HIP streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU of this node over RCCL, same arguments
        import socket
        import subprocess
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but the launcher started %d ranks" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (no CPU path)"
    assert torch.cuda.device_count() >= (local + 1), "rank %d has no GPU (visible devices: %d)" % (rank, torch.cuda.device_count())
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
        assert dist.get_backend() == "nccl" and dist.get_world_size() == args.gpus

    import diffusion_net
    from diffusion_net import _hip, synthetic
    from diffusion_net.dist import FlatParams
    if not os.path.exists(_hip.LIB_PATH):      # fresh checkout: compile the HIP sources once (rank 0), never a fallback
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if world > 1:
            dist.barrier()
    lib = _hip.lib()

    C_in, C_out = 3, 8
    torch.manual_seed(0)                       # identical replicas on every rank
    model = diffusion_net.layers.DiffusionNet(C_in, C_out, C_width=args.cwidth, N_block=args.blocks,
                                              outputs_at="faces", dropout=True)
    model.load_state_dict(synthetic.randomize_times(model.state_dict(), seed=0))
    model.to(device).train()
    nsub = max(1, min(args.streams, args.meshes))
    flat = FlatParams(model, direct_sinks=(nsub == 1))   # concurrent backward streams need autograd's ordered accumulation
    try:                                       # one fused update kernel over the flat parameter buffer
        opt = torch.optim.Adam([flat.master], lr=1e-3, fused=True)
    except (TypeError, RuntimeError):
        opt = torch.optim.Adam([flat.master], lr=1e-3)

    sizes = mesh_sizes(args.meshes, args.verts, rank)
    subs = []
    for j in range(nsub):
        sub_sizes = sizes[j::nsub]
        _, mb_j, gather_j, x_j = build_batch(sub_sizes, args.keig, device, seed0=1000 * rank + 100 * j)
        labels_j = torch.randint(0, C_out, (gather_j.n_out,), device=device)
        subs.append((mb_j, gather_j, x_j, labels_j, sum(sub_sizes)))
    mb = subs[0][0]
    v_step = sum(sizes)
    streams = [torch.cuda.Stream(device) for _ in range(nsub)] if nsub > 1 else [None]

    def step():
        flat.zero_grad()
        if nsub == 1:
            mb_j, gather_j, x_j, labels_j, _ = subs[0]
            out = model.forward_packed(x_j, mb_j, gather_j)
            loss = diffusion_net.utils.nll_loss(F.log_softmax(out, dim=-1), labels_j)
            loss.backward()
        else:
            # sub-batches on separate streams: the store phase of one overlaps the MFMA phase of the other;
            # gradients of all sub-batches accumulate into the same flat bucket (mean over the whole batch)
            cur = torch.cuda.current_stream(device)
            losses = []
            for st, (mb_j, gather_j, x_j, labels_j, v_j) in zip(streams, subs):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    out = model.forward_packed(x_j, mb_j, gather_j)
                    losses.append(diffusion_net.utils.nll_loss(F.log_softmax(out, dim=-1), labels_j) * (1.0 / nsub))
            for st, l in zip(streams, losses):
                with torch.cuda.stream(st):
                    l.backward()
            for st in streams:
                cur.wait_stream(st)
            loss = losses[0]
        flat.all_reduce_mean()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    lib.dn_prof_reset()
    lib.dn_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    lib.dn_prof_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        vt = torch.tensor([v_step], device=device, dtype=torch.float64)
        dist.all_reduce(vt, op=dist.ReduceOp.SUM)
        v_all = float(vt.item())
    else:
        v_all = float(v_step)
    assert torch.isfinite(loss).item()

    # ---- per-kernel-family timing from the library's hipEvent brackets (this rank, timed region)
    fam = []
    buf = (ctypes.c_double * 4)()
    for k in range(5):
        lib.dn_prof_read(k, buf)
        ms, n, fl, by = buf[0], buf[1], buf[2], buf[3]
        if n > 0:
            fam.append({"kernel": lib.dn_prof_kind_name(k).decode(), "ms_total": ms, "launches": int(n),
                        "avg_us": 1e3 * ms / n, "tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                        "gbps": by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                        "flops_per_launch": fl / n, "bytes_per_launch": by / n})
    dom = max(fam, key=lambda f: f["ms_total"])
    # Binding roof of a family = the lower of the two roofs at its arithmetic intensity.  The GEMM families run on
    # split-bf16 MFMA (6 bf16 MFMAs per fp32 product: effective fp32 peak = 2.5 PF / 6); the sparse family is HBM-bound.
    def bind(f):
        eff_peak = PEAK_MFMA_BF16_TFLOPS / 6.0 if "gemm" in f["kernel"] else PEAK_MFMA_F32_TFLOPS
        ai = f["flops_per_launch"] / max(f["bytes_per_launch"], 1.0)
        ridge = eff_peak * 1e12 / (PEAK_HBM_GBPS * 1e9)
        if f["flops_per_launch"] > 0 and ai > ridge:
            return {"bound": "mfma", "achieved": f["tflops"], "peak": eff_peak, "unit": "TFLOP/s", "frac": f["tflops"] / eff_peak}
        return {"bound": "hbm", "achieved": f["gbps"], "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": f["gbps"] / PEAK_HBM_GBPS}
    roof = bind(dom)
    roof["traffic"] = None
    roof["arithmetic_intensity_flop_per_byte"] = dom["flops_per_launch"] / max(dom["bytes_per_launch"], 1.0)
    roof["effective_fp32_tflops"] = dom["tflops"]
    # HBM bytes per launch of that family from the committed FETCH_SIZE / WRITE_SIZE PMC passes (separate rocprofv3
    # --pmc runs of the same workload, gfx950 x2 read correction applied; tools/traffic_summary.py)
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        roof["traffic"] = tr[dom["kernel"]]["hbm_bytes_per_launch"]
        roof["traffic_note"] = "PMC (2*FETCH_SIZE+WRITE_SIZE)*1024 B per launch; algorithmic bytes per launch = %.4g" % dom["bytes_per_launch"]
    except Exception:
        pass
    roof.update({"kernel": dom["kernel"], "avg_launch_us": dom["avg_us"], "launches": dom["launches"],
                 "share_of_kernel_time": dom["ms_total"] / sum(f["ms_total"] for f in fam)})

    # ---- diffusion block (to_basis + exp(-lambda t) + from_basis) on the same batch: HBM GB/s of BASELINE.json
    from diffusion_net import ops
    Cw, K = args.cwidth, args.keig
    v_sub0 = subs[0][4]
    sizes = sizes[0::nsub]
    xb = torch.randn(v_sub0, Cw, device=device)
    tt = torch.full((Cw,), 0.05, device=device)
    with torch.no_grad():
        for _ in range(3):
            ops.DiffusionFn.apply(xb, tt, mb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 20
        for _ in range(reps):
            ops.DiffusionFn.apply(xb, tt, mb)
        e1.record()
        torch.cuda.synchronize()
    t_diff = e0.elapsed_time(e1) * 1e-3 / reps
    bytes_diff = sum(4.0 * (v * (2 * Cw + 2 * K + 1) + 2 * K * Cw + K + Cw) for v in sizes)
    flops_diff = sum(4.0 * v * K * Cw for v in sizes)
    diff = {"ms": t_diff * 1e3, "gbps": bytes_diff / t_diff / 1e9, "frac_hbm_8TBs": bytes_diff / t_diff / 1e9 / PEAK_HBM_GBPS,
            "tflops": flops_diff / t_diff / 1e12, "frac_mfma_f32": flops_diff / t_diff / 1e12 / PEAK_MFMA_F32_TFLOPS,
            "frac_mfma_bf16x3": flops_diff / t_diff / 1e12 / (PEAK_MFMA_BF16_TFLOPS / 6.0),
            "note": "arithmetic intensity KC/(2(K+C)) = %.0f flop/B; ridge 19.7 (f32 MFMA) / 52 (split-bf16 MFMA, used) -> HBM-bound" % (K * Cw / (2.0 * (K + Cw)))}

    if rank == 0:
        res = {
            "metric": "vertices/sec fwd+bwd, C_width=%d K=%d" % (Cw, K),
            "value": v_all * args.steps / elapsed, "unit": "vertices/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "arithmetic": "fp32 storage and accumulation; dense products on 3-term split-bf16 MFMA (fp32-level accuracy: rel-L2 1.6e-7 vs 2.0e-7 for the f32 MFMA chain, profiles/r01_exp_bf16x3.txt)",
            "config": {"workload": "train step (fwd+NLL+bwd+Adam%s) on a ragged batch of %d meshes x ~%d vertices per GPU, "
                                   "DiffusionNet C_in=3 C_out=8 C_width=%d K=%d N_block=%d outputs_at=faces dropout=on"
                                   % ("+RCCL all-reduce" if world > 1 else "", args.meshes, args.verts, Cw, K, args.blocks),
                       "meshes_per_gpu": args.meshes, "verts_per_gpu_step": v_step, "parallelism": "dp%d" % world,
                       "streams_per_gpu": nsub},
            "roofline": roof, "kernel_families": fam, "diffusion_block": diff,
        }
        if not args.no_cpu_baseline and world == 1:   # reported at N = 1 only (the other ranks would sit idle behind it)
            res["cpu_baseline"] = cpu_baseline(args, C_out)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
