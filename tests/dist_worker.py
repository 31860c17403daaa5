"""Worker for tests/test_dist_gloo.py: one process per rank, gloo backend on CPU.  The HIP kernels run on the
fiber emulator here (test infrastructure); on the GPU box the same code path runs with backend nccl (= RCCL)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "diffusion-net_amd"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(rank, world, port, emu_so, sizes, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import diffusion_net
    from diffusion_net import _hip, synthetic
    from diffusion_net.dist import FlatParams, shard_by_cost
    import parity_cases
    _hip._use_library_for_tests(emu_so, True)
    torch.manual_seed(0)                                   # identical replicas
    model = diffusion_net.layers.DiffusionNet(3, 4, C_width=32, N_block=2, dropout=False)   # two per-block gradient buckets
    model.load_state_dict(synthetic.randomize_times(model.state_dict(), seed=0))
    flat = FlatParams(model)
    assert len(flat.buckets) == 2
    opt = torch.optim.Adam([flat.master], lr=1e-2)
    mine = shard_by_cost(sizes, world)[rank]
    meshes, feats = parity_cases.make_ragged(sizes, 16, 3, seed=1)
    mb = parity_cases.pack([meshes[i] for i in mine], "cpu")
    x = torch.cat([feats[i] for i in mine], 0)
    first_grad = None
    for _ in range(2):
        flat.zero_grad()
        out = model.forward_packed(x, mb)
        # per-mesh mean loss so that the rank average equals the global mean over meshes (equal mesh counts per rank)
        off, loss = 0, 0.0
        for i in mine:
            loss = loss + out[off:off + sizes[i]].square().mean()
            off += sizes[i]
        (loss / len(mine)).backward()
        flat.all_reduce_mean()
        if first_grad is None:
            first_grad = flat.grad.clone()
        opt.step()
    torch.save({"flat": flat.flat.clone(), "grad": first_grad, "mine": mine}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def run_accumulate(rank, world, port, emu_so, sizes, out_dir):
    """Two micro-batches per optimizer step on every rank (gradient accumulation), three ways: per-block overlap with both backwards
    sending (``_bucket_reopen`` path), overlap with the first backward under ``no_sync()``, and ``overlap=False`` (one flat
    all-reduce).  ADVICE r2: the first used to leave later contributions unreduced and raced the in-flight collective."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import diffusion_net
    from diffusion_net import _hip, synthetic
    from diffusion_net.dist import FlatParams
    import parity_cases
    _hip._use_library_for_tests(emu_so, True)
    meshes, feats = parity_cases.make_ragged(sizes, 16, 3, seed=1)
    mine = [i for i in range(len(sizes)) if i % world == rank]          # two meshes per rank = two micro-batches
    grads = {}
    for mode in ("overlap", "no_sync", "flat"):
        torch.manual_seed(0)
        model = diffusion_net.layers.DiffusionNet(3, 4, C_width=32, N_block=2, dropout=False)
        model.load_state_dict(synthetic.randomize_times(model.state_dict(), seed=0))
        flat = FlatParams(model, overlap=(mode != "flat"))
        assert len(flat.buckets) == (0 if mode == "flat" else 2)
        flat.zero_grad()
        for j, i in enumerate(mine):
            mb = parity_cases.pack([meshes[i]], "cpu")
            last = j == len(mine) - 1
            if mode == "no_sync" and not last:
                with flat.no_sync():
                    model.forward_packed(feats[i], mb).square().mean().backward()
                assert not flat._sent
            else:
                model.forward_packed(feats[i], mb).square().mean().backward()
        if mode != "flat":
            assert sorted(flat._sent) == [0, 1]
        flat.all_reduce_mean()
        grads[mode] = flat.grad.clone()
    torch.save(grads, os.path.join(out_dir, f"acc_rank{rank}.pt"))
    dist.destroy_process_group()


def run_autograph_dropout(rank, world, port, emu_so, out_dir):
    """Data-parallel replicas are seeded identically; their dropout masks must not be (the check for the automatic
    graph replay path).  Every rank runs the reference-signature forward of the SAME model on the SAME mesh in train mode with the
    automatic capture on (closure-rerun backend on the CPU emulator: same seed plumbing as the HIP-graph backend) and keeps the
    outputs of the replayed calls; the test compares them across ranks."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import diffusion_net
    from diffusion_net import _hip, autograph, synthetic
    from diffusion_net.batch import operator_cache
    _hip._use_library_for_tests(emu_so, True)
    autograph.backend, autograph.warm_calls, autograph.enabled = autograph.RerunBackend, 1, True
    operator_cache.clear()
    torch.manual_seed(0)                                   # identical replicas, identical torch RNG streams
    model = diffusion_net.layers.DiffusionNet(3, 4, C_width=32, N_block=2, dropout=True)
    model.load_state_dict(synthetic.randomize_times(model.state_dict(), seed=0))
    model.train()
    m = synthetic.make_mesh_operators(90, 8, seed=3)
    outs = []
    with torch.no_grad():
        for _ in range(5):
            outs.append(model(m["verts"], m["mass"], L=None, evals=m["evals"], evecs=m["evecs"], gradX=m["gradX"], gradY=m["gradY"]).clone())
    st = dict(autograph.stats)
    torch.save({"outs": torch.stack(outs), "replays": st["replays_fwd"], "captures": st["captures"]}, os.path.join(out_dir, f"ag_rank{rank}.pt"))
    dist.destroy_process_group()
