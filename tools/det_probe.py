import sys, torch, collections
sys.path.insert(0, "diffusion-net_amd"); sys.path.insert(0, "."); 
import bench, diffusion_net
dev = torch.device("cuda:0")
sizes = bench.mesh_sizes(8, 6000, 0)
meshes, mb, gather, x3 = bench.build_batch(sizes, 128, dev, 0)
V, C = sum(sizes), 128
g = torch.Generator().manual_seed(0)
x = torch.randn(V, C, generator=g).to(dev); w = torch.randn(V, C, generator=g).to(dev)
torch.manual_seed(0)
blk = diffusion_net.layers.DiffusionNetBlock(C, [C, C], dropout=True).to(dev).train(True)
blk.drop_seed_provider = lambda: 12345
seen = collections.Counter(); outs = {}
import ctypes, numpy as np
from diffusion_net import _hip
dbg = getattr(ctypes.CDLL(_hip.LIB_PATH), "dn_debug_scales_read", None) if "dbg" in _hip.LIB_PATH else None
scale_sets = collections.Counter()
for it in range(10000):
    xi = x.clone().requires_grad_(True)
    out = blk.forward_packed(xi, mb)
    sav = out.grad_fn.saved_tensors
    names = ["x", "time", "xs", "xd", "words", "gx", "gy", "g", "bre", "bim", "h0", "h1"]
    sig = tuple((n, float(t.double().sum()) if n != "words" else tuple(v.hex() for v in t.cpu().tolist()[:6])) for n, t in zip(names, sav[:12]))
    seen[sig] += 1
    if dbg is not None:
        torch.cuda.synchronize()
        buf = np.zeros(2 * mb.tiles.shape[0], dtype=np.float32)
        dbg(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
        vals = collections.Counter(zip(buf[0::2].tolist(), buf[1::2].tolist()))
        scale_sets[tuple(sorted(vals.items()))] += 1
base = max(seen, key=seen.get)
for k, n in seen.items():
    print(n, "x", "MAJORITY" if k is base else "differs in: " + ", ".join("%s (%s vs %s)" % (a[0], a[1], b[1]) for a, b in zip(k, base) if a != b))
for k, n in scale_sets.items():
    print(n, "reps with (sa, sb) -> #workgroups:", k)
print("mb.amax", mb.amax.tolist())
