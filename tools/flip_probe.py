"""Development aid: saves the post-ReLU activations of one block of the headline parity case (run once per engine), then compares."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "diffusion-net_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
if sys.argv[1] == "run":
    import diffusion_net, parity_cases
    from diffusion_net import synthetic
    from diffusion_net.batch import GatherPattern
    dev = torch.device("cuda:0")
    seed, sizes, K, C = 1, (10500,), 128, 128
    torch.manual_seed(seed)
    model = diffusion_net.layers.DiffusionNet(3, 8, C_width=C, N_block=4, outputs_at="faces", dropout=False)
    model.load_state_dict(synthetic.randomize_times(model.state_dict(), seed=seed))
    model.to(dev).train(False)
    meshes, feats = parity_cases.make_ragged(sizes, K, 3, seed)
    mb = parity_cases.pack(meshes, dev)
    x = torch.cat(feats, 0).to(dev).requires_grad_(True)
    gather = GatherPattern(meshes[0]["faces"].to(dev), sum(sizes))
    keep = {}
    for bi, blk in enumerate(model.blocks):
        blk.register_forward_hook(lambda mod, inp, out, bi=bi: keep.__setitem__(bi, out))
    orig = diffusion_net.layers.DiffusionNetBlock.forward_packed
    def fp(self, x2d, mb_):
        out = orig(self, x2d, mb_)
        keep[id(self)] = out
        return out
    diffusion_net.layers.DiffusionNetBlock.forward_packed = fp
    out = model.forward_packed(x, mb, gather)
    res = {}
    for bi, blk in enumerate(model.blocks):
        sav = keep[id(blk)].grad_fn.saved_tensors
        res[bi] = {"h0": sav[10].cpu(), "h1": sav[11].cpu(), "g": sav[7].cpu(), "xd": sav[3].cpu(), "x": sav[0].detach().cpu()}
    torch.save(res, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for bi in a:
        for k in ("x", "xd", "g", "h0", "h1"):
            ta, tb = a[bi][k], b[bi][k]
            flips = ((ta > 0) != (tb > 0)) if k in ("h0", "h1") else torch.zeros_like(ta, dtype=torch.bool)
            nf = int(flips.sum())
            msg = "block %d %-3s rel-max diff %.2e  scale %.2e" % (bi, k, float((ta - tb).abs().max() / tb.abs().max()), float(tb.abs().max()))
            if k in ("h0", "h1"):
                msg += "  sign flips %d" % nf
                if nf:
                    idx = flips.nonzero()[:5]
                    msg += " at " + ", ".join("(%d,%d): %.3e vs %.3e" % (i, j, float(ta[i, j]), float(tb[i, j])) for i, j in idx.tolist())
            print(msg)
