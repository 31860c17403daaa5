#!/usr/bin/env python
"""Drop-in flow of the reference scripts (human_segmentation_original.py:105-148): one mesh per step through
``model(x, mass, L=, evals=, evecs=, gradX=, gradY=, faces=)``, operators moved to the device every step,
log_softmax + nll_loss, torch.optim.Adam.  Reports ms/step and where the host time goes."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-net_amd"))
import torch
import torch.nn.functional as F

import diffusion_net
from diffusion_net import synthetic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--verts", type=int, default=7000)
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    meshes = [synthetic.make_mesh_operators(a.verts + 13 * i, 128, seed=i) for i in range(4)]
    labels = [torch.randint(0, 8, (m["faces"].shape[0],)) for m in meshes]
    torch.manual_seed(0)
    model = diffusion_net.layers.DiffusionNet(3, 8, C_width=128, N_block=4, outputs_at="faces", dropout=True,
                                              last_activation=lambda x: F.log_softmax(x, dim=-1)).to(dev)
    model.load_state_dict(synthetic.randomize_times(model.state_dict(), seed=0))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    model.train()

    def step(i, timers=None):
        m, lab = meshes[i % 4], labels[i % 4]
        t0 = time.perf_counter()
        verts, faces, mass, evals, evecs = (m[k].to(dev) for k in ("verts", "faces", "mass", "evals", "evecs"))
        gX, gY, lab_d = m["gradX"].to(dev), m["gradY"].to(dev), lab.to(dev)
        t1 = time.perf_counter()
        opt.zero_grad()
        preds = model(verts, mass, L=None, evals=evals, evecs=evecs, gradX=gX, gradY=gY, faces=faces)
        t2 = time.perf_counter()
        loss = F.nll_loss(preds, lab_d)
        loss.backward()
        opt.step()
        t3 = time.perf_counter()
        if timers is not None:
            timers[0] += t1 - t0; timers[1] += t2 - t1; timers[2] += t3 - t2
        return loss

    for i in range(16):         # every mesh seen four times: the automatic graph capture (diffusion_net.autograph) has happened
        step(i)
    torch.cuda.synchronize()
    from diffusion_net import autograph
    print("autograph:", "on" if autograph.enabled else "off", autograph.stats)
    timers = [0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(i, timers)
    acc = loss.item()          # the scripts sync once per step; once at the end here
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"drop-in single-mesh train step, V={a.verts}: {dt*1e3:.2f} ms/step = {a.verts/dt/1e6:.2f} M verts/s "
          f"(host enqueue: H2D {timers[0]/a.steps*1e3:.2f} ms, forward {timers[1]/a.steps*1e3:.2f} ms, "
          f"loss+backward+Adam {timers[2]/a.steps*1e3:.2f} ms)")


if __name__ == "__main__":
    main()
