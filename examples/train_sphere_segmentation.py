#!/usr/bin/env python
"""A miniature of the reference's experiments/human_segmentation_original train loop on synthetic data, showing the
drop-in flow end to end on MI355X: real triangle meshes -> diffusion_net.geometry.get_operators (host precompute + npz cache)
-> diffusion_net.layers.DiffusionNet (HIP hot path) -> log_softmax / nll_loss -> Adam.  Per-face labels = octant of the face centre.

    python examples/train_sphere_segmentation.py --epochs 3
"""
import argparse
import os
import sys

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diffusion-net_amd"))   # where the scripts put src/
import torch

import diffusion_net


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--meshes", type=int, default=6)
    ap.add_argument("--verts", type=int, default=2500)
    ap.add_argument("--k_eig", type=int, default=64)
    ap.add_argument("--cache", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "op_cache"))
    args = ap.parse_args()
    device = torch.device("cuda:0")

    data = []
    for i in range(args.meshes):
        verts, faces = diffusion_net.synthetic.sphere_mesh(args.verts + 37 * i, seed=i, bump=0.15)
        verts = diffusion_net.geometry.normalize_positions(torch.from_numpy(verts).float())
        faces = torch.from_numpy(faces)
        ops = diffusion_net.geometry.get_operators(verts, faces, k_eig=args.k_eig, op_cache_dir=args.cache)
        centre = verts[faces].mean(1)
        labels = ((centre[:, 0] > 0).long() * 4 + (centre[:, 1] > 0).long() * 2 + (centre[:, 2] > 0).long())
        data.append((verts, faces, ops, labels))

    model = diffusion_net.layers.DiffusionNet(C_in=3, C_out=8, C_width=64, N_block=4, outputs_at="faces", dropout=True,
                                              last_activation=lambda x: torch.nn.functional.log_softmax(x, dim=-1)).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    for epoch in range(args.epochs):
        model.train()
        correct = total = 0
        for verts, faces, (frames, mass, L, evals, evecs, gradX, gradY), labels in data:
            verts, faces, mass, evals, evecs, gradX, gradY, labels = (t.to(device) for t in (verts, faces, mass, evals, evecs, gradX, gradY, labels))
            opt.zero_grad()
            preds = model(verts, mass, L=None, evals=evals, evecs=evecs, gradX=gradX, gradY=gradY, faces=faces)
            loss = diffusion_net.utils.nll_loss(preds, labels)
            loss.backward()
            opt.step()
            correct += (preds.argmax(-1) == labels).sum().item()
            total += labels.numel()
        print(f"epoch {epoch}: loss {loss.item():.3f}  train acc {correct / total:.3f}")


if __name__ == "__main__":
    main()
