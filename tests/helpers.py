"""Shared test helpers: golden-fixture loader and error metrics."""
import ast
import glob
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    """Whole-net golden cases (the geom_* fixtures of the host precompute are separate)."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                  if not os.path.basename(p).startswith(("geom_", "feat_", "refcache_", "utils_")))


def load_golden(name, dtype=torch.float32, device="cpu"):
    """Returns (meta, params, inputs, masks, expect) with torch tensors.

    inputs: x_in, mass, evals, evecs, gradX, gradY (torch sparse COO, [V,V] or list for batches),
            edges, faces -- shaped exactly as the reference forward received them."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = ast.literal_eval(str(z["meta"]))
    fl = lambda a: torch.from_numpy(np.asarray(a)).to(dtype).to(device)
    params = {k[len("param."):]: fl(z[k]) for k in z.files if k.startswith("param.")}
    grads = {k[len("grad."):]: fl(z[k]) for k in z.files if k.startswith("grad.")}
    B = meta["B"]
    meshes = []
    for b in range(B or 1):
        pre = f"mesh{b}."
        V = z[pre + "mass"].shape[0]
        idx = torch.from_numpy(z[pre + "grad_idx"].astype(np.int64)).to(device)
        meshes.append(dict(
            mass=fl(z[pre + "mass"]), evals=fl(z[pre + "evals"]), evecs=fl(z[pre + "evecs"]),
            gradX=torch.sparse_coo_tensor(idx, fl(z[pre + "gradX_val"]), (V, V)).coalesce(),
            gradY=torch.sparse_coo_tensor(idx, fl(z[pre + "gradY_val"]), (V, V)).coalesce(),
            faces=torch.from_numpy(z[pre + "faces"].astype(np.int64)).to(device),
            edges=torch.from_numpy(z[pre + "edges"].astype(np.int64)).to(device)))
    if B is None:
        m = meshes[0]
        inputs = dict(x_in=fl(z["x_in"]), **m)
    else:
        st = lambda key: torch.stack([m[key] for m in meshes], 0)
        inputs = dict(x_in=fl(z["x_in"]), mass=st("mass"), evals=st("evals"), evecs=st("evecs"),
                      gradX=[m["gradX"] for m in meshes], gradY=[m["gradY"] for m in meshes],
                      faces=st("faces"), edges=st("edges"))
    masks = [fl(z[k]) for k in sorted((k for k in z.files if k.startswith("mask")), key=lambda s: int(s[4:]))]
    expect = dict(out=fl(z["out"]), loss_w=fl(z["loss_w"]), grads=grads)
    if "ref64.out" in z.files:      # the reference module evaluated in double precision on the same inputs (checkpoint fixtures)
        f64 = lambda a: torch.from_numpy(np.asarray(a)).to(torch.float64)
        expect["out64"] = f64(z["ref64.out"])
        expect["grads64"] = {k[len("ref64.grad."):]: f64(z[k]) for k in z.files if k.startswith("ref64.grad.")}
    return meta, params, inputs, masks, expect


def activation_of(meta):
    if meta["act"] == "log_softmax":
        return lambda t: torch.nn.functional.log_softmax(t, dim=-1)
    return None


def group_masks(meta, params, masks):
    """Flat recorded dropout masks -> per-block lists (one mask per MLP layer i>0)."""
    if not masks:
        return None
    n_block = meta["ctor"].get("N_block", 4)
    per = len(masks) // n_block
    return [masks[i * per:(i + 1) * per] for i in range(n_block)]


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rel_max(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# ------------------------------------------------------------------------------------------------------------------------------
# measured parity margins -> a JSON file (the numbers the tests measure must be auditable from the run that produced
# them, not only asserted).  Every record is {"case": ..., "device": ..., ...measured values...}; the file is rewritten on every
# record, so that a run that dies half-way still leaves what it measured.  Default location under the repo root, ONE FILE PER TIER:
# gpurun_out/parity_margins_cuda.json for records measured on a GPU (merged back from the GPU box by gpurun; the round's copy is committed
# as profiles/rNN_parity_margins.json) and gpurun_out/parity_margins_emu.json for the emulator tier's (round 5 shared one file and the CPU tier
# overwrote the GPU evidence); DN_PARITY_MARGINS overrides both.  tests/test_evidence_files.py rejects a committed
# profiles/r*_parity_margins.json that holds a record from any device but a GPU.
# ------------------------------------------------------------------------------------------------------------------------------
_MARGINS = {"cuda": [], "emu": []}


def margins_tier(device):
    return "cuda" if str(device).startswith("cuda") else "emu"


def margins_path(device="cuda"):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return os.environ.get("DN_PARITY_MARGINS", os.path.join(root, "gpurun_out", "parity_margins_%s.json" % margins_tier(device)))


def record_margin(case, device, **values):
    import json
    rec = {"case": case, "device": str(device)}
    for k, v in values.items():
        rec[k] = float(v) if isinstance(v, (float, int)) and not isinstance(v, bool) else v
    recs = _MARGINS[margins_tier(device)]
    recs.append(rec)
    try:
        path = margins_path(device)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(recs, f, indent=1, sort_keys=True)
    except OSError:
        pass
    return rec
