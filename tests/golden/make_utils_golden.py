"""Golden vectors for ``diffusion_net.utils`` from the *reference itself* (dev container only; needs /root/reference):
seeded ``random_rotation_matrix`` / ``random_rotate_points``, ``hash_arrays`` and the scipy<->torch sparse round trip.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_utils_golden.py
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402

ref = import_reference()
out = {}
for seed in range(5):
    out["rot_%d" % seed] = ref.utils.random_rotation_matrix(np.random.RandomState(seed))
pts = torch.from_numpy(np.random.RandomState(7).randn(50, 3).astype(np.float32))
out["pts"] = pts.numpy()
out["pts_rot_seed3"] = ref.utils.random_rotate_points(pts, np.random.RandomState(3)).numpy()
a = np.random.RandomState(1).randn(40, 3).astype(np.float32)
b = np.random.RandomState(2).randint(0, 40, size=(70, 3)).astype(np.int64)
out["hash_a"], out["hash_b"] = a, b
out["hash_hex"] = np.array(ref.utils.hash_arrays((a, b)))
m = sp.random(30, 20, density=0.2, random_state=np.random.RandomState(4), format="csr", dtype=np.float64)
t = ref.utils.sparse_np_to_torch(m)
out["sp_data"], out["sp_indices"], out["sp_indptr"] = m.data, m.indices, m.indptr
out["sp_t_indices"], out["sp_t_values"] = t.indices().numpy(), t.values().numpy()
back = ref.utils.sparse_torch_to_np(t)
out["sp_back_data"], out["sp_back_indices"], out["sp_back_indptr"] = back.data, back.indices, back.indptr
np.savez_compressed(os.path.join(HERE, "utils_ref.npz"), **out)
print("wrote utils_ref.npz", {k: getattr(v, "shape", None) for k, v in out.items()})
