"""Host-side operator precompute: pinned against the reference's own frame / gradient functions through
tests/golden/geom_sphere300.npz, and through invariants for the parts the reference delegates to potpourri3d/ARPACK
(eigenvectors are only defined up to sign / rotation in degenerate eigenspaces)."""
import os

import numpy as np
import scipy.sparse as sp
import torch

import helpers
from diffusion_net import geometry, precompute, synthetic


def _load():
    z = np.load(os.path.join(helpers.GOLDEN_DIR, "geom_sphere300.npz"))
    return z, z["verts"], z["faces"].astype(np.int64)


def test_frames_and_gradient_match_reference_functions():
    z, verts, faces = _load()
    frames = precompute.tangent_frames(precompute.vertex_normals(verts, faces))
    assert np.abs(frames - z["frames"]).max() < 5e-6            # reference ran in fp32
    Lc = precompute.cotan_laplacian(verts, faces).tocoo()
    grad = precompute.gradient_operator(verts, frames, np.stack([Lc.row, Lc.col])).tocsr()
    ref = sp.coo_matrix((z["grad_re"] + 1j * z["grad_im"], (z["grad_row"], z["grad_col"])), shape=grad.shape).tocsr()
    diff = (grad - ref)
    assert abs(diff).max() < 2e-4 * abs(ref).max()              # fp32 edge vectors in the reference
    assert (grad != 0).nnz == (ref != 0).nnz
    assert abs(grad.sum(axis=1)).max() < 1e-9                   # gradient of a constant is zero


def test_laplacian_mass_and_eigenbasis_invariants():
    _, verts, faces = _load()
    L = precompute.cotan_laplacian(verts, faces)
    assert abs(L - L.T).max() < 1e-12 and abs(L.sum(axis=1)).max() < 1e-9
    mass = precompute.vertex_areas(verts, faces)
    c = verts[faces]
    area = 0.5 * np.linalg.norm(np.cross(c[:, 1] - c[:, 0], c[:, 2] - c[:, 0]), axis=1).sum()
    assert abs(mass.sum() - area) < 1e-9 * area and (mass > 0).all()
    evals, evecs = precompute.laplacian_eigenbasis(L, mass, 16)
    assert evals[0] < 1e-6 and np.all(np.diff(evals) >= -1e-9)
    gram = evecs.T @ (mass[:, None] * evecs)
    assert np.abs(gram - np.eye(16)).max() < 1e-6               # Phi^T M Phi = I
    resid = L @ evecs - (mass[:, None] * evecs) * evals[None]
    assert np.abs(resid).max() < 1e-6 * max(1.0, evals[-1])     # L Phi = M Phi Lambda


def test_get_operators_contract_and_cache(tmp_path):
    _, verts, faces = _load()
    vt, ft = torch.from_numpy(verts).float(), torch.from_numpy(faces)
    vt = geometry.normalize_positions(vt)
    out = geometry.get_operators(vt, ft, k_eig=12, op_cache_dir=str(tmp_path))
    frames, mass, L, evals, evecs, gX, gY = out
    assert frames.shape == (300, 3, 3) and mass.shape == (300,) and evals.shape == (12,) and evecs.shape == (300, 12)
    assert L.is_sparse and gX.is_sparse and torch.equal(gX.coalesce().indices(), gY.coalesce().indices())
    assert all(t.dtype == torch.float32 for t in (frames, mass, evals, evecs)) and gX.dtype == torch.float32
    files = os.listdir(tmp_path)
    assert len(files) == 1 and files[0].endswith("_0.npz")
    z = np.load(os.path.join(tmp_path, files[0]))
    for key in ("verts", "frames", "faces", "k_eig", "mass", "L_data", "L_indices", "L_indptr", "L_shape", "evals", "evecs",
                "gradX_data", "gradX_indices", "gradX_indptr", "gradX_shape", "gradY_data", "gradY_indices", "gradY_indptr",
                "gradY_shape"):
        assert key in z, key                                      # the reference's cache layout (geometry.py:548-568)
    again = geometry.get_operators(vt, ft, k_eig=8, op_cache_dir=str(tmp_path))     # hit: truncated to 8 eigenpairs
    assert again[3].shape == (8,) and torch.equal(again[4], evecs[:, :8]) and torch.equal(again[1], mass)
    assert torch.equal(again[5].coalesce().values(), gX.coalesce().values())
    more = geometry.get_operators(vt, ft, k_eig=20, op_cache_dir=str(tmp_path))     # too few cached -> rebuilt
    assert more[3].shape == (20,) and len(os.listdir(tmp_path)) == 1


def test_hks_and_normalize():
    _, verts, faces = _load()
    vt = torch.from_numpy(verts).float()
    pos = geometry.normalize_positions(vt)
    assert abs(float(pos.norm(dim=-1).max()) - 1.0) < 1e-6 and float(pos.mean(0).abs().max()) < 1e-6
    ev, ph = torch.rand(6).sort().values, torch.randn(50, 6)
    hks = geometry.compute_hks_autoscale(ev, ph, 4)
    sc = torch.logspace(-2, 0.0, steps=4)
    ref = torch.stack([(torch.exp(-ev * s) * ph * ph).sum(-1) for s in sc], -1)
    assert hks.shape == (50, 4) and torch.allclose(hks, ref, atol=1e-6)


def test_reads_operator_cache_written_by_the_reference(tmp_path):
    """tests/golden/refcache_<sha1>_0.npz was written by the reference's geometry.get_operators (make_golden.py, its own hashing,
    key names and CSC triplets).  Ours must find it under the same file name for the same mesh and return its content -- the
    eigenvectors of a recomputation would differ by signs/rotations, so equality proves the file was read."""
    import glob
    import shutil
    import scipy.sparse as sp
    from diffusion_net import precompute, synthetic
    src = glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refcache_*_0.npz"))
    assert len(src) == 1
    name = os.path.basename(src[0])[len("refcache_"):]
    shutil.copy(src[0], tmp_path / name)
    z = np.load(src[0], allow_pickle=True)
    verts, faces = torch.from_numpy(z["verts"]).float(), torch.from_numpy(z["faces"]).long()
    K = int(z["k_eig"].item())
    assert precompute.hash_arrays((z["verts"], z["faces"])) + "_0.npz" == name          # same key derivation (utils.py:71-76)
    before = sorted(os.listdir(tmp_path))
    frames, mass, L, evals, evecs, gX, gY = precompute.get_operators(verts, faces, k_eig=K, op_cache_dir=str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == before                                           # nothing rebuilt or rewritten
    assert np.array_equal(evecs.numpy(), z["evecs"][:, :K]) and np.array_equal(evals.numpy(), z["evals"][:K])
    assert np.array_equal(mass.numpy(), z["mass"]) and np.array_equal(frames.numpy(), z["frames"])
    ref_gx = sp.csc_matrix((z["gradX_data"], z["gradX_indices"], z["gradX_indptr"]), shape=tuple(z["gradX_shape"])).toarray()
    assert np.array_equal(gX.to_dense().numpy(), ref_gx)
    # and the other direction: a file written by ours carries the same keys and dtypes the reference's reader indexes
    out_dir = tmp_path / "mine"
    v2, f2 = synthetic.sphere_mesh(150, seed=3)
    precompute.get_operators(torch.from_numpy(v2).float(), torch.from_numpy(f2).long(), k_eig=8, op_cache_dir=str(out_dir))
    mine = np.load(str(out_dir / os.listdir(out_dir)[0]), allow_pickle=True)
    assert sorted(mine.files) == sorted(z.files)
    for k in z.files:
        assert mine[k].dtype == z[k].dtype, (k, mine[k].dtype, z[k].dtype)
