"""CPU tier: the C-ABI library loads and exports every symbol include/diffnet_hip.h declares; the
product path refuses to run without a ROCm device; host-side packing logic."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "diffnet_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dn_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def product_lib():
    import __graft_entry__
    __graft_entry__.build()
    from diffusion_net import _hip
    assert os.path.exists(_hip.LIB_PATH)
    return ctypes.CDLL(_hip.LIB_PATH)


def test_library_exports_every_declared_symbol(product_lib):
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(product_lib, n), n
    from diffusion_net import _hip
    assert sorted(_hip.EXPORTED_SYMBOLS) == names      # the ctypes binding covers the whole header
    assert product_lib.dn_version() >= 100 and product_lib.dn_tile_rows() == 128


def test_struct_layouts_match_header():
    from diffusion_net import _hip
    P = ctypes.sizeof(ctypes.c_void_p)
    assert ctypes.sizeof(_hip.MeshBatchStruct) == 6 * 4 + 15 * P
    assert ctypes.sizeof(_hip.BlockParamsStruct) == (4 + 9) * 4 + 4 + (3 + 3 * 8) * P + 8 + P   # 4 B padding before the pointers, uint64 drop_seed, then the device seed pointer
    assert ctypes.sizeof(_hip.BlockSavedStruct) == (7 + 8) * P
    assert ctypes.sizeof(_hip.BlockGradsStruct) == (4 + 16) * P
    assert _hip.TILE_DTYPE.itemsize == 16


def test_product_path_refuses_cpu_tensors(product_lib):
    import diffusion_net
    from diffusion_net import synthetic
    m = synthetic.make_mesh_operators(64, 8, seed=0)
    model = diffusion_net.layers.DiffusionNet(3, 4, C_width=32, N_block=1)
    with pytest.raises(RuntimeError, match="ROCm device"):
        model(m["verts"], m["mass"], evals=m["evals"], evecs=m["evecs"], gradX=m["gradX"], gradY=m["gradY"])


def test_error_conventions(product_lib):
    import diffusion_net
    D = diffusion_net.layers.DiffusionNet
    with pytest.raises(ValueError):
        D(3, 4, outputs_at="corners")                      # layers.py:278
    with pytest.raises(ValueError):
        D(3, 4, diffusion_method="explicit")               # layers.py:288
    model = D(3, 4, C_width=32, N_block=1)
    with pytest.raises(ValueError):
        model(torch.zeros(10, 5), torch.ones(10))          # layers.py:343-344
    with pytest.raises(ValueError):
        model(torch.zeros(2, 2, 10, 3), torch.ones(10))    # layers.py:363


def test_state_dict_keys_match_reference_layout():
    import diffusion_net
    model = diffusion_net.layers.DiffusionNet(3, 8, C_width=128, N_block=4)
    keys = list(model.state_dict().keys())
    assert len(keys) == 40 and sum(p.numel() for p in model.parameters()) == 462344
    assert "block_2.mlp.miniMLP_mlp_layer_001.weight" in keys
    assert "block_0.gradient_features.A_im.weight" in keys and "block_3.diffusion.diffusion_time" in keys
    norot = diffusion_net.layers.DiffusionNet(3, 8, C_width=32, N_block=1, with_gradient_rotations=False)
    assert "block_0.gradient_features.A.weight" in norot.state_dict()
    assert float(model.block_0.diffusion.diffusion_time.abs().max()) == 0.0             # layers.py:41


def test_tile_and_chunk_tables():
    from diffusion_net.batch import build_tables, default_chunk_rows
    tiles, chunks, mco, mrows = build_tables([300, 128, 1], chunk_rows=256, tile_rows=128)
    assert tiles[:, 1].sum() == 429 and chunks[:, 1].sum() == 429
    assert (tiles[:, 1] <= 128).all() and (chunks[:, 1] <= 256).all()
    assert list(mco) == [0, 2, 3, 4] and list(mrows[:, 0]) == [0, 300, 428]
    # tiles never straddle meshes
    for r0, n, m, _ in tiles:
        assert mrows[m, 0] <= r0 and r0 + n <= mrows[m, 0] + mrows[m, 1]
    assert default_chunk_rows(10_000) == 128 and default_chunk_rows(160_000) % 32 == 0
    assert default_chunk_rows(10 ** 7) == 1024


def test_synthetic_operators_are_consistent():
    from diffusion_net import synthetic
    m = synthetic.make_mesh_operators(500, 16, seed=1)
    gram = (m["evecs"].double().T * m["mass"].double()) @ m["evecs"].double()
    assert torch.allclose(gram, torch.eye(16, dtype=torch.float64), atol=1e-5)           # Phi^T M Phi = I
    assert torch.equal(m["gradX"].indices(), m["gradY"].indices())
    assert m["gradX"].to_dense().sum(1).abs().max() < 1e-3                                 # zero row sums
    nnz_per_row = np.bincount(m["gradX"].indices()[0].numpy(), minlength=500)
    assert (nnz_per_row == 7).all() and m["faces"].shape == (1000, 3)


@pytest.mark.parametrize("name", ["ckpt_human_seg_xyz_v600", "ckpt_human_seg_hks_v600"])
def test_shipped_checkpoints_load_strict(name):
    """The reference's trained checkpoints (human_segmentation_original/pretrained_models/*.pth, tensors carried by the golden
    fixtures) load into the drop-in module with strict=True, as human_segmentation_original.py:83 does, key for key."""
    import helpers
    import diffusion_net
    meta, params, _, _, _ = helpers.load_golden(name)
    assert meta["checkpoint"].endswith(".pth") and len(params) == 40
    model = diffusion_net.layers.DiffusionNet(last_activation=helpers.activation_of(meta), **meta["ctor"])
    missing = model.load_state_dict(params, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    for k, v in model.state_dict().items():
        assert torch.equal(v, params[k]), k
    assert sum(p.numel() for p in model.parameters()) == 462344 + (meta["ctor"]["C_in"] - 3) * 128
