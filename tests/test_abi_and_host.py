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


def test_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors of the C structs against the header itself: a C program built from include/diffnet_hip.h prints sizeof and every
    field offset, which must equal what the Python binding uses."""
    import subprocess
    from diffusion_net import _hip
    pairs = (("dn_mesh_batch_t", _hip.MeshBatchStruct), ("dn_block_params_t", _hip.BlockParamsStruct),
             ("dn_block_saved_t", _hip.BlockSavedStruct), ("dn_block_grads_t", _hip.BlockGradsStruct))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "diffnet_hip.h"', 'int main(void) {']
    for cname, st in pairs:
        lines.append('printf("%s.sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in st._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines.append('printf("dn_tile_t.sizeof %zu\\n", sizeof(dn_tile_t)); printf("DN_MAX_MLP_LAYERS %d\\n", DN_MAX_MLP_LAYERS); '
                 'printf("DN_BLOCK_AMAX_WORDS %d\\n", DN_BLOCK_AMAX_WORDS); return 0; }')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True, capture_output=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, st in pairs:
        assert int(got[cname + ".sizeof"]) == ctypes.sizeof(st), cname
        for fname, _ in st._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(st, fname).offset, (cname, fname)
    assert int(got["dn_tile_t.sizeof"]) == _hip.TILE_DTYPE.itemsize == 16
    assert int(got["DN_MAX_MLP_LAYERS"]) == _hip.MAX_MLP and int(got["DN_BLOCK_AMAX_WORDS"]) == _hip.BLOCK_AMAX_WORDS


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
    # balanced tables: as many chunks as workgroup slots (when the meshes allow), nearly equal sizes, every mesh covered without gaps
    from diffusion_net.batch import balanced_chunk_rows
    sizes = [9057, 9803, 9519, 10988, 10001, 9333, 10750, 9100, 10640, 9999, 9001, 10900, 9777, 10321, 9650, 10400]
    per = balanced_chunk_rows(sizes, 256)
    t2, c2, mco2, _ = build_tables(sizes, per, tile_rows=128)
    assert 250 <= c2.shape[0] <= 256 and all(r % 32 == 0 for r in per)
    assert int(c2[:, 1].sum()) == sum(sizes) and int(c2[:, 1].max()) <= 1.25 * sum(sizes) / 256 + 32
    assert all(int(c2[i, 0]) + int(c2[i, 1]) == int(c2[i + 1, 0]) for i in range(c2.shape[0] - 1))
    assert balanced_chunk_rows([100, 40], 256) == [128, 64] and balanced_chunk_rows([300] * 400, 256) == [320] * 400   # tiny meshes / more meshes than slots


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


def test_integration_stub_struct_matches_the_binding():
    """INTEGRATION.md shows the ctypes stub a maintainer would add to the reference: its dn_mesh_batch_t must be the struct the library reads
    (a stub with fewer fields hands over a shorter struct and the library reads past its end)."""
    import ctypes as C
    import re
    from diffusion_net import _hip
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class dn_mesh_batch_t\(C\.Structure\):.*?\n(    _fields_ = .*?\n)\n", text, re.S)
    assert m, "stub not found"
    ns = {"C": C}
    exec("class dn_mesh_batch_t(C.Structure):\n" + m.group(1), ns)
    stub = ns["dn_mesh_batch_t"]
    assert [f[0] for f in stub._fields_] == [f[0] for f in _hip.MeshBatchStruct._fields_]
    assert C.sizeof(stub) == C.sizeof(_hip.MeshBatchStruct)
