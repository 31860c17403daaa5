"""Pin the CPU oracle against the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Runs on CPU; no GPU, no /root/reference needed."""
import pytest
import torch

import helpers
from oracle import diffusionnet_oracle as orc

# fp32 oracle vs fp32 reference: same maths, different contraction order -> round-off only
FWD_TOL = 1e-5      # north_star tolerance (relative, fp32)
GRAD_TOL = 2e-4     # gradients accumulate over V; fp32-vs-fp64 floor of the reference itself is ~1e-4..7e-4


@pytest.mark.parametrize("name", helpers.golden_names())
def test_oracle_matches_reference_fp32(name):
    meta, params, inputs, masks, expect = helpers.load_golden(name)
    out, grads = orc.net_forward_backward(
        params, inputs, outputs_at=meta["ctor"]["outputs_at"], last_activation=helpers.activation_of(meta),
        keep_masks=helpers.group_masks(meta, params, masks), loss_weights=expect["loss_w"])
    assert out.shape == expect["out"].shape
    assert helpers.rel_max(out, expect["out"]) < FWD_TOL, name
    for k, g in expect["grads"].items():
        assert helpers.rel_l2(grads[k], g) < GRAD_TOL, (name, k, helpers.rel_l2(grads[k], g))


@pytest.mark.parametrize("name", helpers.golden_names())
def test_fp64_oracle_brackets_reference(name):
    """The fp64 oracle is the yard-stick for end-to-end checks: the fp32 reference must sit
    within fp32 round-off of it."""
    meta, params, inputs, masks, expect = helpers.load_golden(name, dtype=torch.float64)
    out, _ = orc.net_forward_backward(
        params, inputs, outputs_at=meta["ctor"]["outputs_at"], last_activation=helpers.activation_of(meta),
        keep_masks=helpers.group_masks(meta, params, masks), loss_weights=expect["loss_w"])
    assert helpers.rel_max(out, expect["out"]) < 2e-5, name


def test_oracle_error_conventions():
    meta, params, inputs, masks, expect = helpers.load_golden("nograd_v300_c32_k16")
    bad = dict(inputs)
    bad["x_in"] = inputs["x_in"][:, :2]
    with pytest.raises(ValueError):
        orc.net_forward(params, **bad)
    with pytest.raises(ValueError):
        orc.remap_outputs(torch.zeros(1, 4, 3), "corners")
