"""CPU: pins of the composite material-step oracle's (oracle/mat_step.py) own building blocks against fixtures produced by RUNNING the
reference (tests/golden/spec_render.npz: models/mat_nvdiffrast.py render + specular_reflectance values and autograd gradients;
tests/golden/render_loss.npz: models/loss.py RenderLoss values and gradients, L1 / L2 x stages 0 / 1 / 2), so that the GPU test
tests/test_gpu_mat_step_oracle.py compares the product with an oracle that is itself tied to the reference."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import mat_step as MS


def test_hammersley_points_match_reference(golden):
    g = golden("gen_dir.npz")
    for N in (1, 16, 64, 2048, 100):
        got = MS.hammersley_points(N)
        if N & (N - 1) == 0:
            assert np.array_equal(got, g["ham_%d" % N])
        else:
            assert np.abs(got - g["ham_%d" % N]).max() < 1e-7


def test_spec_render_restatement_matches_reference_values_and_grads(golden):
    g = golden("spec_render.npz")
    t = lambda k: torch.from_numpy(g[k])
    a = t("albedo").clone().requires_grad_(True)
    r = t("roughness").clone().requires_grad_(True)
    rgb, l = MS.spec_render(t("normal"), a, r, t("points"), t("irr"), t("cam"), t("shift"), int(g["S"]), t("Ls"))
    assert rel_l2(rgb.detach().numpy(), g["rgb"]) < 1e-6
    assert rel_l2(l.numpy(), g["l"]) < 1e-6
    (rgb * t("d_rgb")).sum().backward()
    assert rel_l2(a.grad.numpy(), g["d_albedo"]) < 1e-6
    assert rel_l2(r.grad.numpy(), g["d_roughness"]) < 1e-5
    h = MS.ggx_half_vectors(t("normal"), t("roughness"), t("shift"), int(g["S"]))
    assert rel_l2(h.numpy(), g["h"]) < 1e-6


@pytest.mark.parametrize("loss_type", ["L1", "L2"])
@pytest.mark.parametrize("stage", [0, 1, 2])
def test_render_loss_restatement_matches_reference(golden, loss_type, stage):
    g = golden("render_loss.npz")
    t = lambda k: torch.from_numpy(g[k])
    leaves = {k: t(k).clone().requires_grad_(True) for k in ("rgb", "albedo", "roughness", "roughness_womipmap")}
    preds = dict(leaves, empty_mask=t("empty_mask"))
    loss, seg = MS.render_loss(t("gt"), preds, t("gt_mask"), t("floor_max_mask"), t("seg_mask"), stage, t("room_seg_mask"), loss_type)
    k = "%s_s%d_" % (loss_type, stage)
    assert abs(float(loss.detach()) - float(g[k + "loss"])) < 1e-6 * max(1.0, abs(float(g[k + "loss"])))
    assert abs(float(seg) - float(g[k + "seg"])) < 1e-6 * max(1.0, abs(float(g[k + "seg"])))
    loss.backward()
    for name, key in (("d_rgb", "rgb"), ("d_albedo", "albedo"), ("d_roughness", "roughness"), ("d_roughness_womipmap", "roughness_womipmap")):
        ref = g[k + name]
        got = leaves[key].grad.numpy() if leaves[key].grad is not None else np.zeros_like(ref)
        if np.abs(ref).max() == 0:
            assert np.abs(got).max() == 0, name
        else:
            assert rel_l2(got, ref) < 1e-6, (name, rel_l2(got, ref))
