"""Evaluation re-use of the hot path (VERDICT r1 missing #1-#3): the evaluation model's render against the
reference's own output (golden, models/test_nvdiffrast.py), the traced diffuse term against diffuse_reflectance (golden,
models/mat_nvdiffrast.py:252-258), the four tester runners and the MatSyn tail end to end on a synthetic scene."""
import math
import os
import shutil
import types

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def test_diffuse_reflectance_matches_reference(golden):
    """diffuse_reflectance(query_irf(p, l), l, n, albedo, type) / N of the reference (both sample types, raw normals in n.l) ==
    texir_diffuse_irradiance x albedo / pi"""
    from texir_code_amd import scene as S
    g = golden("diffuse.npz")
    sc = S.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    for mode in ("uniform", "cosine"):
        E = S.diffuse_irradiance(sc, t(g["points"]), t(g["normal"]), t(g["shift_" + mode]), int(g["N"]), mode)
        d = (E * t(g["albedo"]) / math.pi).cpu().numpy()
        assert rel_l2(d, g["diffuse_" + mode]) < 1e-3, mode
        assert rel_l2(d, g["diffuse_" + mode]) < 5e-5, mode


@pytest.mark.parametrize("relight", [False, True])
def test_evaluation_render_matches_reference(golden, relight):
    """models/test_nvdiffrast.py render() at S = 256 (1e-6 BRDF floors; relighting: diffuse term traced at N0 = 64) captured from the
    reference, vs tester.test_model.MaterialModel.render drawing the same CPU-generator shifts"""
    from texir_code_amd import scene as S
    from texir_code_amd.tester.test_model import MaterialModel
    g = golden("test_render.npz")
    tag = "relight" if relight else "plain"
    m = MaterialModel.__new__(MaterialModel)
    torch.nn.Module.__init__(m)
    m.scene = S.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])
    m.device, m.sample_l, m.sample_type, m.relighting = m.scene.device, [int(g["N0"]), int(g["S"])], ["uniform", "importance"], relight
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    torch.manual_seed(70 + int(relight))
    res = m.render(t(g["normal"]), t(g["albedo"]), t(g["roughness"]), t(g["points"]), t(g["cam"]), t(g["irr"]))
    torch.manual_seed(70 + int(relight))                       # the draws render() made are the fixture's
    P = g["normal"].reshape(-1, 3).shape[0]
    if relight:
        assert np.array_equal(torch.rand(P, 1, 2).reshape(P, 2).numpy(), g["shift_diff_relight"])
    assert np.array_equal(torch.rand(P, 1, 2).reshape(P, 2).numpy(), g["shift_spec_" + tag])
    assert rel_l2(res["rgb"].cpu().numpy(), g["rgb_" + tag]) < 1e-3
    assert rel_l2(res["rgb"].cpu().numpy(), g["rgb_" + tag]) < 1e-4


def test_render_loss_image_term_variants(golden):
    """RenderLoss(loss_type = psnr | ssim | msssim), stage 0 (models/loss.py:65-73,117-140): SegLoss from the fused kernel + the image
    term in torch; gradients flow to rgb and albedo; stages 1/2 are refused like the reference's shapes refuse them"""
    from texir_code_amd.loss import RenderLoss
    from texir_code_amd import metrics as M
    from texir_code_amd.models import hdr_scale
    g = golden("render_loss.npz")
    t = lambda k: torch.from_numpy(g[k]).cuda()
    base = RenderLoss("L1", 1)
    rgb0, alb0 = t("rgb"), t("albedo")
    preds = {"rgb": rgb0, "albedo": alb0, "roughness": t("roughness"), "roughness_womipmap": t("roughness_womipmap"), "empty_mask": t("empty_mask")}
    l1_total, l1_seg = base(t("gt"), preds, t("gt_mask"), t("floor_max_mask"), t("seg_mask"), 0)
    for lt in ("psnr", "ssim"):
        rgb, alb = rgb0.clone().requires_grad_(True), alb0.clone().requires_grad_(True)
        p2 = dict(preds, rgb=rgb, albedo=alb)
        loss, seg = RenderLoss(lt, 1)(t("gt"), p2, t("gt_mask"), t("floor_max_mask"), t("seg_mask"), 0)
        assert abs(seg - l1_seg) < 1e-6 * max(1.0, abs(l1_seg))                    # the same SegLoss
        a = hdr_scale(rgb0 * t("empty_mask") * t("gt_mask")).permute(0, 3, 1, 2)
        b = hdr_scale(t("gt") * t("gt_mask")).permute(0, 3, 1, 2)
        want = -M.mse_to_psnr(torch.mean((b - a) ** 2)) if lt == "psnr" else 1.0 - M.ssim(b, a)
        assert abs(float(loss) - (float(want) + seg)) < 1e-5 * max(1.0, abs(float(loss)))
        loss.backward()
        assert torch.isfinite(rgb.grad).all() and float(rgb.grad.abs().sum()) > 0 and float(alb.grad.abs().sum()) > 0
    with pytest.raises(ValueError):
        RenderLoss("psnr", 1)(t("gt"), preds, t("gt_mask"), t("floor_max_mask"), t("seg_mask"), 1)
    with pytest.raises(Exception):
        RenderLoss("nope", 1)


def test_tester_runners_and_matsyn_tail_end_to_end(tmp_path):
    """--trainstage IrrT -> MatSyn (incl. its render_calculate tail at 256 specular samples) -> --teststage Error / Editing /
    Relighting / View on one synthetic scene, through the two CLIs"""
    from texir_code_amd import conf as C, datasets as D, io_formats as IO
    from texir_code_amd.trainer import exp_runner as ER
    from texir_code_amd.tester import exp_runner as TR
    from texir_code_amd.trainer.train_material import MatTrainSynRunner
    root = str(tmp_path / "ds")
    sc = D.write_synthetic_dataset(root, T=2000, texel_res=64, tex_res=64, n_side=2)
    conf_irt = str(tmp_path / "irt.conf")
    D.write_conf(conf_irt, root, cube_res=16, spp=(64, 16), model="irt")
    ER.main(["--conf", conf_irt, "--trainstage", "IrrT", "--gpu", "0"])
    mesh_dir = os.path.join(root, "vrproc", "hdr_texture")
    shutil.copy(os.path.join(mesh_dir, "0_irr_texture.hdr"), os.path.join(mesh_dir, "irt.hdr"))
    conf_mat = str(tmp_path / "mat.conf")
    D.write_conf(conf_mat, root, cube_res=16, spp=(64, 16), albedo_res=64, rough_res=64, epochs=1, model="mat")
    D.render_gt_views(root, C.parse_file(conf_mat), sc, 64, 64)
    assert ER.runner_class("MatSyn") is MatTrainSynRunner
    exps = str(tmp_path / "exps")
    runner = MatTrainSynRunner(conf=conf_mat, exps_folder_name=exps, expname="t", frame_skip=1, max_niters=10, is_continue=False,
                               timestamp="latest", checkpoint="latest", gpu_index=0)
    runner.run()
    assert runner.model.sample_l[1] == 256                                  # train_material_syn.py:735
    mt = runner.metrics
    assert all(np.isfinite(v) for v in mt.values()) and 0 <= mt["mse"] < 1 and mt["psnr"] > 5 and 0.5 <= mt["ssim"] <= 1
    out = {}
    # the reference ships a separate conf for the fly-through (configs/test_novel.conf: dataset_class = datasets.dataset.ImageCubeNovel)
    conf_novel = str(tmp_path / "novel.conf")
    txt = open(conf_mat).read()
    head, tail = txt.split("test{", 1)
    open(conf_novel, "w").write(head + "test{" + tail.replace("datasets.dataset.ImageCubeSyn", "datasets.dataset.ImageCubeNovel", 1))
    for stage in ("Error", "Editing", "Relighting", "View"):
        r = TR.main(["--conf", conf_novel if stage == "View" else conf_mat, "--exps_folder_name", exps, "--expname", "t", "--teststage", stage, "--gpu", "0"])
        assert r.outputs and all(os.path.exists(p) for p in r.outputs), stage
        out[stage] = r
    assert np.isfinite(list(out["Error"].metrics.values())).all() and out["Error"].metrics["psnr"] > 5
    # Error re-renders the training views with the trained textures at the test{} sample count: same scene, same views as MatSyn's tail
    assert len(out["Error"].outputs) == 4 and len(out["Relighting"].outputs) == 4 and len(out["View"].outputs) == 60
    img = IO.read_hdr(out["Error"].outputs[0])
    assert img.shape == (32, 64, 3) and np.isfinite(img).all() and img.max() > 0
    # Editing: 21 albedo key frames + 21 roughness key frames of view 0; recolouring floor and walls changes the picture
    assert len(out["Editing"].outputs) == 42
    e0, e7 = IO.read_png(out["Editing"].outputs[0]), IO.read_png(out["Editing"].outputs[7])
    assert e0.shape == (32, 64, 3) and (e0.astype(int) - e7.astype(int)).__abs__().mean() > 1.0
    # Relighting: light sources recoloured to (2.14, 1.38, 0.2) x 2^exposure -> the re-lit picture is warmer than the re-rendering
    rl, er = IO.read_hdr(out["Relighting"].outputs[0]), IO.read_hdr(out["Error"].outputs[0])
    assert np.isfinite(rl).all() and rl[..., 0].mean() / max(rl[..., 2].mean(), 1e-9) > 1.5 * er[..., 0].mean() / max(er[..., 2].mean(), 1e-9)
