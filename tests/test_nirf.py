"""NIrF slice: the reference's tracer_o3d_irrf.TracerO3d.forward, MatNetwork and IRFLoss, run stub-imported
in the build container (oracle/make_golden.py nirf), are the golden; GPU tests call the IrT kernel through the C-ABI."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2


def _load_net(g, dims=(64, 64, 64, 64)):
    from texir_code_amd.nirf import MatNetwork
    net = MatNetwork(points_multires=10, p_input_dim=3, p_out_dim=3, dims=list(dims))
    sd = {k: torch.from_numpy(g["w_" + k.replace(".", "_")]) for k in net.state_dict().keys()}
    net.load_state_dict(sd)          # same state_dict keys as the reference's module
    return net


def test_matnetwork_and_irfloss_match_reference(golden):
    from texir_code_amd.nirf import IRFLoss, get_embedder
    g = golden("nirf.npz")
    fn, dim = get_embedder(10)
    assert dim == 63 and fn(torch.zeros(2, 3)).shape == (2, 63)
    net = _load_net(g)
    pred = net(torch.from_numpy(g["points"]))
    assert np.allclose(pred.detach().numpy(), g["pred"], rtol=1e-5, atol=1e-5)
    res = {"gt": torch.from_numpy(g["gt"]), "pred": torch.from_numpy(g["pred"])}
    assert abs(IRFLoss("L1")(res).item() - float(g["loss_l1"])) < 1e-6 * max(1.0, abs(float(g["loss_l1"])))
    assert abs(IRFLoss("L2")(res).item() - float(g["loss_l2"])) < 1e-6 * max(1.0, abs(float(g["loss_l2"])))
    with pytest.raises(Exception):
        IRFLoss("huber")


def test_meshpoint_samples_lie_on_the_offset_surface(tmp_path):
    from texir_code_amd import datasets as D, io_formats as IO, plugin, synth
    sc = synth.make_scene(200, seed=5, tex_res=8)
    path = str(tmp_path / "out1.obj")
    IO.write_obj(path, sc["verts"], sc["tris"], sc["tri_uvs"])
    assert plugin.get_class("datasets.dataset.MeshPoint") is D.MeshPoint
    np.random.seed(1)
    ds = D.MeshPoint(path, 500)
    assert len(ds) == 500 and ds.get_AABB().shape == (2, 3)
    ds.change_points()
    p, n = ds.points, ds.normals
    assert p.shape == (500, 3) and np.allclose(np.linalg.norm(n, axis=-1), 1, atol=1e-5)
    # every sample, moved back by delta along its normal, lies on some triangle's plane inside the AABB
    q = p - 1e-2 * n
    assert np.all(q >= ds.AABB[0] - 1e-4) and np.all(q <= ds.AABB[1] + 1e-4)
    v = sc["verts"][sc["tris"]]
    fn = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    fn /= np.linalg.norm(fn, axis=-1, keepdims=True)
    dist = np.abs(((q[:, None, :] - v[None, :, 0, :]) * fn[None]).sum(-1)).min(1)
    assert dist.max() < 1e-4
    first = ds.points.copy()
    ds.change_points()
    assert not np.array_equal(first, ds.points)
    s = ds[3]
    assert s["point"].shape == (3,) and s["normal"].dtype == torch.float32


@pytest.mark.gpu
def test_nirf_gt_matches_reference_forward(golden):
    from texir_code_amd import scene as S
    from texir_code_amd.nirf import TracerO3dIrrF
    from texir_code_amd.conf import parse_string
    g = golden("nirf.npz")
    conf = parse_string("train{ path_mesh_open3d = none\n std_jit = 5e-2 }\nmodels{ irrf_network{ dims = [64,64,64,64]\n p_input_dim = 3\n p_out_dim = 3 } }")
    sc = S.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])
    m = TracerO3dIrrF(conf, scene=sc).cuda()
    m.ir_radiance_network.load_state_dict(_load_net(g).state_dict())
    pts, nrm = torch.from_numpy(g["points"]).cuda(), torch.from_numpy(g["normals"]).cuda()
    gt = m.trace_gt(pts, nrm, g["res"].tolist(), shift=torch.from_numpy(g["shift"]))
    assert rel_l2(gt.cpu().numpy(), g["gt"]) < 1e-3          # north_star tolerance
    assert rel_l2(gt.cpu().numpy(), g["gt"]) < 2e-5
    # forward(): same CPU-generator stream for the shifts as the reference (torch.rand(b,1,1,2) after manual_seed)
    torch.manual_seed(123)
    res = m(pts, nrm, g["res"].tolist())
    assert set(res) == {"gt", "pred", "pred_jit"}
    assert rel_l2(res["gt"].cpu().numpy(), g["gt"]) < 2e-5
    assert np.allclose(res["pred"].detach().cpu().numpy(), g["pred"], rtol=1e-4, atol=1e-4)
    assert set(m(pts, nrm, g["res"].tolist(), True)) == {"pred", "pred_jit"}


@pytest.mark.gpu
def test_irrf_runner_end_to_end(tmp_path):
    """--trainstage IRRF on a synthetic scene: loss goes down, checkpoint + validation panorama are written"""
    from texir_code_amd import datasets as D
    from texir_code_amd.trainer import exp_runner as ER
    root = str(tmp_path / "scene")
    D.write_synthetic_dataset(root, T=2000, texel_res=32, tex_res=64, n_side=1)
    mesh = os.path.join(root, "vrproc", "hdr_texture", "out1.obj")
    conf = str(tmp_path / "irrf.conf")
    with open(conf, "w") as f:
        f.write("""train{
    expname = syn
    dataset_class = datasets.dataset.MeshPoint
    model_class = models.tracer_o3d_irrf.TracerO3d
    irf_loss_class = models.loss.IRFLoss
    plot_freq = 40
    ckpt_freq = 30
    irf_epoch = 3
    irf_learning_rate = 2e-3
    irf_sched_step = 100
    irf_sched_factor = 0.5
    std_jit = 5e-2
    env_res = [8,16]
    val_sample_res = [4,8]
    batch_size = 256
    samples_point_mesh = 4096
    is_hdr_texture = True
    hdr_exposure = 0
    path_mesh_open3d = %s
}
val{
    dataset_class = datasets.dataset.ImageMeshPoint
    env_res = [8,16]
    batch_size = 64
}
irf_loss{
    loss_type = L1
}
models{
    irrf_network{
        dims = [64, 64, 64]
        p_input_dim = 3
        p_out_dim = 3
    }
}
""" % mesh)
    assert ER.runner_class("IRRF").__name__ == "IRRFTrainRunner"
    torch.manual_seed(3)
    np.random.seed(3)
    r = ER.runner_class("IRRF")(conf=conf, exps_folder_name="exps", expname="t", frame_skip=1, max_niters=60, is_continue=False,
                                timestamp="latest", checkpoint="latest", gpu_index=0, exps_root=str(tmp_path))
    r.run()
    assert r.cur_iter == 60 and len(r.losses) == 2 and all(np.isfinite(r.losses)) and r.losses[-1] < r.losses[0]
    assert os.path.exists(os.path.join(r.checkpoints_path, "ModelParameters", "latest.pth"))
    assert sorted(os.listdir(r.plots_dir)) == ["irf_0.hdr", "irf_40.hdr"]
    # continue from the checkpoint
    r2 = ER.runner_class("IRRF")(conf=conf, exps_folder_name="exps", expname="t", frame_skip=1, max_niters=10, is_continue=True,
                                 timestamp="latest", checkpoint="latest", gpu_index=0, exps_root=str(tmp_path))
    a = r.model.state_dict()
    for k, v in r2.model.state_dict().items():
        assert torch.equal(v.cpu(), a[k].cpu())
