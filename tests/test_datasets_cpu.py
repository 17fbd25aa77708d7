"""Dataset adapters on the reference's raw on-disk layout: Pano2Cube golden-pinned against the reference's
own class; the cv2 image operations are restated from OpenCV's documented semantics and checked on hand-computable cases."""
import os

import numpy as np
import pytest
import torch


def test_pano2cube_matches_reference(golden):
    from texir_code_amd.pano2cube import Pano2Cube
    g = golden("pano2cube.npz")
    p2c = Pano2Cube(1, 64, 32, 8, 5)
    uv = torch.stack([u[0] for u in p2c.uv]).numpy()
    assert np.allclose(uv, g["uv"], atol=2e-6)
    pano = torch.from_numpy(g["pano"])
    assert np.allclose(p2c.Tocube(pano, mode="bilinear").numpy(), g["cube_bilinear"], atol=2e-5)
    near, ref = p2c.Tocube(pano, mode="nearest").numpy(), g["cube_nearest"]
    assert (near == ref).mean() > 0.995             # a sample that lands within 1e-6 of a pixel boundary may round the other way


def test_image_ops_follow_opencv_semantics():
    from texir_code_amd import imgops as cv
    a = np.arange(20, dtype=np.float32).reshape(4, 5)
    gx, gy = cv.sobel3(a)                                           # a = 5*y + x: d/dx = 1, d/dy = 5, kernel gain 8
    assert np.allclose(gx[:, 1:-1], 8.0) and np.allclose(gy[1:-1, :], 40.0)
    assert np.allclose(gx[:, 0], 0.0) and np.allclose(gy[0, :], 0.0)            # BORDER_REFLECT_101: symmetric -> zero gradient
    assert np.allclose(cv.magnitude(np.float32([3]), np.float32([4])), 5)
    assert np.allclose(cv.gray(np.float32([[[1, 0, 0], [0, 1, 0], [0, 0, 1]]]), "RGB"), [[0.299, 0.587, 0.114]])
    assert np.allclose(cv.gray(np.float32([[[1, 0, 0]]]), "BGR"), [[0.114]])
    m = np.full((9, 9, 1), 255, np.uint8)
    m[4, 4] = 0
    e = cv.erode(m, 5)
    assert e.shape == (9, 9) and (e[2:7, 2:7] == 0).all() and e.sum() == 255 * (81 - 25)      # borders are not eroded
    s = np.arange(12).reshape(3, 4)
    assert cv.resize_nearest(s, (8, 6)).tolist() == np.repeat(np.repeat(s, 2, 0), 2, 1).tolist()
    assert cv.resize_nearest(s, (2, 1)).tolist() == [[0, 2]]
    u = np.array([[0, 100], [200, 100]], np.uint8)
    r = cv.resize_linear(u, (4, 4))
    assert r.dtype == np.uint8 and r[0, 0] == 0 and r[0, 3] == 100 and r[3, 0] == 200 and r[1, 1] in (62, 63)     # .75*.25*100 + .25*.75*200 + .0625*100 = 62.5
    assert cv.resize_linear(np.float32([[1, 3]]), (4, 1)).tolist() == [[1.0, 1.5, 2.5, 3.0]]


def _write_raw_scene(root, n_views=2, h=32, w=64, novel=True):
    from texir_code_amd import cameras, io_formats as IO
    rng = np.random.default_rng(4)
    os.makedirs(os.path.join(root, "info"))
    os.makedirs(os.path.join(root, "hdr_texture"))
    ids = ["1657%03d" % i for i in range(n_views)]
    truth = {}
    for i in ids:
        os.makedirs(os.path.join(root, "derived", i))
        os.makedirs(os.path.join(root, "hdr", i))
        rgba = np.full((h, w, 4), 255, np.uint8)
        rgba[10:12, 20:22, 3] = 0                                       # an invalid blob, grows to 6x6 after the 5x5 erosion
        IO.write_png(os.path.join(root, "derived", i, "panoImage_orig.jpg"), rgba)      # (.jpg name, PNG content: decoded by magic)
        col = rng.uniform(0, 2, (h, w, 3)).astype(np.float32)
        IO.write_hdr(os.path.join(root, "hdr", i, "ccm.hdr"), col)
        seg = rng.integers(0, 49, (h // 2, w // 2)).astype(np.uint8)
        IO.write_png(os.path.join(root, "derived", i, "panoImage_gray.png"), seg)
        IO.write_png(os.path.join(root, "derived", i, "albedo.png"), np.full((h, w, 3), 128, np.uint8))
        IO.write_png(os.path.join(root, "derived", i, "roughness.png"), np.full((h, w), 64, np.uint8))
        truth[i] = (IO.read_hdr(os.path.join(root, "hdr", i, "ccm.hdr")), seg)
    E = cameras.grid_cameras(2)[:n_views]
    for name, idn in (("aligned.txt", "final_extrinsics.txt"), ("novel.txt", "novel_extrinsics.txt")):
        if name == "novel.txt" and not novel:
            continue
        with open(os.path.join(root, "info", name), "w") as f:
            f.write("\n".join(ids) + "\n")
        with open(os.path.join(root, "info", idn), "w") as f:
            f.write("%d \n" % n_views)
            for e in E:
                for r in e:
                    f.write(" ".join("%.9g" % x for x in r) + " \n")          # trailing blank as in the reference's files (dataset.py:408)
    return ids, truth


def test_raw_layout_adapters(tmp_path, golden):
    from texir_code_amd import datasets as D, plugin
    from texir_code_amd.pano2cube import Pano2Cube
    root = str(tmp_path / "scene")
    os.makedirs(root)
    ids, truth = _write_raw_scene(root)
    mesh = os.path.join(root, "hdr_texture", "out1.obj")                # reference layout: root = dirname(dirname(mesh))
    assert plugin.get_class("datasets.dataset.ImageCubeSyn") is D.ImageCubeSyn
    assert plugin.get_class("datasets.dataset.ImageCubeDerived") is D.ImageCubeDerived
    ds = D.ImageCubeSyn(mesh, [16, 32], hdr_exposure=1.0)
    assert ds.path_root == root and ds.ids == ids and len(ds) == 2 and ds.cube_res == 8
    assert ds.novel_ids == ids and len(ds.novel_images_items) == 2 and ds.novel_extrinsics_list[0].shape == (6, 4, 4)
    it = ds[1]
    assert set(it) == {"color", "mask", "segs", "cam_to_world", "id", "cam_position", "rgb_grad", "gt_albedo", "gt_roughness"}
    assert it["color"].shape == (6, 8, 8, 3) and it["mask"].shape == (6, 8, 8, 1) and it["segs"].shape == (6, 8, 8, 1)
    assert it["cam_to_world"].shape == (6, 4, 4) and it["id"] == ids[1]
    # every face pixel is the panorama pixel nearest to its direction (Pano2Cube grid), colour scaled by 2^exposure
    col, seg = truth[ids[1]]
    p2c = Pano2Cube(1, 64, 32, 8, 3)
    want = p2c.Tocube(torch.from_numpy(col * 2.0).permute(2, 0, 1)[None], mode="nearest")[0].reshape(6, 3, 8, 8).permute(0, 2, 3, 1)
    assert torch.equal(it["color"], want)
    seg_full = np.repeat(np.repeat(seg, 2, 0), 2, 1).astype(np.float32)
    wseg = p2c.Tocube(torch.from_numpy(seg_full)[None, None], mode="nearest")[0].reshape(6, 1, 8, 8).permute(0, 2, 3, 1)
    assert torch.equal(it["segs"], wseg)
    assert set(np.unique(it["mask"].numpy()).tolist()) <= {0.0, 1.0} and 0.0 < float(it["mask"].mean()) < 1.0
    assert it["rgb_grad"].shape == (6, 8, 8, 1) and float(it["rgb_grad"].min()) >= 0
    assert it["gt_albedo"].shape == (16, 32, 3) and np.allclose(it["gt_albedo"].numpy(), (128 / 255) ** 2.2, atol=1e-6)
    assert it["gt_roughness"].shape == (16, 32) and np.allclose(it["gt_roughness"].numpy(), 64 / 255, atol=1e-6)
    assert ds.images_items[0]["segs_pano"].shape == (16, 32, 1)
    # ImageCubeDerived: same files, no GT materials / novel views
    dd = D.ImageCubeDerived(mesh, [16, 32], hdr_exposure=0.0)
    assert set(dd[0]) == {"color", "mask", "segs", "cam_to_world", "id", "cam_position", "rgb_grad"}
    assert torch.allclose(dd[1]["color"] * 2.0, it["color"])


def test_synthetic_layout_still_resolves(tmp_path):
    from texir_code_amd import datasets as D
    root = str(tmp_path / "syn")
    D.write_synthetic_dataset(root, T=200, texel_res=8, tex_res=8, n_side=1)
    ds = D.ImageCubeSyn(os.path.join(root, "vrproc", "hdr_texture", "out1.obj"), [16, 32], 0.0)
    assert ds.path_root == root and len(ds.ids) == len(ds.extrinsics_list) > 0 and ds.novel_ids == []
