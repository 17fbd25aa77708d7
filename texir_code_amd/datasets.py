"""Dataset side of the hot path.

SynCubeDataset exposes what the runners read from datasets.dataset.ImageCubeDerived / ImageCubeSyn
(datasets/dataset.py:352-549, 669-893): .ids, .extrinsics_list (per-view [6,4,4] mvp), .cam_position_list, .images_items
and the __getitem__ dict keys -- for dataset directories in the layout write_synthetic_dataset() produces (the reference's
own dataset is private, README.md:21-34).  ids / extrinsics use the reference's text formats (info/aligned.txt,
info/final_extrinsics.txt); per-view cube-face images are stored as cube/<id>.npz.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import cameras, io_formats as IO


def find_root(path_mesh):
    """the directory that holds info/aligned.txt: the reference uses dirname(dirname(path_mesh)) (datasets/dataset.py:356, mesh at
    <root>/hdr_texture/out1.obj); write_synthetic_dataset() nests the mesh one level deeper (<root>/vrproc/hdr_texture/out1.obj)"""
    two = os.path.dirname(os.path.dirname(path_mesh))
    if os.path.exists(os.path.join(two, "info", "aligned.txt")):
        return two
    return os.path.dirname(two)


class SynCubeDataset(Dataset):
    def __init__(self, path_mesh, resolution=[1000, 2000], hdr_exposure=1.0):
        super().__init__()
        self.path_mesh = path_mesh
        self.path_root = find_root(path_mesh)
        self.resolution = resolution
        self.cube_res = int(resolution[1] / 4)
        self.hdr_exposure = hdr_exposure
        self.ids = self.read_id()
        self.extrinsics_list, self.cam_position_list = self.read_extrinsic()
        self.images_items = self.read_images(self.ids)

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, index):
        it = self.images_items[index]
        return {"color": it["color"], "mask": it["mask"], "segs": it["segs"], "cam_to_world": self.extrinsics_list[index],
                "id": self.ids[index], "cam_position": self.cam_position_list[index]}

    def read_id(self, txt_name="aligned.txt"):
        with open(os.path.join(self.path_root, "info", txt_name), "r") as f:
            return [l.strip() for l in f.readlines() if l.strip()]

    def read_extrinsic(self, txt_name="final_extrinsics.txt"):
        with open(os.path.join(self.path_root, "info", txt_name), "r") as f:
            lines = [l.replace(" \n", "\n") for l in f.readlines()]
        ext = np.loadtxt(lines[1:], delimiter=" ").reshape(-1, 4, 4)             # first line is a header (dataset.py:409)
        mvps, cams = [], []
        for E in ext:
            mvp, cam = cameras.cube_mvps(E.astype(np.float32))
            mvps.append(mvp)
            cams.append(cam)
        return mvps, cams

    def read_images(self, ids):
        items = []
        for i in ids:
            p = os.path.join(self.path_root, "cube", "%s.npz" % i)
            if not os.path.exists(p):
                items.append(None)
                continue
            z = np.load(p)
            c = z["color"].shape[1]
            if c != self.cube_res:
                raise ValueError("%s holds %d^2 cube faces but train.pano_img_res implies %d^2" % (p, c, self.cube_res))
            items.append({"color": torch.from_numpy(z["color"]) * (2 ** self.hdr_exposure), "mask": torch.from_numpy(z["mask"]),
                          "segs": torch.from_numpy(z["segs"])})
        return items


def _read_pano_rgba(path):
    """cv2.imread(path, -1) of the (4-channel) panorama; the reference's file is named .jpg, so decode by content"""
    with open(path, "rb") as f:
        magic = f.read(8)
    if magic.startswith(b"\x89PNG"):
        return IO.read_png(path)
    try:
        from PIL import Image
    except ImportError as e:
        raise ValueError("%s is not a PNG: Pillow is needed to decode it" % path) from e
    return np.asarray(Image.open(path))


class ImageCubeDerived(SynCubeDataset):
    """datasets/dataset.py:352-549 on the reference's raw layout:
        <root>/info/aligned.txt, final_extrinsics.txt
        <root>/derived/<id>/panoImage_orig.jpg (RGBA, alpha = validity), panoImage_gray.png (class ids)
        <root>/hdr/<id>/ccm.hdr (Radiance RGBE)
    Per view: colour * 2^exposure, alpha eroded 5x5 and /255, class ids nearest-resized to the panorama, Sobel gradient magnitude
    of the grey image -- all warped to six cube faces with Pano2Cube (nearest).  Falls back to the cube/<id>.npz files of
    write_synthetic_dataset() when the raw directories are absent."""

    pano_hw = (4000, 8000)               # Pano2Cube(1, 4000, 8000, ...): only stored, the warp grid is resolution-free (:361)

    def _raw_path(self, i):
        return os.path.join(self.path_root, "derived", i, "panoImage_orig.jpg")

    def read_images(self, ids):
        if not ids or not os.path.exists(self._raw_path(ids[0])):
            return super().read_images(ids)
        from . import imgops as cv
        from .pano2cube import Pano2Cube
        self.pano2cube = Pano2Cube(1, self.pano_hw[0], self.pano_hw[1], self.cube_res, 6)
        items = []
        for i in ids:
            rgba = _read_pano_rgba(self._raw_path(i))
            h, w = rgba.shape[:2]
            mask = rgba[:, :, 3:4]
            color = IO.read_hdr(os.path.join(self.path_root, "hdr", i, "ccm.hdr"))              # RGB
            color = np.clip(color, 0.0, np.finfo(np.float32).max) * np.float32(2 ** self.hdr_exposure)
            gx, gy = cv.sobel3(cv.gray(color, "RGB"))
            rgb_grad = cv.magnitude(gx, gy)
            mask = cv.erode(mask, 5).astype(np.float32) / 255.0                                     # [h, w]
            segs = IO.read_png(os.path.join(self.path_root, "derived", i, "panoImage_gray.png"))
            if segs.ndim == 3:
                segs = segs[..., 0]
            segs = cv.resize_nearest(segs, (w, h)).astype(np.float32)
            img = np.concatenate([color, mask[..., None], segs[..., None], rgb_grad[..., None]], axis=-1)
            c = img.shape[-1]
            cube = self.pano2cube.Tocube(torch.from_numpy(img).permute(2, 0, 1).reshape(1, c, h, w), mode="nearest")
            faces = cube[0].reshape(6, c, self.cube_res, self.cube_res).permute(0, 2, 3, 1)
            item = {"color": faces[..., 0:3].contiguous(), "mask": faces[..., 3:4].contiguous(), "segs": faces[..., 4:5].contiguous(),
                    "rgb_grad": faces[..., 5:6].contiguous()}
            item.update(self._extra(i, segs))
            items.append(item)
        return items

    def _extra(self, i, segs):
        return {}

    def __getitem__(self, index):
        it = super().__getitem__(index)
        for k in ("rgb_grad", "gt_albedo", "gt_roughness"):
            if k in self.images_items[index]:
                it[k] = self.images_items[index][k]
        return it


class ImageCubeSyn(ImageCubeDerived):
    """datasets/dataset.py:669-893: as above plus ground-truth albedo.png / roughness.png panoramas (resized to 4c x 2c, albedo
    sRGB -> linear) and the novel-view list (info/novel.txt, info/novel_extrinsics.txt) when present"""

    pano_hw = (512, 1024)

    def __init__(self, path_mesh, resolution=[1000, 2000], hdr_exposure=1.0):
        super().__init__(path_mesh, resolution, hdr_exposure)
        self.novel_ids, self.novel_extrinsics_list, self.novel_cam_position_list, self.novel_images_items = [], [], [], []
        if os.path.exists(os.path.join(self.path_root, "info", "novel.txt")):
            self.novel_ids = self.read_id("novel.txt")
            self.novel_extrinsics_list, self.novel_cam_position_list = self.read_extrinsic("novel_extrinsics.txt")
            self.novel_images_items = self.read_images(self.novel_ids)

    def _extra(self, i, segs):
        from . import imgops as cv
        size = (self.cube_res * 4, self.cube_res * 2)
        out = {"segs_pano": torch.from_numpy(cv.resize_nearest(segs, size)[..., None])}
        pa = os.path.join(self.path_root, "derived", i, "albedo.png")
        if os.path.exists(pa):
            alb = cv.resize_linear(IO.read_png(pa)[..., :3], size).astype(np.float32) / 255.0
            out["gt_albedo"] = torch.from_numpy(alb ** np.float32(2.2))
        pr = os.path.join(self.path_root, "derived", i, "roughness.png")
        if os.path.exists(pr):
            rough = IO.read_png(pr)
            if rough.ndim == 3 and rough.shape[2] == 1:
                rough = rough[..., 0]                        # cv2.imread(-1) returns [h, w] for a single-channel file
            out["gt_roughness"] = torch.from_numpy(cv.resize_linear(rough, size).astype(np.float32) / 255.0)
        return out


class ImageCubeNovel(Dataset):
    """datasets/dataset.py:552-667: no images, a fly-through of 60 cameras -- camera 2 of final_extrinsics.txt moved by (-0.2, 0, -0.6)
    and then along +x in steps of 6/60 -- each as six cube-face mvps built like ImageCubeDerived's (tester --teststage View)"""

    start_index, offset, direction, num, length = 2, (-0.2, 0.0, -0.6), (1.0, 0.0, 0.0), 60, 6.0

    def __init__(self, path_mesh, resolution=[1000, 2000], hdr_exposure=5.0):
        super().__init__()
        self.path_mesh, self.path_root = path_mesh, find_root(path_mesh)
        self.resolution, self.cube_res, self.hdr_exposure = resolution, int(resolution[1] / 4), hdr_exposure
        self.ids = []
        self.extrinsics_list, self.cam_position_list = self.read_extrinsic()

    def __len__(self):
        return len(self.extrinsics_list)

    def __getitem__(self, index):
        return {"cam_to_world": self.extrinsics_list[index], "cam_position": self.cam_position_list[index]}

    def read_extrinsic(self):
        with open(os.path.join(self.path_root, "info", "final_extrinsics.txt"), "r") as f:
            lines = [l.replace(" \n", "\n") for l in f.readlines()]
        ext = np.loadtxt(lines[1:], delimiter=" ").reshape(-1, 4, 4)
        start = ext[min(self.start_index, ext.shape[0] - 1)].astype(np.float32).copy()
        start[0:3, 3] += np.asarray(self.offset, np.float32)
        mvps, cams = [], []
        for i in range(self.num):
            E = start.copy()
            E[0:3, 3] += np.asarray(self.direction, np.float32) * np.float32(self.length / self.num * i)
            mvp, cam = cameras.cube_mvps(E)
            mvps.append(mvp)
            cams.append(cam)
        return mvps, cams


def parse_roomseg(path):
    """utils/general.py:115-125"""
    with open(os.path.join(path, "originOccupancyGrid_f0.meta"), "r") as f:
        s, w, h, xmin, zmin = f.readline().strip().split(" ")
    img = IO.read_png(os.path.join(path, "roomSegs_uchar_f0.png")).astype(np.float32)
    if img.shape[2] == 1:
        img = np.repeat(img, 3, axis=2)
    # the reference reads the file with cv2.imread (3 channels, BGR order) and keeps channel 0, i.e. BLUE (utils/general.py:121-123)
    room = torch.from_numpy(np.ascontiguousarray(img[:, :, 2:3])).unsqueeze(0).permute(0, 3, 1, 2)
    return float(s), float(w), float(h), float(xmin), float(zmin), room


def write_synthetic_dataset(root, T=20000, texel_res=256, tex_res=256, n_side=2, seed=666, style="room", compress=True):
    """mesh + radiance texture + index texture + exact texel G-buffer + cameras (no GT images: see render_gt_views).
    compress=False stores the texel G-buffer un-deflated (4096^2: 400 MB that np.load maps back in a fraction of a second)"""
    from . import synth
    sc = synth.make_scene(T, seed=seed, tex_res=tex_res, style=style)
    d = os.path.join(root, "vrproc", "hdr_texture")
    os.makedirs(d, exist_ok=True)
    os.makedirs(os.path.join(root, "info"), exist_ok=True)
    os.makedirs(os.path.join(root, "cube"), exist_ok=True)
    os.makedirs(os.path.join(root, "roomseg"), exist_ok=True)
    IO.write_obj(os.path.join(d, "out1.obj"), sc["verts"], sc["tris"], sc["tri_uvs"])
    IO.write_hdr(os.path.join(d, "hdr_texture.hdr"), sc["hdr"][::-1])             # file orientation = un-flipped
    pos, nrm, valid = synth.make_texel_gbuffer(sc, texel_res)
    (np.savez_compressed if compress else np.savez)(os.path.join(d, "texel_gbuffer.npz"), position=pos[::-1].copy(), normal=nrm[::-1].copy())
    idx = np.zeros((texel_res, texel_res, 3), np.uint16)
    idx[valid[::-1] > 0] = (1, 1, 0)                                               # non-zero code: not a seam; codes unused with texel_gbuffer
    IO.write_png(os.path.join(d, "0.png"), idx[..., ::-1])                        # cv2 stores BGR
    # class-id texture of the evaluation model (models/test_nvdiffrast.py:79-81): the chart's class in every texel of its rect (+ gutter)
    seg = np.zeros((texel_res, texel_res), np.uint8)
    for p in sc["patches"]:
        x, y, w, h = p.rect
        c0, c1 = max(0, int(np.floor((x - synth.GUTTER / 2) * texel_res))), min(texel_res, int(np.ceil((x + w + synth.GUTTER / 2) * texel_res)))
        r0, r1 = max(0, int(np.floor((y - synth.GUTTER / 2) * texel_res))), min(texel_res, int(np.ceil((y + h + synth.GUTTER / 2) * texel_res)))
        seg[r0:r1, c0:c1] = p.cls
    IO.write_png(os.path.join(d, "0_seg_gray.png"), np.ascontiguousarray(seg[::-1]))
    cams = cameras.grid_cameras(n_side, room=synth.HOUSE) if style == "house" else cameras.grid_cameras(n_side)      # (house: the grid spans all 3 x 3 rooms)
    with open(os.path.join(root, "info", "aligned.txt"), "w") as f:
        f.write("\n".join("view%03d" % i for i in range(len(cams))) + "\n")
    with open(os.path.join(root, "info", "final_extrinsics.txt"), "w") as f:
        f.write("%d\n" % len(cams))
        for E in cams:
            for r in E:
                f.write(" ".join("%.9g" % x for x in r) + "\n")
    with open(os.path.join(root, "roomseg", "originOccupancyGrid_f0.meta"), "w") as f:
        f.write("0.05 200 200 -1 -1\n")
    IO.write_png(os.path.join(root, "roomseg", "roomSegs_uchar_f0.png"), np.ones((200, 200, 3), np.uint8))
    return sc


def write_index_texture_from_panoramas(root, conf_irt, chunk=1 << 21):
    """Replace the placeholder codes of `0.png` by real ones (needs the GPU; asset preparation, not a stage): what the reference's asset pipeline stores per texel
    is WHERE one of the scene's panoramas sees it -- (row code, column code, panorama id), models/tracer_o3d_irt.py:119-135 -- and TracerO3d gathers the texel's
    position and normal from that panorama's G-buffer.  Here the panoramas are the product's own (TracerO3d.generate_positions), every valid texel of the exact
    texel G-buffer is projected into all of them (equirectangular: theta = atan2(x, z), phi = asin(y), utils/Cube2Pano.py:57-70) and assigned to the panorama whose
    stored position at that pixel lies nearest to the texel (i.e. sees it un-occluded, if any does).  Returns the share of texels whose gathered position is
    within 5 cm of their true one."""
    from .conf import ConfigFactory
    from .models import TracerO3d
    mesh_dir = os.path.join(root, "vrproc", "hdr_texture")
    conf = ConfigFactory.parse_file(conf_irt)
    ds = SynCubeDataset(conf.get_string("train.path_mesh_open3d"), conf.get_list("train.pano_img_res"), conf.get_float("train.hdr_exposure"))
    model = TracerO3d(conf, ds.ids, ds.extrinsics_list)
    model.generate_positions()
    panos = torch.stack([p_[..., 0:3] for p_ in model.position_normal_list], 0)            # [K, h, w, 3]
    K, h, w, _ = panos.shape
    cams = torch.stack([c_.to(panos.device).reshape(3) for c_ in ds.cam_position_list], 0)
    z = np.load(os.path.join(mesh_dir, "texel_gbuffer.npz"))
    pos = torch.from_numpy(z["position"]).to(panos.device)
    H, W, _ = pos.shape
    valid = torch.from_numpy((np.abs(z["normal"]).sum(-1) > 0)).to(panos.device).reshape(-1)
    pos = pos.reshape(-1, 3)
    codes = torch.zeros((H * W, 3), dtype=torch.int32, device=panos.device)
    near = torch.zeros(H * W, dtype=torch.bool, device=panos.device)
    idx_all = torch.nonzero(valid)[:, 0]

    def pixel_of(d, sg):
        theta, phi = torch.atan2(sg[0] * d[:, 0], sg[2] * d[:, 2]), torch.asin((sg[1] * d[:, 1]).clamp(-1, 1))
        col = ((theta + np.pi) / (2 * np.pi) * (w - 1)).round().long().clamp(0, w - 1)
        row = ((0.5 * np.pi - phi) / np.pi * (h - 1)).round().long().clamp(0, h - 1)
        return row, col

    # the panorama's axis conventions are whatever the cube faces + Cube2Pano produce: measured, not assumed -- the sign triple under which the panoramas show
    # the texels where they are
    sub = idx_all[:: max(1, idx_all.numel() // 100000)]
    score = {}
    for sg in [(a_, b_, c_) for a_ in (1.0, -1.0) for b_ in (1.0, -1.0) for c_ in (1.0, -1.0)]:
        p_ = pos[sub]
        best = torch.full((sub.numel(),), float("inf"), device=panos.device)
        for k in range(K):
            d = p_ - cams[k]
            row, col = pixel_of(d / d.norm(dim=-1, keepdim=True).clamp(min=1e-12), sg)
            best = torch.minimum(best, (panos[k, row, col] - p_).norm(dim=-1))
        score[sg] = float((best < 0.05).float().mean().item())
    sg = max(score, key=score.get)
    if score[sg] < 0.2:
        raise RuntimeError("no axis convention maps the texels into the panoramas (best %s: %.3f)" % (sg, score[sg]))
    for a in range(0, idx_all.numel(), chunk):
        ids = idx_all[a:a + chunk]
        p_ = pos[ids]
        best = torch.full((ids.numel(),), float("inf"), device=panos.device)
        best_code = torch.zeros((ids.numel(), 3), dtype=torch.int32, device=panos.device)
        for k in range(K):
            d = p_ - cams[k]
            row, col = pixel_of(d / d.norm(dim=-1, keepdim=True).clamp(min=1e-12), sg)
            err = (panos[k, row, col] - p_).norm(dim=-1)
            better = err < best
            best = torch.where(better, err, best)
            # the reference decodes int(code / 50000 * size): store the code of the pixel's centre, never (0, 0, 0) -- that is a seam
            rc = ((row.float() + 0.5) / h * 50000).round().clamp(1, 50000).int()
            cc = ((col.float() + 0.5) / w * 50000).round().clamp(1, 50000).int()
            kk = torch.full_like(rc, k)
            best_code = torch.where(better[:, None], torch.stack([rc, cc, kk], -1), best_code)
        codes[ids] = best_code
        near[ids] = best < 0.05
    idx = codes.reshape(H, W, 3).cpu().numpy().astype(np.uint16)
    IO.write_png(os.path.join(mesh_dir, "0.png"), np.ascontiguousarray(idx[..., ::-1]))        # the file stores RGB = (panorama id, column code, row code)
    return float(near[idx_all].float().mean().item())


def write_conf(path, root, cube_res=32, spp=(64, 16), albedo_res=256, rough_res=256, epochs=1, model="mat"):
    txt = """train{
    expname = synthetic
    dataset_class = datasets.dataset.ImageCubeSyn
    model_class = %s
    irf_loss_class = models.loss.RenderLoss
    plot_freq = 1000
    ckpt_freq = 1000
    mat_epoch = %d
    mat_learning_rate = 3e-2
    mat_sched_step = 20
    mat_sched_factor = 0.8
    optim_cam = False
    pano_img_res = [%d,%d]
    sample_light = [%d, %d]
    hdr_exposure = 0
    env_res = [8,16]
    batch_size = 1
    albedo_res = %d
    roughness_res = %d
    irt_res = native
    path_mesh_open3d = %s
}
test{
    expname = synthetic
    dataset_class = datasets.dataset.ImageCubeSyn
    model_class = models.test_nvdiffrast.MaterialModel
    irf_loss_class = models.loss.RenderLoss
    pano_img_res = [%d,%d]
    sample_light = [%d, %d]
    hdr_exposure = 0
    path_mesh_open3d = %s
}
render_loss
{
    loss_type = L1
    w_gradient = 1
}
models{
    render{
        sample_type = [ uniform, importance]
    }
}
""" % ("models.mat_nvdiffrast.MaterialModel" if model == "mat" else "models.tracer_o3d_irt.TracerO3d", epochs, cube_res * 2, cube_res * 4,
       spp[0], spp[1], albedo_res, rough_res, os.path.join(root, "vrproc", "hdr_texture", "out1.obj"),
       cube_res * 2, cube_res * 4, spp[0], max(spp[1], 32), os.path.join(root, "vrproc", "hdr_texture", "out1.obj"))
    with open(path, "w") as f:
        f.write(txt)


def render_gt_views(root, conf, sc, albedo_res, rough_res, seed=666):
    """GT cube images rendered by our own forward from ground-truth materials; seg ids = chart class"""
    from . import synth
    from .models import MaterialModel
    ds = SynCubeDataset(conf.get_string("train.path_mesh_open3d"), conf.get_list("train.pano_img_res"), conf.get_float("train.hdr_exposure"))
    model = MaterialModel(conf, ds.ids, ds.extrinsics_list).cuda()
    alb, rgh = synth.make_gt_materials(sc, albedo_res, rough_res, seed)
    with torch.no_grad():
        model.materials_a.copy_(torch.from_numpy(alb))
        model.materials_r.copy_(torch.from_numpy(rgh))
        tri_class = torch.from_numpy(sc["tri_class"]).cuda()
        for i, vid in enumerate(ds.ids):
            res = model(ds.extrinsics_list[i], vid, ds.cam_position_list[i], 2)
            tri = model._gbuffer(ds.extrinsics_list[i], vid)["tri_id"].long()
            segs = torch.where(tri > 0, tri_class[(tri - 1).clamp(min=0)].long(), torch.zeros_like(tri)).float().unsqueeze(-1)
            np.savez_compressed(os.path.join(root, "cube", "%s.npz" % vid), color=res["rgb"].cpu().numpy(),
                                mask=res["empty_mask"].cpu().numpy(), segs=segs.cpu().numpy())
    return alb, rgh


# ---------------------------------------------------------------------------------------------------------------------
# NIrF datasets
# ---------------------------------------------------------------------------------------------------------------------
def _vertex_normals(vertices, indices):
    """area-weighted vertex normals (Open3D compute_vertex_normals: sum of un-normalised face normals, then normalise)"""
    v = vertices[indices]
    fn = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    vn = np.zeros_like(vertices, dtype=np.float64)
    for k in range(3):
        np.add.at(vn, indices[:, k], fn)
    ln = np.linalg.norm(vn, axis=-1, keepdims=True)
    return (vn / np.maximum(ln, 1e-20)).astype(np.float32), fn


class MeshPoint(Dataset):
    """datasets/dataset.py:39-92: `num_sample` points drawn uniformly (by area) on the mesh with interpolated vertex normals,
    moved `delta` along the normal; change_points() redraws them (called once per epoch, train_irrf.py:237).
    Open3D's sample_points_uniformly uses its own RNG; here numpy's global generator (the runners seed it)."""

    def __init__(self, path_mesh, num_sample, delta=1e-2):
        super().__init__()
        self.path_mesh, self.num_sample, self.delta = path_mesh, int(num_sample), delta
        obj = IO.load_obj(path_mesh)
        self.vertices, self.indices = obj["vertices"].astype(np.float32), obj["indices"]
        self.vertex_normals, fn = _vertex_normals(self.vertices.astype(np.float64), self.indices)
        area = 0.5 * np.linalg.norm(fn, axis=-1)
        self.cdf = np.cumsum(area / area.sum())
        self.AABB = np.stack([self.vertices.min(0), self.vertices.max(0)], axis=0)
        self.points = self.normals = None

    def __len__(self):
        return self.num_sample

    def sample_mesh(self):
        t = np.minimum(np.searchsorted(self.cdf, np.random.rand(self.num_sample)), len(self.indices) - 1)
        r1, r2 = np.sqrt(np.random.rand(self.num_sample, 1)), np.random.rand(self.num_sample, 1)
        w = np.concatenate([1 - r1, r1 * (1 - r2), r1 * r2], axis=-1)            # uniform barycentrics
        tri = self.indices[t]
        p = (self.vertices[tri] * w[..., None]).sum(1)
        n = (self.vertex_normals[tri] * w[..., None]).sum(1)
        n = n / np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-20)
        return (p + n * self.delta).astype(np.float32), n.astype(np.float32)

    def change_points(self):
        self.points, self.normals = self.sample_mesh()

    def get_AABB(self):
        return self.AABB

    def __getitem__(self, index):
        return {"point": torch.from_numpy(self.points[index]), "normal": torch.from_numpy(self.normals[index])}


class ImageMeshPoint(Dataset):
    """datasets/dataset.py:96-260: the validation set is the G-buffer (position + delta*normal, normal) of ONE equirectangular
    view, one sample per pixel.  The reference rasterises it with pyredner from a hard-coded capture of its private scene;
    here the view is the first row of `<root>/cameras.txt` when present, else the AABB centre, and the G-buffer is ray cast
    (texir_trace_shade hit records) when arrange_buffers() is first called."""

    def __init__(self, path_mesh, resolution=(128, 256), delta=1e-2, scene=None, eye=None):
        super().__init__()
        self.path_mesh, self.resolution, self.delta = path_mesh, [int(resolution[0]), int(resolution[1])], delta
        self.scene, self.eye = scene, eye
        self.points = self.normals = None

    def __len__(self):
        return self.resolution[0] * self.resolution[1]

    def arrange_buffers(self):
        if self.points is not None:
            return
        from .scene import Scene
        obj = IO.load_obj(self.path_mesh)
        if self.scene is None:
            self.scene = Scene(obj["vertices"], obj["indices"], IO.triangle_uvs_open3d(obj), np.zeros((2, 2, 3), np.float32))
        if self.eye is None:
            self.eye = 0.5 * (obj["vertices"].min(0) + obj["vertices"].max(0))
        h, w = self.resolution
        # equirectangular directions, y up: row 0 = zenith, column 0 = -x ... (utils/Cube2Pano.py convention)
        theta = (torch.arange(h, dtype=torch.float32) + 0.5) / h * np.pi
        phi = (torch.arange(w, dtype=torch.float32) + 0.5) / w * 2 * np.pi - np.pi
        theta, phi = torch.meshgrid(theta, phi, indexing="ij")
        d = torch.stack([torch.sin(theta) * torch.sin(phi), torch.cos(theta), -torch.sin(theta) * torch.cos(phi)], -1).reshape(-1, 3)
        o = torch.from_numpy(np.asarray(self.eye, np.float32)).expand_as(d).contiguous()
        _, t, pid, _ = self.scene.trace_shade(o, d, return_hits=True)
        t, pid = t.cpu(), pid.cpu().long()
        hit = torch.isfinite(t) & (pid >= 0)
        p = o + d * torch.where(hit, t, torch.zeros_like(t))[:, None]
        v = torch.from_numpy(obj["vertices"].astype(np.float32))[torch.from_numpy(obj["indices"].astype(np.int64))[pid.clamp(min=0)]]
        n = torch.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0], dim=-1)
        n = n / n.norm(dim=-1, keepdim=True).clamp(min=1e-20)
        n = torch.where(((n * d).sum(-1, keepdim=True) > 0), -n, n)                  # face the viewer
        n = torch.where(hit[:, None], n, torch.zeros_like(n))
        self.points, self.normals = (p + self.delta * n).numpy(), n.numpy()

    def __getitem__(self, index):
        return {"point": torch.from_numpy(self.points[index]), "normal": torch.from_numpy(self.normals[index])}
