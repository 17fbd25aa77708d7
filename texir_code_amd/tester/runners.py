"""The four evaluation runners of tester/ (tester/exp_runner.py:42-47): material editing, novel views, relighting and the
re-rendering error, re-using the trained material textures of a `Mat-<expname>` run.  Each renders the views with the evaluation
model (tester/test_model.py), converts the six cube faces to a panorama (Cube2Pano.ToPano) and writes the image the reference's
plot_mat (utils/plots.py:69-85) writes: `<name>_0.hdr` (Radiance RGBE) or, tone-mapped, `<name>_0.png`.
tensorboard / scatter plots are debug output and are not reproduced."""
import os
import sys

import numpy as np
import torch

from .. import io_formats as IO, metrics as M
from ..conf import ConfigFactory
from ..cube2pano import Cube2Pano
from ..plugin import get_class


def change_color(frame_per_color, colors, channel=3):
    """tester/test_editing.py:165-199: piecewise-linear key-frame interpolation (the first key, then frame_per_color frames per segment)"""
    out = [np.asarray(colors[0], np.float32)]
    for i in range(1, len(colors)):
        t = np.linspace(0, 1, frame_per_color)[:, None]
        seg = (1 - t) * np.asarray(colors[i - 1], np.float64)[None, :channel] + t * np.asarray(colors[i], np.float64)[None, :channel]
        out += [s.astype(np.float32) for s in seg]
    return out


class _EvalRunner:
    subdir = "eval"
    relighting = False

    def __init__(self, **kwargs):
        torch.set_default_dtype(torch.float32)
        torch.set_num_threads(1)
        self.conf = ConfigFactory.parse_file(kwargs["conf"])
        self.exps_folder_name = kwargs["exps_folder_name"]
        self.expname = "Mat-" + kwargs["expname"]
        self.expdir = os.path.join("../", self.exps_folder_name, self.expname)
        timestamp = kwargs.get("timestamp", "latest")
        if timestamp == "latest":
            if not os.path.isdir(self.expdir) or not os.listdir(self.expdir):
                raise FileNotFoundError("no training run under %s: run --trainstage Mat first" % self.expdir)
            timestamp = sorted(os.listdir(self.expdir))[-1]
        self.timestamp = timestamp
        self.plots_dir = os.path.join(self.expdir, self.timestamp, "plots")
        self.editing_dir = os.path.join(self.plots_dir, self.subdir)
        os.makedirs(self.editing_dir, exist_ok=True)
        torch.manual_seed(666)
        torch.cuda.manual_seed(666)
        np.random.seed(666)
        print("shell command : {0}".format(" ".join(sys.argv)))
        print("Loading data ...")
        self.train_dataset = get_class(self.conf.get_string("test.dataset_class"))(
            self.conf.get_string("test.path_mesh_open3d"), self.conf.get_list("test.pano_img_res"), self.conf.get_float("test.hdr_exposure"))
        print("Finish loading data ...")
        self.model = get_class(self.conf.get_string("test.model_class"))(
            conf=self.conf, cam_position_list=self.train_dataset.cam_position_list, checkpoint_material=self.plots_dir, relighting=self.relighting)
        self.model.cuda()
        self.model.eval()
        self.pano_res = self.conf.get_list("test.pano_img_res")
        self.cube_lenth = int(self.pano_res[1] / 4)
        self.cube2pano = Cube2Pano(pano_width=self.pano_res[1], pano_height=self.pano_res[0], cube_lenth=self.cube_lenth)
        self.outputs = []

    def to_pano(self, cube):
        """[6,c,c,k] -> [pano_h, pano_w, k] (the runners' permute / reshape / ToPano idiom)"""
        c = self.cube_lenth
        x = cube.detach().cpu().permute(0, 3, 1, 2).reshape(1, -1, c, c)
        return self.cube2pano.ToPano(x)[0].permute(1, 2, 0)

    def plot_mat(self, img, name, tonemapped):
        """utils/plots.py:69-85 with iters = 0"""
        a = img.numpy()
        if tonemapped:
            path = "{0}/{1}_0.png".format(self.editing_dir, name)
            IO.write_png(path, (np.clip(np.clip(a, 0, None) ** (1 / 2.2), 0.0, 1.0) * 255).astype(np.uint8))
        else:
            path = "{0}/{1}_0.hdr".format(self.editing_dir, name)
            IO.write_hdr(path, np.ascontiguousarray(a, np.float32))
        print("saving render img to {0}".format(path))
        self.outputs.append(path)

    def view(self, i):
        d = self.train_dataset
        return d.extrinsics_list[i], d.ids[i] if i < len(d.ids) else 0, d.cam_position_list[i].cuda()


class MatEditingRunner(_EvalRunner):
    """tester/test_editing.py: run() = plot_to_disk_varying (:236-316): view 0 re-rendered along a key-framed sequence of floor / wall
    albedo edits, then of floor roughness edits"""
    subdir = "editing-varying"
    albedo_floors = [[0.56, 0.93, 0.56], [0.52, 0.00, 0.08], [0.12, 0.00, 0.58], [0.12, 0.50, 0.08], [0.56, 0.93, 0.56]]
    albedo_walls = [[0.48, 0.63, 0.73], [0.1, 0.00, 0.63], [0.12, 0.60, 0.0], [0.81, 0.60, 0.08], [0.48, 0.63, 0.73]]
    roughness_floors = [[0.01], [0.2], [0.4], [0.6], [0.8]]
    frame_per_color = 5

    def plot_to_disk_cube(self):
        with torch.no_grad():
            for i in range(len(self.train_dataset.ids)):
                mvp, vid, cam = self.view(i)
                self.plot_mat(self.to_pano(self.model(mvp, vid, cam, True)["rgb"]), "editing_{}".format(i), False)

    def plot_to_disk_varying(self):
        floors = change_color(self.frame_per_color, self.albedo_floors)
        walls = change_color(self.frame_per_color, self.albedo_walls)
        roughs = change_color(self.frame_per_color, self.roughness_floors, 1)
        mvp, vid, cam = self.view(0)
        with torch.no_grad():
            for j in range(len(floors) + len(roughs)):
                if j < len(floors):
                    res = self.model(mvp, vid, cam, True, floors[j], walls[j])
                else:
                    res = self.model(mvp, vid, cam, True, None, None, roughs[j - len(floors)])
                self.plot_mat(self.to_pano(res["rgb"]), "editing_{}".format(j), True)

    def run(self):
        print("testing...")
        self.plot_to_disk_varying()


class NovelViewRunner(_EvalRunner):
    """tester/test_novel.py:170-190: every camera of the (novel-view) dataset rendered with the trained materials"""
    subdir = "novel_view"

    def run(self):
        print("testing...")
        with torch.no_grad():
            for i in range(len(self.train_dataset.extrinsics_list)):
                mvp, cam = self.train_dataset.extrinsics_list[i], self.train_dataset.cam_position_list[i].cuda()
                self.plot_mat(self.to_pano(self.model(mvp, 0, cam, True)["rgb"]), "novel_view_{}".format(i), False)


class RelightingRunner(_EvalRunner):
    """tester/test_relighting.py:198-216: the scene re-lit by recoloured light sources, diffuse term traced (relighting=True)"""
    subdir = "relighting"
    relighting = True

    def run(self):
        print("testing...")
        with torch.no_grad():
            for i in range(len(self.train_dataset.ids)):
                mvp, vid, cam = self.view(i)
                self.plot_mat(self.to_pano(self.model(mvp, vid, cam)["rgb"]), "relighting_{}".format(i), False)


class MatErrorRunner(_EvalRunner):
    """tester/test_error.py:170-200: re-rendering error of every training view against its photograph: MSE, PSNR and SSIM of the
    tone-mapped panoramas (accumulated exactly as the reference does, incl. its PSNR-of-the-running-sum, :188)"""
    subdir = "error"

    def run(self):
        print("testing...")
        n = len(self.train_dataset.ids)
        mse_error = psnr_error = ssim_error = 0.0
        with torch.no_grad():
            for i in range(n):
                mvp, vid, cam = self.view(i)
                gt = self.to_pano(self.train_dataset.images_items[i]["color"])
                pred = self.to_pano(self.model(mvp, vid, cam, False)["rgb"])
                a, b = M.tonemapping(gt.unsqueeze(0)), M.tonemapping(pred.unsqueeze(0))
                ssim_error += 1.0 - (1.0 - float(M.ssim(a.permute(0, 3, 1, 2), b.permute(0, 3, 1, 2))))
                mse_error += float(torch.mean((a - b) ** 2))
                psnr_error += float(M.mse_to_psnr(torch.tensor(mse_error)))
                self.plot_mat(pred, "rendering_{}".format(i), False)
        self.metrics = {"mse": mse_error / n, "psnr": psnr_error / n, "ssim": ssim_error / n}
        print("re-rendering error: mse: {}, psnr: {}, ssim: {}".format(self.metrics["mse"], self.metrics["psnr"], self.metrics["ssim"]))
