"""IRRFTrainRunner -- drop-in for trainer/train_irrf.py:27-276 (`--trainstage IRRF`): fit the NIrF MLP to irradiance traced at
random mesh points.  Same kwargs, conf keys, checkpoint layout (ModelParameters/latest.pth) and loop order as the reference;
the GT tracing runs on the IrT kernel, the MLP on stock PyTorch-ROCm."""
import os
import time

import numpy as np
import torch

from .. import io_formats as IO
from ..nirf import hdr_recover
from ..plugin import get_class
from .base import RunnerBase


class IRRFTrainRunner(RunnerBase):
    def __init__(self, **kwargs):
        self.setup_experiment(kwargs, "IRRF", exps_root=kwargs.get("exps_root", "../"))
        self.val_batch_size = self.conf.get_int("val.batch_size")
        self.nepochs = self.conf.get_int("train.irf_epoch")
        self.is_hdr_texture = self.conf.get_bool("train.is_hdr_texture")
        is_continue, timestamp = kwargs["is_continue"], kwargs["timestamp"]
        if is_continue and timestamp == "latest":
            is_continue, timestamp = (True, self.prior_timestamps[-1]) if self.prior_timestamps else (False, None)

        print("Loading training data ...")
        path_mesh = self.conf.get_string("train.path_mesh_open3d")
        self.train_dataset = get_class(self.conf.get_string("train.dataset_class"))(path_mesh, self.conf.get_int("train.samples_point_mesh"))
        self.AABB = self.train_dataset.get_AABB()
        self.val_dataset = get_class(self.conf.get_string("val.dataset_class"))(path_mesh, self.conf.get_list("val.env_res"))
        self.train_dataloader = torch.utils.data.DataLoader(self.train_dataset, batch_size=self.train_batch_size, shuffle=True)
        self.plot_dataloader = torch.utils.data.DataLoader(self.val_dataset, batch_size=self.val_batch_size, shuffle=False)
        self.model = get_class(self.conf.get_string("train.model_class"))(conf=self.conf, AABB=self.AABB, is_hdr_texture=self.is_hdr_texture)
        self.model.cuda()
        if hasattr(self.val_dataset, "scene") and self.val_dataset.scene is None:
            self.val_dataset.scene = self.model.scene                 # one BVH for GT tracing and the validation G-buffer
        self.irf_loss = get_class(self.conf.get_string("train.irf_loss_class"))(**self.conf.get_config("irf_loss"))
        self.irf_optimizer = torch.optim.Adam(self.model.ir_radiance_network.parameters(), lr=self.conf.get_float("train.irf_learning_rate"))
        self.irf_scheduler = torch.optim.lr_scheduler.StepLR(self.irf_optimizer, self.conf.get_int("train.irf_sched_step", default=1000),
                                                             gamma=self.conf.get_float("train.irf_sched_factor", default=0.0))
        self.start_epoch = 0
        if is_continue:
            old = os.path.join(self.expdir, timestamp, "checkpoints", self.model_params_subdir, str(kwargs["checkpoint"]) + ".pth")
            print("Loading pretrained model: ", old)
            saved = torch.load(old)
            self.model.load_state_dict(saved["model_state_dict"])
            self.start_epoch = saved["epoch"]
        self.n_batches = len(self.train_dataloader)
        self.plot_freq = self.conf.get_int("train.plot_freq")
        self.ckpt_freq = self.conf.get_int("train.ckpt_freq")
        self.val_gt, self.first_val = None, True
        self.train_resolution = self.conf.get_list("train.env_res", default=[8, 16])
        self.val_resolution = self.conf.get_list("train.val_sample_res", default=[8, 16])
        self.losses = []

    def plot_to_disk(self):
        """train_irrf.py:184-231: irradiance panorama of the validation view, traced GT (first call only) next to the prediction"""
        self.model.eval()
        env_res = self.conf.get_list("val.env_res")
        self.val_dataset.arrange_buffers()
        if self.first_val:
            self.val_gt = torch.zeros(env_res[0] * env_res[1], 3)
        pred_ir = torch.zeros(env_res[0] * env_res[1], 3)
        at = 0
        with torch.no_grad():
            for one_sample in self.plot_dataloader:
                points, normals = one_sample["point"].float().cuda(), one_sample["normal"].float().cuda()
                res = self.model(points, normals, self.val_resolution, not self.first_val)
                b = points.shape[0]
                if self.first_val:
                    self.val_gt[at:at + b] = res["gt"].cpu()
                pred_ir[at:at + b] = res["pred"].cpu()
                at += b
        gt = self.val_gt.reshape(env_res[0], env_res[1], 3).numpy()
        pred = hdr_recover(pred_ir.reshape(env_res[0], env_res[1], 3)).numpy()
        IO.write_hdr(os.path.join(self.plots_dir, "irf_%d.hdr" % self.cur_iter), np.concatenate([gt, pred], axis=0).astype(np.float32))
        self.model.train()
        self.first_val = False

    def run(self):
        """train_irrf.py:233-275 as hooks on the shared loop: new surface points every epoch; before a step the periodic checkpoint and the
        validation panorama (which consumes CPU random numbers on its first call: its position in the loop is part of the trajectory)"""
        print("training...")
        self.cur_iter = self.start_epoch * len(self.train_dataloader)
        t0 = [0.0]

        def before_step(epoch, data_index):
            t0[0] = time.time()
            if self.cur_iter % self.ckpt_freq == 0 and not self.cur_iter == 0:
                self.save_checkpoints(epoch)
            if self.cur_iter % self.plot_freq == 0:
                self.plot_to_disk()

        def step(one_sample):
            points, normals = one_sample["point"].float().cuda(), one_sample["normal"].float().cuda()
            radiance_loss = self.irf_loss(self.model(points, normals, self.train_resolution))
            self.irf_optimizer.zero_grad()
            radiance_loss.backward()
            self.irf_optimizer.step()
            return radiance_loss

        def after_step(epoch, data_index, radiance_loss):
            if (self.cur_iter - 1) % 50 == 0:
                print("{0} [{1}] ({2}/{3}): radiance_loss = {4}, batch cost time : {5:.4f}s".format(
                    self.expname, epoch, data_index, self.n_batches, radiance_loss.item(), time.time() - t0[0]))
                self.losses.append(radiance_loss.item())
                self.writer.add_scalar("radiance_loss", self.losses[-1], self.cur_iter - 1)       # trainer/train_irrf.py:272
            if self.cur_iter >= self.max_niters:
                self.save_checkpoints(epoch)
                return True
            return False

        ended = self.fit(self.train_dataloader, self.start_epoch, self.nepochs, step, epoch_begin=lambda epoch: self.train_dataset.change_points(),
                         before_step=before_step, after_step=after_step, epoch_end=lambda epoch: self.irf_scheduler.step())
        if not ended:
            self.save_checkpoints(self.nepochs)
        self.writer.flush()
