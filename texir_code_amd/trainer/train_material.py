"""MatTrainRunner -- the three-stage material optimisation loop of trainer/train_material.py:408-605 (and its `_syn`
twin): stage order, requires_grad toggling, optimiser + scheduler re-creation, per-step clamps and the mask
construction of plot_to_disk_cube's first-validation branch (:251-296).  Plots / tensorboard / metrics are debug
output and are not reproduced."""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

from .. import dist_util, io_formats as IO
from ..datasets import parse_roomseg
from ..models import rgb_to_intensity
from ..optim import FusedAdam
from ..plugin import get_class
from ..runlog import phases
from .base import RunnerBase

N_SEG_CLASSES = 49          # trainer/train_material.py:188


def build_masks(segs, stage_m1_rgb, room_img=None, positions=None, room_meta=None):
    """trainer/train_material.py:251-296 on one view.
    segs [6,h,w,1] (class ids), stage_m1_rgb [6,h,w,3] (stage -1 render: light-source-only specular)
    -> seg_mask [49,6,h,w,1] (one-hot, un-eroded copy), floor_max_mask [49,6,h,w,1], room_seg_mask [R,6,h,w,1]"""
    tag = torch.arange(N_SEG_CLASSES, dtype=torch.float32, device=segs.device)
    seg_mask = ((tag.reshape(-1, 1, 1, 1, 1) - segs.unsqueeze(0)) == 0).float()
    source_seg_mask = seg_mask.clone()
    floor = seg_mask[-3]                                                            # class 46 = floor
    floor = -F.max_pool2d(-floor.permute(0, 3, 1, 2), kernel_size=15, stride=1, padding=7).permute(0, 2, 3, 1)
    seg_mask[-3] = floor
    pred = stage_m1_rgb.unsqueeze(0) * seg_mask
    floor_max_mask = (rgb_to_intensity(torch.abs(pred)) > 0.).float()
    room_mask = None
    if room_img is not None:
        scale, w, h, xmin, zmin = room_meta
        u = (positions[..., 0:1] - xmin) / scale / w
        v = (positions[..., 2:3] - zmin) / scale / h
        uv = torch.cat([u * 2 - 1, v * 2 - 1], dim=-1)
        ri = F.grid_sample(room_img.to(uv.device).expand(6, -1, -1, -1), uv, mode="nearest", padding_mode="border", align_corners=False).permute(0, 2, 3, 1)
        room_mask = ((ri.unsqueeze(0) - torch.unique(ri).reshape(-1, 1, 1, 1, 1)) == 0).float()
    return source_seg_mask, floor_max_mask, room_mask


# ALIASING CONTRACT: the batch's tensors are unsqueeze(0) VIEWS of the dataset's own storage (default_collate would stack copies: 1.3 MB per step at c = 128).
# Nothing in the runners writes into a batch; code that wants to must clone first -- an in-place op on a batch would edit the dataset for every later epoch.
def collate_one_view(batch):
    """what torch's default_collate returns for a batch of ONE item -- tensors with a leading dimension of 1, strings in a list, numbers as tensors -- without
    copying the tensors"""
    if len(batch) != 1:
        return torch.utils.data.default_collate(batch)
    out = {}
    for k, v in batch[0].items():
        if torch.is_tensor(v):
            out[k] = v.unsqueeze(0)
        elif isinstance(v, (str, bytes)):
            out[k] = [v]
        else:
            out[k] = torch.utils.data.default_collate([v])
    return out


class MatTrainRunner(RunnerBase):
    def __init__(self, **kwargs):
        # --is_continue: the reference's resume path is dead code; the flags are accepted and ignored
        self.setup_experiment(kwargs, "Mat", make_dirs=int(os.environ.get("RANK", "0")) == 0 and not kwargs.get("dry_dirs", False), keep_conf_copy=True)
        self.nepochs = self.conf.get_int("train.mat_epoch")
        torch.manual_seed(666)
        torch.cuda.manual_seed(666)
        np.random.seed(666)
        print("Loading data ...")
        with phases.phase("dataset", sync=False):
            self.train_dataset = get_class(self.conf.get_string("train.dataset_class"))(
                self.conf.get_string("train.path_mesh_open3d"), self.conf.get_list("train.pano_img_res"), self.conf.get_float("train.hdr_exposure"))
        print("Finish loading data ...")
        # batch_size = 1, shuffle = True as in the reference (train_material.py:101-104: the sampler's draws are part of the CPU random stream); the batch
        # is the item itself with a leading dimension of one -- views instead of default_collate's stacked copies (1.3 MB per step at c = 128)
        self.train_dataloader = torch.utils.data.DataLoader(self.train_dataset, batch_size=1, shuffle=True, collate_fn=collate_one_view)
        with phases.phase("model_init"):
            self.model = get_class(self.conf.get_string("train.model_class"))(
                conf=self.conf, ids=self.train_dataset.ids, extrinsics=self.train_dataset.extrinsics_list, optim_cam=self.conf.get_bool("train.optim_cam"))
            self.model.cuda()
        self.model.lean_outputs = True        # the step's only consumer of the forward dict is the loss: skip what it does not read (models.py forward)
        self.mat_loss = get_class(self.conf.get_string("train.irf_loss_class"))(**self.conf.get_config("render_loss"))
        self._new_optimizer()
        self.start_epoch = 0
        self.n_batches = len(self.train_dataloader)
        self.plot_freq = self.conf.get_int("train.plot_freq")
        # new optional key (default on): forward + loss + backward of a (view, stage) pair are replayed as one hipGraph -- same kernels,
        # same CPU-generator draws, same trajectory as the eager call order (tests/test_gpu_trainer.py); a failed capture falls back
        self.use_graph = self.conf.get_bool("train.hipgraph", default=True)
        # multi-GPU: "pixel" = one view split across ranks (same trajectory as one GPU), "view" = one view per rank per step
        self.mat_shard = self.conf.get_string("train.mat_shard", default="pixel")
        self.pano_res = self.conf.get_list("train.pano_img_res")
        self.cube_lenth = int(self.pano_res[1] / 4)
        self.first_val = True
        self.floor_max_mask, self.seg_mask, self.room_seg_mask = {}, {}, {}
        rs = os.path.join(os.path.dirname(os.path.dirname(self.conf.get_string("train.path_mesh_open3d"))), "roomseg")
        rs = rs if os.path.isdir(rs) else os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(self.conf.get_string("train.path_mesh_open3d")))), "roomseg")
        (self.room_meta_scale, self.room_meta_w, self.room_meta_h, self.room_meta_xmin, self.room_meta_zmin, self.room_img) = parse_roomseg(rs)
        self.cur_iter = 0
        self.log = []
        # new optional key: with train.log_lag = n the loss values of a step are copied to pinned host memory asynchronously and logged / printed n steps
        # later, so that the recorded steps are queued back to back; same values, same lines, same order (the last n lines of a stage appear when the stage ends).
        # Default 1 (round 6): the step's line is printed while the NEXT step runs -- the GPU no longer idles through the host's report of every step;
        # 0 = the reference's own timing: `.item()` + print right after each step, one host synchronisation per step (train_material.py:459-468)
        self.log_lag = max(0, self.conf.get_int("train.log_lag", default=1))

    def _view_inputs(self, gt_item, vid0):
        """device-resident, long-lived inputs of a view (what a recorded step reads)"""
        self._gs_inputs = getattr(self, "_gs_inputs", {})
        if vid0 not in self._gs_inputs:
            gt = gt_item["color"].float().cuda()
            h, w, c = gt.shape[-3:]
            mvp = gt_item["cam_to_world"].float()
            cam = gt_item["cam_position"].float().cuda()
            self._gs_inputs[vid0] = (mvp[0] if mvp.dim() == 4 else mvp, (cam[0] if cam.dim() == 2 else cam).contiguous(),
                                     gt.reshape(-1, h, w, c).contiguous(), gt_item["mask"].float().cuda().reshape(-1, h, w, 1).contiguous())
        return self._gs_inputs[vid0]

    def _sharded_step(self, gt_item, stage):
        """several ranks, train.mat_shard = pixel, stages 1 / 2: the view's pixels are split across the ranks for the specular trace and its backward,
        the texture side is replicated (sharded_step.ShardedMatStep): the single-GPU trajectory bit for bit, two small all_gathers per step"""
        from ..sharded_step import ShardedMatStep
        vid = gt_item["id"]
        vid0 = vid[0] if isinstance(vid, (list, tuple)) else vid
        if getattr(self, "_ss", None) is None or self._ss.opt is not self.mat_optimizer:
            self.mat_loss.lazy_item = True
            self.mat_loss.unit_upstream = True
            self._ss = ShardedMatStep(self.model, self.mat_loss, self.mat_optimizer, [self.model.materials_a, self.model.materials_r],
                                      use_graph=getattr(self, "use_graph", False))
        if (vid0, stage) not in self._ss.views:
            mvp, cam, gt, gmask = self._view_inputs(gt_item, vid0)
            self._ss.capture(vid0, mvp, cam, gt, gmask, self.seg_mask[str(vid0)], self.floor_max_mask[str(vid0)],
                             self.room_seg_mask[str(vid0)] if stage == 2 else None, stage)
        self._ss.step(vid0, stage)
        out = self._ss.views[(vid0, stage)]["out"]
        return out[0], out[1]

    def _graph_step(self, gt_item, stage, replicated=False):
        """train.hipgraph = true: forward + loss + backward of a (view, stage) pair replayed as one hipGraph (graph_step.py).
        replicated: several ranks run the IDENTICAL full step (stage 0 of the pixel-sharded mode: no specular term, nothing to split): the optimiser
        step stays inside the graph and nothing is reduced"""
        from ..graph_step import GraphedMatStep
        vid = gt_item["id"]
        vid0 = vid[0] if isinstance(vid, (list, tuple)) else vid
        if getattr(self, "_gs", None) is None or self._gs.opt is not self.mat_optimizer:
            self.mat_loss.lazy_item = True
            self.mat_loss.unit_upstream = True        # train_step back-propagates the loss itself (loss.backward())
            if replicated:
                for p in (self.model.materials_a, self.model.materials_r):
                    p._texir_replicated_grads = True
            self._gs = GraphedMatStep(self.model, self.mat_loss, self.mat_optimizer, [self.model.materials_a, self.model.materials_r],
                                      step_in_graph=True if replicated else None)
            self._gs_inputs = getattr(self, "_gs_inputs", {})
        if (vid0, stage) not in self._gs.graphs:
            # a captured (view, stage) graph owns no gradient memory (the stacks live in the optimiser's arena, shared by all graphs); the cap
            # on their number (TEXIR_MAX_GRAPHS, default 1024; beyond it further views run the eager step) is a safety valve only
            if len(self._gs.graphs) >= int(os.environ.get("TEXIR_MAX_GRAPHS", "1024")):
                return None
            mvp, cam, gt, gmask = self._view_inputs(gt_item, vid0)
            try:
                with phases.phase("in_stages:graph_capture"):
                    self._gs.capture(vid0, mvp, cam, gt, gmask, self.seg_mask[str(vid0)], self.floor_max_mask[str(vid0)],
                                     self.room_seg_mask[str(vid0)] if stage == 2 else None, stage)
            except Exception as e:          # capture is an optimisation, not a requirement
                print("hipGraph capture unavailable (%s); continuing with eager steps" % (str(e).splitlines()[0][:160],), file=sys.stderr)
                self.use_graph = False
                self.model._static_shift = None
                return None
        world = dist_util.world_info()[1]
        staged = getattr(self, "_shift_staged_for", None) == (id(self._gs), vid0, stage)       # (_prepare_next_step drew and placed this step's shifts already)
        self._shift_staged_for = None
        self._gs.step(vid0, stage, reduce_grads=dist_util.reduce_texture_grads if (world > 1 and not replicated) else None, staged=staged)
        out = self._gs.outs[(vid0, stage)]
        return out[0], out[1]

    def _prepare_next_step(self, next_item, stage):
        """between two steps of one epoch (RunnerBase.fit between_steps): the NEXT step's GGX shifts are drawn from the CPU generator and written where
        that view's recorded kernels read them, while the step just launched runs on the GPU -- before this step's `.item()` instead of after it.
        The generator's stream is consumed in the same order as in the reference's loop (nothing else draws between two steps of an epoch: the
        sampler drew its permutation when the epoch's iterator was made, the validation forwards run at epoch starts), so the trajectory is unchanged;
        what changes is that the 0.6 ms draw no longer sits between two 0.6 ms steps.  Only for a view whose (view, stage) graph exists."""
        gs = getattr(self, "_gs", None)
        if stage == 0 or gs is None or not getattr(self, "use_graph", False) or gs.opt is not self.mat_optimizer:
            return
        if dist_util.world_info()[1] > 1 and getattr(self, "mat_shard", "pixel") == "pixel":
            return                                      # (the pixel-sharded step draws inside ShardedMatStep.step)
        vid = next_item["id"]
        vid0 = vid[0] if isinstance(vid, (list, tuple)) else vid
        if (vid0, stage) not in gs.graphs:
            return
        if not gs.request_shift(vid0):              # (helper thread, straight into the view's pinned buffer; awaited by GraphedMatStep.step)
            gs.stage_shift(vid0, gs.draw_shift())
        self._shift_staged_for = (id(gs), vid0, stage)

    def _new_optimizer(self):
        """fresh Adam + StepLR over ALL model parameters (train_material.py:122-128, 472-476, 539-543); the post-step
        clamps of :458/:592-593 are fused into the optimiser kernel"""
        # the last mip fold of the texture backward is fused into the optimiser's read of the gradient (bit-identical); with several
        # ranks dist_util.reduce_texture_grads sums the (level 0, level 1) pair instead of the folded gradient
        self.mat_optimizer = FusedAdam(self.model.parameters(), lr=self.conf.get_float("train.mat_learning_rate"), fuse_mip_fold=True)
        self.mat_optimizer.set_clamp(self.model.materials_r, 1e-2, 0.8)
        self.mat_scheduler = torch.optim.lr_scheduler.StepLR(self.mat_optimizer, self.conf.get_int("train.mat_sched_step", default=100),
                                                            gamma=self.conf.get_float("train.mat_sched_factor", default=0.0))

    def plot_materials(self):
        """the user-facing output of plot_to_disk_cube (train_material.py:352-353): the current material textures as
        plots/mat_albedo-1_<iter>.hdr and plots/mat_roughness-1_<iter>.hdr (Radiance RGBE, roughness replicated to 3 channels)"""
        plots = getattr(self, "plots_dir", None)
        if int(os.environ.get("RANK", "0")) != 0 or not plots or not os.path.isdir(plots):
            return
        # device snapshot + pinned copy + encode + write on a worker thread (plot_writer.py): the step loop keeps queuing meanwhile
        from ..plot_writer import AsyncPlotWriter
        if getattr(self, "_plots", None) is None:
            self._plots = AsyncPlotWriter()
        with phases.phase("in_stages:plot_submit", sync=False):
            self._plots.submit(os.path.join(self.plots_dir, "mat_albedo-1_%d.hdr" % self.cur_iter), self.model.materials_a.detach()[:, :, 0:3])
            self._plots.submit(os.path.join(self.plots_dir, "mat_roughness-1_%d.hdr" % self.cur_iter), self.model.materials_r.detach()[:, :, 0:1], repeat3=True)

    # first-validation branch of plot_to_disk_cube (train_material.py:251-296): per-view masks from a stage -1 render
    def build_view_masks(self):
        self.model.eval()
        with torch.no_grad():
            for i, vid in enumerate(self.train_dataset.ids):
                segs = self.train_dataset.images_items[i]["segs"].float().cuda()
                res = self.model(self.train_dataset.extrinsics_list[i], vid, self.train_dataset.cam_position_list[i].cuda(), -1)
                seg, fm, room = build_masks(segs, res["rgb"], self.room_img, res["position"],
                                            (self.room_meta_scale, self.room_meta_w, self.room_meta_h, self.room_meta_xmin, self.room_meta_zmin))
                self.seg_mask[str(vid)], self.floor_max_mask[str(vid)], self.room_seg_mask[str(vid)] = seg, fm, room
        self.model.train()
        self.first_val = False
        self.plot_materials()

    def train_step(self, gt_item, stage):
        """one optimiser step (train_material.py:424-458 / 486-525 / 552-593)"""
        rank, world, _ = dist_util.world_info()
        pixel = world > 1 and getattr(self, "mat_shard", "pixel") == "pixel"
        if pixel and stage > 0:
            # parity mode: the pixels of this ONE view are split across the ranks for the specular trace; everything on the texture
            # side is replicated -- the optimisation trajectory is the single-GPU one bit for bit (sharded_step.py)
            return self._sharded_step(gt_item, stage)
        if getattr(self, "use_graph", False):
            res = self._graph_step(gt_item, stage, replicated=pixel)
            if res is not None:
                return res
        gt_color = gt_item["color"].float().cuda()
        h, w, c = gt_color.shape[-3:]
        gt_color = gt_color.reshape(-1, h, w, c)
        gt_mask = gt_item["mask"].float().cuda().reshape(-1, h, w, 1)
        mvp = gt_item["cam_to_world"].float()
        vid = gt_item["id"]
        vid0 = vid[0] if isinstance(vid, (list, tuple)) else vid
        cam = gt_item["cam_position"].float().cuda()
        fm, seg = self.floor_max_mask[str(vid0)], self.seg_mask[str(vid0)]
        if pixel:
            for p in (self.model.materials_a, self.model.materials_r):
                p._texir_replicated_grads = True             # (stage 0 of the pixel mode: every rank runs the identical Lambertian step)
        preds = self.model(mvp[0] if mvp.dim() == 4 else mvp, vid0, cam[0] if cam.dim() == 2 else cam, stage)
        out = self.mat_loss(gt_color, preds, gt_mask, fm, seg, stage=stage, room_seg_mask=self.room_seg_mask[str(vid0)] if stage == 2 else None)
        loss = out[0]
        self.mat_optimizer.zero_grad()
        loss.backward()
        if world > 1 and not pixel:
            dist_util.reduce_texture_grads([self.model.materials_a, self.model.materials_r])
        self.mat_optimizer.step()
        return loss, out[1]

    def validation_forward(self, stage):
        """non-first branch of plot_to_disk_cube (train_material.py:319-345): one eval forward per view.  The images it
        plots are debug output (not reproduced), but the forward draws GGX shifts from the global CPU generator, so it is
        kept to consume the random stream exactly like the reference does."""
        self.model.eval()
        with phases.phase("in_stages:validation_forward"), torch.no_grad():
            for i, vid in enumerate(self.train_dataset.ids):
                self.model(self.train_dataset.extrinsics_list[i], vid, self.train_dataset.cam_position_list[i].cuda(), stage)
        self.model.train()
        self.plot_materials()

    def _stage(self, stage, max_steps=None):
        """one stage of train_material.py:416-605 as hooks on the shared loop"""
        rank, world, _ = dist_util.world_info()
        t0 = [0.0]

        def epoch_begin(epoch):
            if stage > 0 and epoch % self.plot_freq == 0 and not self.cur_iter == 0:      # train_material.py:484-485, 550-551
                self.validation_forward(stage)
            if world > 1 and self.mat_shard == "view" and len(self.train_dataset) % world:
                raise ValueError("train.mat_shard = view needs the number of views (%d) to be a multiple of the world size (%d)"
                                 % (len(self.train_dataset), world))

        def takes(data_index):
            # throughput mode: `world` different views per optimiser step (gradients summed)
            return not (world > 1 and self.mat_shard == "view" and data_index % world != rank)

        def before_step(epoch, data_index):
            t0[0] = time.time()

        ring, ring_next = [], [0]
        pending = []             # log_lag > 0: (epoch, data_index, pinned [2] buffer, event, seconds the step's launch took)

        def report(epoch, data_index, loss_v, seg_v, dt, it):
            self.log.append((stage, epoch, data_index, float(loss_v), float(seg_v)))     # the reference prints .item() every step too
            # scalars under the reference's names at the step's own (pre-increment) iteration index (train_material.py:465-468, 532-536, 600-604)
            w = getattr(self, "writer", None)                 # (a runner assembled without setup_experiment -- tests -- has no run directory and no log)
            if w is not None:
                lt = self.conf.get_string("render_loss.loss_type")
                w.add_scalar("img_loss_%s_stage%d" % (lt, stage), loss_v, it)
                w.add_scalar("seg_loss_%s_stage%d" % (lt, stage), seg_v, it)
                if stage > 0:
                    w.add_scalar("tv_loss_%s_stage%d" % (lt, stage), 0.0, it)       # (always 0 in the reference: models/loss.py:105,115)
            print("{0} [{1}] ({2}/{3}): img_loss_stage{7} ({5}) = {4}, seg_loss = {6}, batch cost time : {8:.4f}s".format(
                self.expname, epoch, data_index, self.n_batches, loss_v, self.conf.get_string("render_loss.loss_type"), seg_v, stage, dt))

        def drain(keep):
            while len(pending) > keep:
                epoch, data_index, host, ev, dt, it = pending.pop(0)
                ev.synchronize()
                report(epoch, data_index, float(host[0]), float(host[1]), dt, it)

        def after_step(epoch, data_index, out):
            loss, seg_item = out
            if getattr(self, "log_lag", 0) > 0 and torch.is_tensor(loss) and loss.is_cuda:
                # (a recorded step's loss lives in a static tensor the next replay overwrites: copy it out on the stream, right behind the step)
                if not ring:                     # (lag + 2 pinned pairs and their events, reused round robin: a slot is rewritten only after its line was printed)
                    ring.extend((torch.empty(2, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(getattr(self, "log_lag", 0) + 2))
                host, ev = ring[ring_next[0] % len(ring)]
                ring_next[0] += 1
                host[0:1].copy_(loss.detach().reshape(1), non_blocking=True)
                if torch.is_tensor(seg_item):
                    host[1:2].copy_(seg_item.detach().reshape(1).to(torch.float32), non_blocking=True)
                else:
                    host[1] = float(seg_item)
                ev.record()
                pending.append((epoch, data_index, host, ev, time.time() - t0[0], self.cur_iter - 1))
                drain(getattr(self, "log_lag", 0))
            else:
                with phases.phase("in_stages:loss_item_wait", sync=False):
                    loss_v = loss.item()
                report(epoch, data_index, loss_v, seg_item, time.time() - t0[0], self.cur_iter - 1)
            return max_steps is not None and self.cur_iter >= max_steps

        try:
            def between_steps(next_item, epoch, next_index):
                # (never for a step that will not happen: a step budget that ends the run after the current step must leave the generator where the reference's loop leaves it)
                if max_steps is None or self.cur_iter < max_steps:
                    self._prepare_next_step(next_item, stage)

            self.fit(self.train_dataloader, self.start_epoch, self.nepochs, lambda gt_item: self.train_step(gt_item, stage), epoch_begin=epoch_begin,
                     takes=takes, before_step=before_step, after_step=after_step, epoch_end=lambda epoch: (drain(getattr(self, "log_lag", 0)), self.mat_scheduler.step()),
                     between_steps=between_steps)
        finally:
            drain(0)

    def run(self):
        print("training...")
        self.cur_iter = self.start_epoch * len(self.train_dataloader)
        with phases.phase("view_masks"):
            self.build_view_masks()                              # "generate vhl mask" (train_material.py:413)
        # stage 0: albedo only (:416-469)
        self.model.materials_r.requires_grad = False
        self.model.materials_a.requires_grad = True
        with phases.phase("stage0"):
            self._stage(0)
        # stage 1: roughness on highlights (:472-536)
        self._new_optimizer()
        self.model.materials_a.data = torch.clamp(self.model.materials_a.data, 0.)
        self.model.materials_a.requires_grad = False
        self.model.materials_r.requires_grad = True
        with phases.phase("stage1"):
            self._stage(1)
        # stage 2: joint (:539-605); materials_a >= 0 after every step (:592)
        self._new_optimizer()
        self.mat_optimizer.set_clamp(self.model.materials_a, 0.0, float("inf"))
        self.model.materials_a.requires_grad = True
        self.model.materials_r.requires_grad = True
        with phases.phase("stage2"):
            self._stage(2)
        if int(os.environ.get("RANK", "0")) == 0 and os.path.isdir(os.path.join(self.checkpoints_path, "ModelParameters")):
            with phases.phase("checkpoint", sync=False):
                self.save_checkpoints(self.nepochs)
        self.plot_materials()                                    # final textures (the reference's last plot is one plot_freq earlier)
        self.finish_outputs()

    def finish_outputs(self):
        """every queued plot on disk, the scalar log flushed"""
        with phases.phase("plot_drain", sync=False):
            if getattr(self, "_plots", None) is not None:
                self._plots.close()
                self._plots = None
        if getattr(self, "writer", None) is not None:
            self.writer.flush()


class MatTrainSynRunner(MatTrainRunner):
    """trainer/train_material_syn.py: the same three-stage loop (its run(), :535-733, is MatTrainRunner.run) followed by the tail
    `self.model.sample_l[1] = 256; self.render_calculate()` (:735-736): every training view and every novel view re-rendered at 256
    specular samples and scored against the synthetic ground truth."""

    def postprocessing_materials(self, segs, albedo, roughness):
        """train_material_syn.py:374-392: lamps (27) and ceiling (43) get fixed materials for all methods"""
        for cls, a, r in ((27.0, 0.8, 1.0), (43.0, 0.9, 0.8)):
            m = segs == cls
            albedo = torch.where(m, torch.full_like(albedo, a), albedo)
            roughness = torch.where(m, torch.full_like(roughness, r), roughness)
        return albedo, roughness

    def render_calculate(self, stage=2):
        """train_material_syn.py:394-523: MSE / PSNR / SSIM of the tone-mapped re-renderings (training and novel views) and of the
        albedo (least-squares scaled, :440) and roughness panoramas against the dataset's ground truth.  (LPIPS is commented out in
        the reference and prints 0.)  Returns the numbers it prints."""
        from .. import metrics as M
        from ..cube2pano import Cube2Pano
        self.model.eval()
        c2p = Cube2Pano(pano_width=self.pano_res[1], pano_height=self.pano_res[0], cube_lenth=self.cube_lenth)
        c = self.cube_lenth

        def pano(x):
            return c2p.ToPano(x.detach().cpu().permute(0, 3, 1, 2).reshape(1, -1, c, c))[0].permute(1, 2, 0)

        def ssim_err(a, b):
            # the reference accumulates 1 - SSIMLoss = the SSIM value itself under the name "ssim_error" (:453)
            return float(M.ssim(a.unsqueeze(0).permute(0, 3, 1, 2), b.unsqueeze(0).permute(0, 3, 1, 2)))

        def mse(a, b):
            return float(torch.mean((a - b) ** 2))

        ds = self.train_dataset
        acc = {k: 0.0 for k in ("mse", "ssim", "albedo_mse", "albedo_ssim", "roughness_mse", "roughness_ssim")}
        n_mat = 0
        with torch.no_grad():
            for i in range(len(ds.ids)):
                it = ds.images_items[i]
                gt_img = pano(it["color"])
                res = self.model(ds.extrinsics_list[i], ds.ids[i], ds.cam_position_list[i].cuda(), stage)
                pred_img = pano(res["rgb"])
                acc["ssim"] += ssim_err(M.tonemapping(gt_img), M.tonemapping(pred_img))
                acc["mse"] += mse(M.tonemapping(gt_img), M.tonemapping(pred_img))
                if "gt_albedo" in it and "gt_roughness" in it:
                    segs = it["segs_pano"].expand(-1, -1, 3).float()
                    pred_albedo, pred_r = pano(res["albedo"]), pano(res["roughness"].expand(-1, -1, -1, 3))
                    gt_a = it["gt_albedo"]
                    gt_r = it["gt_roughness"] if it["gt_roughness"].dim() == 3 else it["gt_roughness"].unsqueeze(-1).expand(-1, -1, 3)
                    pred_albedo, pred_r = self.postprocessing_materials(segs, M.scale_compute(gt_a, pred_albedo) * pred_albedo, pred_r)
                    acc["albedo_ssim"] += ssim_err(gt_a, torch.clamp(pred_albedo, 0.0, 1.0))
                    acc["albedo_mse"] += mse(gt_a, torch.clamp(pred_albedo, 0.0, 1.0))
                    acc["roughness_ssim"] += ssim_err(gt_r, torch.clamp(pred_r, 0.0, 1.0))
                    acc["roughness_mse"] += mse(gt_r, torch.clamp(pred_r, 0.0, 1.0))
                    n_mat += 1
            n = max(1, len(ds.ids))
            out = {"mse": acc["mse"] / n, "ssim": acc["ssim"] / n}
            out["psnr"] = float(M.mse_to_psnr(torch.tensor(out["mse"])))
            print("re-rendering error: mse: {}, psnr: {}, ssim: {}, lpips: {}".format(out["mse"], out["psnr"], out["ssim"], 0.0))
            for k in ("albedo", "roughness"):
                if n_mat:
                    out[k + "_mse"], out[k + "_ssim"] = acc[k + "_mse"] / n_mat, acc[k + "_ssim"] / n_mat
                    out[k + "_psnr"] = float(M.mse_to_psnr(torch.tensor(out[k + "_mse"])))
                    print("{} error: mse: {}, psnr: {}, ssim: {}, lpips: {}".format(k, out[k + "_mse"], out[k + "_psnr"], out[k + "_ssim"], 0.0))
            nv = getattr(ds, "novel_ids", [])
            if nv:
                m_, s_ = 0.0, 0.0
                for i in range(len(nv)):
                    gt_img = pano(ds.novel_images_items[i]["color"])
                    res = self.model(ds.novel_extrinsics_list[i], nv[i], ds.novel_cam_position_list[i].cuda(), stage)
                    pred_img = pano(res["rgb"])
                    s_ += ssim_err(M.tonemapping(gt_img), M.tonemapping(pred_img))
                    m_ += mse(M.tonemapping(gt_img), M.tonemapping(pred_img))
                out["novel_mse"], out["novel_ssim"] = m_ / len(nv), s_ / len(nv)
                out["novel_psnr"] = float(M.mse_to_psnr(torch.tensor(out["novel_mse"])))
                print("novel view re-rendering error: mse: {}, psnr: {}, ssim: {}, lpips: {}".format(out["novel_mse"], out["novel_psnr"], out["novel_ssim"], 0.0))
        self.model.train()
        self.metrics = out
        return out

    def run(self):
        super().run()
        self.model.sample_l[1] = 256                      # train_material_syn.py:735
        self.render_calculate()                           # :736
