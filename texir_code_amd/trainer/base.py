"""What the runners of trainer/*.py share: experiment directories, checkpoints, and ONE epoch / batch loop with hooks.

The reference repeats this scaffolding in every runner (trainer/train_material.py:31-130, trainer/train_irrf.py:27-140 set up the
same ../<exps>/<prefix>-<name>/<timestamp>/{plots,checkpoints/ModelParameters} tree; their run() methods are the same nested loop with
different things hung on it).  Here the runners declare WHAT happens at each hook; the order of the hooks is the order in which the
reference's loops do things, which is what keeps the optimisation trajectories (and the CPU random stream) identical."""
import os
import sys
from datetime import datetime

import torch

from ..conf import ConfigFactory
from ..runlog import phases


class RunnerBase:
    model_params_subdir = "ModelParameters"

    def setup_experiment(self, kwargs, prefix, exps_root="../", make_dirs=True, keep_conf_copy=False):
        """conf, names and the run directory layout of the reference's runners (kwargs of trainer/exp_runner.py:69-80)"""
        torch.set_default_dtype(torch.float32)
        torch.set_num_threads(1)                 # as the reference's runners (e.g. trainer/train_material.py:34): host torch ops are tiny
        self.conf = ConfigFactory.parse_file(kwargs["conf"])
        self.exps_folder_name = kwargs["exps_folder_name"]
        self.train_batch_size = self.conf.get_int("train.batch_size")
        self.max_niters = kwargs["max_niters"]
        self.GPU_INDEX = kwargs["gpu_index"]
        self.expname = prefix + "-" + kwargs["expname"]
        self.expdir = os.path.join(exps_root, self.exps_folder_name, self.expname)
        # earlier runs of this experiment, listed BEFORE this run's own directory is made (`--is_continue --timestamp latest` picks the last one)
        self.prior_timestamps = sorted(os.listdir(self.expdir)) if os.path.isdir(self.expdir) else []
        self.timestamp = "{:%Y_%m_%d_%H_%M_%S}".format(datetime.now())
        self.plots_dir = os.path.join(self.expdir, self.timestamp, "plots")
        self.checkpoints_path = os.path.join(self.expdir, self.timestamp, "checkpoints")
        if make_dirs:
            for d in (self.plots_dir, os.path.join(self.checkpoints_path, self.model_params_subdir)):
                os.makedirs(d, exist_ok=True)
            if keep_conf_copy:
                try:
                    import shutil
                    shutil.copy(kwargs["conf"], os.path.join(self.expdir, self.timestamp, "runconf.conf"))
                except OSError:
                    pass
        # the reference's `self.writer = SummaryWriter(os.path.join(self.expdir, self.timestamp))` (trainer/train_material.py:82): scalars.jsonl there
        from ..runlog import ScalarLog
        # (rank 0 only: several ranks appending to one scalars.jsonl would interleave duplicate lines; line-buffered, so an exception mid-stage loses nothing)
        self.writer = ScalarLog(os.path.join(self.expdir, self.timestamp) if (make_dirs and int(os.environ.get("RANK", "0")) == 0) else None)
        print("shell command : {0}".format(" ".join(sys.argv)))

    def save_checkpoints(self, epoch):
        torch.save({"epoch": epoch, "model_state_dict": self.model.state_dict()},
                   os.path.join(self.checkpoints_path, self.model_params_subdir, "latest.pth"))

    def fit(self, loader, first_epoch, last_epoch, step, epoch_begin=None, takes=None, before_step=None, after_step=None, epoch_end=None, between_steps=None):
        """for epoch in first_epoch..last_epoch: epoch_begin(epoch); for every batch this rank `takes`: before_step, step, after_step;
        epoch_end(epoch).  A hook that returns True from after_step ends the run (the runners' step budgets).  Returns True if ended early.
        between_steps(next_batch, epoch, next_index) runs after a step has been LAUNCHED and before after_step (which is where the reference's loops
        synchronise with `.item()`), with the next batch of the SAME epoch -- the place to prepare the next step's host-side inputs while the GPU
        works (the batch order of an epoch is fixed when its iterator is made, so looking one batch ahead draws nothing)."""
        for epoch in range(first_epoch, last_epoch + 1):
            if epoch_begin is not None:
                epoch_begin(epoch)
            it = iter(loader)
            data_index = -1
            with phases.phase("in_stages:dataloader", sync=False):
                nxt = next(it, None)
            while nxt is not None:
                batch = nxt
                data_index += 1
                if takes is not None and not takes(data_index):
                    with phases.phase("in_stages:dataloader", sync=False):
                        nxt = next(it, None)
                    continue
                self.model.train()
                if before_step is not None:
                    before_step(epoch, data_index)
                with phases.phase("in_stages:step_host", sync=False):
                    out = step(batch)
                self.cur_iter += 1
                with phases.phase("in_stages:dataloader", sync=False):
                    nxt = next(it, None)
                if nxt is not None and between_steps is not None and (takes is None or takes(data_index + 1)):
                    with phases.phase("in_stages:next_step_prep", sync=False):
                        between_steps(nxt, epoch, data_index + 1)
                if after_step is not None and after_step(epoch, data_index, out):
                    return True
            if epoch_end is not None:
                epoch_end(epoch)
        return False
