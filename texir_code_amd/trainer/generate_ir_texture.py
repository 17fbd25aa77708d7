"""IrrTextureRunner -- drop-in for trainer/generate_ir_texture.py:31-82 (same kwargs, same output file)."""
import os
import sys

import numpy as np
import torch

from .. import io_formats as IO
from ..conf import ConfigFactory
from ..plugin import get_class
from ..runlog import phases


class IrrTextureRunner:
    def __init__(self, **kwargs):
        torch.set_default_dtype(torch.float32)
        torch.set_num_threads(1)                 # as the reference's runners (e.g. trainer/train_material.py:34): host torch ops are tiny
        self.conf = ConfigFactory.parse_file(kwargs["conf"])
        self.exps_folder_name = kwargs["exps_folder_name"]
        self.train_batch_size = self.conf.get_int("train.batch_size")
        self.nepochs = self.conf.get_int("train.mat_epoch")
        self.max_niters = kwargs["max_niters"]
        self.GPU_INDEX = kwargs["gpu_index"]
        # fix random seed (generate_ir_texture.py:45-47): the per-texel shifts come from this CPU generator
        torch.manual_seed(666)
        torch.cuda.manual_seed(666)
        np.random.seed(666)
        print("shell command : {0}".format(" ".join(sys.argv)))
        print("Loading data ...")
        with phases.phase("dataset", sync=False):
            self.train_dataset = get_class(self.conf.get_string("train.dataset_class"))(
                self.conf.get_string("train.path_mesh_open3d"), self.conf.get_list("train.pano_img_res"), self.conf.get_float("train.hdr_exposure"))
        print("Finish loading data ...")
        with phases.phase("model_init"):
            self.model = get_class(self.conf.get_string("train.model_class"))(
                conf=self.conf, ids=self.train_dataset.ids, extrinsics=self.train_dataset.extrinsics_list, optim_cam=self.conf.get_bool("train.optim_cam"))
            self.model.cuda()
        self.start_epoch = 0

    def run(self):
        print("generating...")
        irr_texture = self.model()
        target = self.conf.get_string("train.path_mesh_open3d").replace("out1.obj", "0_irr_texture.hdr")
        rank = int(os.environ.get("RANK", "0"))
        with phases.phase("download", sync=False):
            # pinned staging: the 4096^2 texture is 201 MB; a pageable .cpu() goes through the driver's bounce buffers at a fraction of the link rate
            host = torch.empty(irr_texture.shape, dtype=irr_texture.dtype, pin_memory=True)
            host.copy_(irr_texture)
            arr = host.numpy()
        print(arr.shape)
        if rank == 0:
            with phases.phase("write_hdr", sync=False):
                IO.write_hdr(target, arr)          # Radiance RGBE (RLE scanlines) like cv2.imwrite('.hdr') (generate_ir_texture.py:82)
        return irr_texture
