"""CLI -- drop-in for trainer/exp_runner.py:26-80: same flags and defaults, same --trainstage keys, same Runner kwargs.

    python -m texir_code_amd.trainer.exp_runner --conf configs/x.conf --trainstage IrrT --gpu 0
    python -m torch.distributed.run --nproc-per-node 8 -m texir_code_amd.trainer.exp_runner ...   (texel / gradient sharding)
"""
import argparse
import os

import torch

IN_SCOPE = ("IrrT", "Mat", "MatSyn", "IRRF")
ALL_STAGES = ("IRF", "Mat", "IRRF", "PIL", "Inv", "Neilf", "IrrT", "RecMLP", "MatSyn", "RecMLPSyn", "NeilfSyn", "InvSyn")


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--conf", type=str, default="")
    parser.add_argument("--exps_folder_name", type=str, default="exps")
    parser.add_argument("--expname", type=str, default="")
    parser.add_argument("--trainstage", type=str, default="IRF", help="")
    parser.add_argument("--frame_skip", type=int, default=1, help="skip frame when training")
    parser.add_argument("--max_niter", type=int, default=200001, help="max number of iterations to train for")
    parser.add_argument("--is_continue", default=False, action="store_true", help="If set, indicates continuing from a previous run.")
    parser.add_argument("--timestamp", default="latest", type=str, help="The timestamp of the run to be used in case of continuing from a previous run.")
    parser.add_argument("--checkpoint", default="latest", type=str, help="The checkpoint epoch number of the run to be used in case of continuing from a previous run.")
    parser.add_argument("--gpu", type=str, default="auto", help="GPU to use [default: GPU auto]")
    parser.add_argument("--detect_anomaly", default=False, action="store_true", help="opt-in torch.autograd.set_detect_anomaly (the reference enables it globally)")
    return parser


def pick_gpu(opt_gpu):
    """--gpu auto: under torchrun use LOCAL_RANK; otherwise the device with most free memory (GPUtil in the reference)"""
    if "LOCAL_RANK" in os.environ:
        return int(os.environ["LOCAL_RANK"])
    if opt_gpu != "auto":
        return int(opt_gpu)
    best, best_free = 0, -1
    for i in range(torch.cuda.device_count()):
        free, _ = torch.cuda.mem_get_info(i)
        if free > best_free:
            best, best_free = i, free
    return best


def runner_class(stage):
    if stage not in ALL_STAGES:
        raise KeyError(stage)
    if stage not in IN_SCOPE:
        raise NotImplementedError("--trainstage %s is outside the IrT + material-estimation hot path this build covers (in scope: %s)"
                                  % (stage, ", ".join(IN_SCOPE)))
    from .generate_ir_texture import IrrTextureRunner
    from .train_irrf import IRRFTrainRunner
    from .train_material import MatTrainRunner, MatTrainSynRunner
    return {"IrrT": IrrTextureRunner, "Mat": MatTrainRunner, "MatSyn": MatTrainSynRunner, "IRRF": IRRFTrainRunner}[stage]


def main(argv=None):
    opt = build_parser().parse_args(argv)
    if opt.detect_anomaly:
        torch.autograd.set_detect_anomaly(True)
    gpu = pick_gpu(opt.gpu)
    torch.cuda.set_device(gpu)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))
    runner = runner_class(opt.trainstage)(conf=opt.conf, exps_folder_name=opt.exps_folder_name, expname=opt.expname, frame_skip=opt.frame_skip,
                                          max_niters=opt.max_niter, is_continue=opt.is_continue, timestamp=opt.timestamp,
                                          checkpoint=opt.checkpoint, gpu_index=gpu)
    runner.run()


if __name__ == "__main__":
    main()
