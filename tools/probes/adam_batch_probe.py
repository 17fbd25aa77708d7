"""The batched Adam launch alone (texir_adam_step_tex_dev_batch over a 4096^2 x 3 and a 4096^2 x 1 texture, the material step's pair): microseconds per launch
and bytes moved / time, in the step's own conditions -- level-0 gradient absent, level-1 stack read through a ~1 % mask, level-2 stack dense, mip level 1 written.
The arrays (1.7 GB per launch) do not fit the 256 MB Infinity Cache, so repeated launches stream from HBM as the step does.
usage: [TEXIR_HIP_LIB=...] [TEXIR_ADAM_GRID_Y=n] python tools/probes/adam_batch_probe.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from texir_code_amd import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = "cuda"
jobs, keep, nbytes = [], [], 0
A = _lib.addr
hyper = torch.tensor([[1e-3, 0.9], [1e-3, 0.9]], device=dev)
for k, C in enumerate((3, 1)):
    H = W = 4096
    p, m, v = (torch.rand(H, W, C, device=dev) for _ in range(3))
    n1, n2 = (H // 2) * (W // 2), (H // 4) * (W // 4)
    g1, g2, mip1 = torch.randn(n1 * C, device=dev), torch.randn(n2 * C, device=dev), torch.empty(n1 * C, device=dev)
    bits = torch.rand(n1, device=dev) < 0.01
    w = (bits.view(-1, 32).to(torch.int64) << torch.arange(32, device=dev)).sum(1) & 0xFFFFFFFF
    mask = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()
    keep += [p, m, v, g1, g2, mip1, mask]
    jobs.append(_lib.AdamTexJob(A(p), None, None, A(g1), A(mask), A(g2), A(m), A(v), A(mip1), H, W, C, A(hyper[k]), 0.9, 0.999, 1e-8, 0.0, 1.0))
    nbytes += 6 * p.numel() * 4 + mip1.numel() * 4 + g2.numel() * 4 + int(bits.sum()) * 4 * C + mask.numel() * 4
for _ in range(3):
    _lib.batch_call("texir_adam_step_tex_dev_batch", jobs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = []
for _ in range(3):
    e0.record()
    for _ in range(reps):
        _lib.batch_call("texir_adam_step_tex_dev_batch", jobs)
    e1.record()
    torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) * 1e3 / reps)
us = min(best)
print("adam batch  %-28s grid_y %-5s  %.1f us  (%.2f TB/s on %.3f GB)   runs %s" % (os.path.basename(os.environ.get("TEXIR_HIP_LIB", "default")), os.environ.get("TEXIR_ADAM_GRID_Y", "-"),
                                                                                  us, nbytes / us / 1e6, nbytes / 1e9, " ".join("%.1f" % b for b in best)), flush=True)
