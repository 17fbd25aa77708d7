"""Run-directory side outputs of the runners that are not on the device path: the scalar log and per-phase wall-clock.

ScalarLog -- the reference hands its step scalars to tensorboardX (`self.writer = SummaryWriter(os.path.join(self.expdir, self.timestamp))`,
`self.writer.add_scalar('img_loss_{}_stage0'.format(loss_type), radiance_loss.item(), self.cur_iter)`: trainer/train_material.py:82,465-466,
532-534,600-602; trainer/train_irrf.py:81,272).  tensorboardX is not part of this image; the same (tag, value, step) triples go to
`<expdir>/<timestamp>/scalars.jsonl`, one JSON object per line, under the reference's tag names, so that a maintainer can feed them to any
plotting tool (or to a SummaryWriter) unchanged.

PhaseTimer -- `with phases.phase("name"):` accumulates wall-clock per named phase of a stage (load, BVH, G-buffer, kernel, write ...) when
timing is switched on (bench.py --e2e, TEXIR_STAGE_TIMING=1); switched off it costs one attribute test and never synchronises."""
import json
import os
import time
from contextlib import contextmanager


def _json_float(v):
    return repr(v) if v == v and v not in (float("inf"), float("-inf")) else ("NaN" if v != v else ("Infinity" if v > 0 else "-Infinity"))


class ScalarLog:
    def __init__(self, directory, name="scalars.jsonl", flush_every=64):
        self.path = os.path.join(directory, name) if directory and os.path.isdir(directory) else None
        self._f = open(self.path, "a", buffering=1) if self.path else None          # line-buffered: a crash mid-stage keeps every scalar written so far
        self._n, self._every = 0, flush_every

    def add_scalar(self, tag, value, step):
        """SummaryWriter.add_scalar(tag, scalar_value, global_step)"""
        if self._f is None:
            return
        # (written by hand: three of these per optimiser step; json.dumps of a dict cost 14 us each -- tags are the reference's plain ASCII names)
        self._f.write('{"tag": "%s", "value": %s, "step": %d, "wall_time": %.3f}\n' % (tag, _json_float(float(value)), int(step), time.time()))
        self._n += 1
        if self._n % self._every == 0:
            self._f.flush()

    def flush(self):
        if self._f is not None:
            self._f.flush()

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_scalars(path):
    """-> {tag: [(step, value), ...]} of a scalars.jsonl"""
    out = {}
    with open(path) as f:
        for line in f:
            if line.strip():
                r = json.loads(line)
                out.setdefault(r["tag"], []).append((r["step"], r["value"]))
    return out


class PhaseTimer:
    def __init__(self):
        self.enabled = os.environ.get("TEXIR_STAGE_TIMING", "0") not in ("", "0")
        self.times, self.order = {}, []

    def reset(self, enabled=True):
        self.enabled, self.times, self.order = enabled, {}, []

    @contextmanager
    def phase(self, name, sync=True):
        if not self.enabled:
            yield
            return
        t0 = time.perf_counter()
        try:
            yield
        finally:
            if sync:
                try:
                    import torch
                    if torch.cuda.is_available():
                        torch.cuda.synchronize()
                except Exception:
                    pass
            if name not in self.times:
                self.times[name] = 0.0
                self.order.append(name)
            self.times[name] += time.perf_counter() - t0

    def report(self):
        return {k: round(self.times[k], 4) for k in self.order}


phases = PhaseTimer()
