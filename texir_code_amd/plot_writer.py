"""Texture plots off the critical path.

The reference's plot_to_disk_cube ends with `plt.plot_mat(..., self.model.materials_a.cpu().detach()[:,:,0:3], ...)` (trainer/train_material.py:350-353):
a synchronous device -> host copy and a cv2.imwrite('.hdr') of both material textures every `plot_freq` epochs, with the optimisation
stopped meanwhile.  At 4096^2 textures that is ~270 MB over PCIe plus two RGBE encodes per event against 0.6 ms optimiser steps.

AsyncPlotWriter keeps the same files and contents but takes them off the step loop: submit() snapshots the tensor on the device (one
D2D copy ordered on the caller's stream -- the optimiser overwrites the parameters in place right after), copies the snapshot to pinned host
memory on a side stream, and hands (event, buffer, path) to one worker thread that waits for the copy, encodes (io_formats.write_hdr:
the library's threaded RGBE + RLE loops, GIL released) and writes.  The caller's cost is the snapshot launch; flush() waits for everything
(end of run, or a test that reads a plot back)."""
import queue
import threading

import numpy as np
import torch

from . import io_formats as IO


_LIVE = None          # weak set of the writers with a worker thread


_GATE = threading.Lock()          # held by a worker around its HIP calls, and by a stream capture for its whole duration


class capture_gate:
    """`with capture_gate():` around a hipGraph capture (graph_step / sharded_step).  While a stream is being captured in the default (global) mode, a HIP call
    from ANY thread -- the worker's hipEventSynchronize on its device -> host copy, or the event the caching allocator records when the worker drops the
    device snapshot -- is an error that also invalidates the capture (seen in round 5: the stage-1 captures right behind the epoch-0 plot fell back to eager
    steps).  Round 5 waited for the writer's whole queue (encode + write, ~0.1 s per plot at 4096^2: 0.27 s of the 40-epoch c4 stage); the gate only keeps the
    worker's two HIP-touching lines out of the capture -- its encode and its write run on while graphs are being recorded."""

    def __enter__(self):
        _GATE.acquire()
        return self

    def __exit__(self, *exc):
        _GATE.release()
        return False


def quiesce():
    """wait until no writer has anything in flight (kept for callers that want the files on disk; captures use capture_gate)"""
    if _LIVE:
        for w in list(_LIVE):
            w._q.join()            # (wait only: a failed write is reported by the writer's own flush() / close(), not disguised as a failed capture)


class AsyncPlotWriter:
    def __init__(self, max_pending=4):
        global _LIVE
        if _LIVE is None:
            import weakref
            _LIVE = weakref.WeakSet()
        _LIVE.add(self)
        self._q = queue.Queue()
        self._pool = {}                 # (shape, dtype) -> [pinned tensors]
        self._lock = threading.Lock()
        self._stream = None
        self._worker = None
        self._errors = []
        self._wide, self._scratch = {}, {}      # worker-owned buffers kept between plot events (no 67-200 MB allocate / release per file)
        self._pending = threading.Semaphore(max_pending)        # bounds pinned memory: submit() blocks when the disk is that far behind

    def _pinned(self, shape, dtype):
        key = (tuple(shape), dtype)
        with self._lock:
            free = self._pool.setdefault(key, [])
            if free:
                return free.pop()
        return torch.empty(shape, dtype=dtype).pin_memory()

    def _run(self):
        while True:
            job = self._q.get()
            if job is None:
                self._q.task_done()
                return
            path, host, ev, repeat3, snap = job
            job = None
            try:
                with _GATE:                     # (never inside a stream capture: capture_gate)
                    if ev is not None:
                        ev.synchronize()
                    snap = None                 # the device snapshot goes back to the allocator here
                if repeat3:
                    # (torch, not numpy: np.repeat writes its 200 MB at 4096^2 with the interpreter lock held -- a 27 ms stall of the training loop per plot
                    # event in the round-6 kernel trace; the destination is kept between events for the same reason)
                    key = (host.shape[0], host.shape[1])
                    wide = self._wide.get(key)
                    if wide is None:
                        wide = self._wide[key] = torch.empty((host.shape[0], host.shape[1], 3), dtype=torch.float32)
                    wide.copy_(host.expand(-1, -1, 3))
                    a = wide.numpy()
                else:
                    a = host.numpy()
                IO.write_hdr(path, a, scratch=self._scratch)
            except Exception as e:              # surfaced by flush()
                self._errors.append((path, e))
            finally:
                if host.is_pinned():
                    with self._lock:
                        self._pool.setdefault((tuple(host.shape), host.dtype), []).append(host)
                self._pending.release()
                self._q.task_done()

    def submit(self, path, tensor, repeat3=False):
        """write `tensor` [H,W,C] (C = 3, or 1 with repeat3) as a Radiance .hdr; returns at once"""
        if self._worker is None:
            self._worker = threading.Thread(target=self._run, name="texir-plot-writer", daemon=True)
            self._worker.start()
        self._pending.acquire()
        t = tensor.detach()
        if t.is_cuda:
            snap = t.to(torch.float32).clone(memory_format=torch.contiguous_format)          # ordered before the next optimiser step
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=t.device)
            host = self._pinned(snap.shape, snap.dtype)
            self._stream.wait_stream(torch.cuda.current_stream(t.device))
            with torch.cuda.stream(self._stream):
                host.copy_(snap, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._stream)
            snap.record_stream(self._stream)
            self._q.put((path, host, ev, repeat3, snap))
        else:
            self._q.put((path, t.to(torch.float32).contiguous().clone(), None, repeat3, None))

    def flush(self):
        self._q.join()
        if self._errors:
            path, e = self._errors[0]
            self._errors = []
            raise RuntimeError("plot writer failed on %s: %s" % (path, e))

    def close(self):
        if self._worker is not None:
            self.flush()
            self._q.put(None)
            self._worker.join()
            self._worker = None
