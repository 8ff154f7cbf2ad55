"""HIP-graph execution of the root-localisation forward (fixed shapes): camera-table fetch, heat-map re-tiling (planar
hand-over only), unprojection, V2V convs + fused epilogues, NMS/top-k are captured once and replayed, so ~70 kernel
launches cost one graph launch.  Per step the host packs the batch's camera table (calibration / crop data) into a
slot of a PINNED ring; the graph's first node (sp3d_fetch_ring) pulls the slot of its replay into the static device
table the captured kernels read - no copy command between graph launches (that cost ~90 us of idle GPU per step)."""
from __future__ import annotations

from typing import List, Sequence

import torch

from . import _lib
from .camera_pack import pack_cameras


class GraphedRootNet:
    """The wrapped net is left exactly as it was: the static camera table is only in force while capturing
    (``ProjectLayer.static_camera_table``), so eager calls on the same net keep working with their own ``meta``."""

    RING = 4      # slots of the pinned ring: the host may run RING - 1 launches ahead of the GPU
    # torch.cuda.Event(blocking=True) for the ring slots was measured (tools/experiments/r05_blocking_events.py,
    # profiles/r05_blocking_events.json): same step time, same host CPU time per step (1.57 ms of the 1.59 ms step either way) -
    # the host's time is not the event wait.  Left selectable for that script; the default is HIP's own wait.
    BLOCKING_EVENTS = False

    def __init__(self, net, heatmaps: Sequence[torch.Tensor], meta: Sequence[dict], flip_xcoords=None, warmup: int = 3,
                 time_unprojection: bool = False, copies: int = 1):
        """``time_unprojection``: measurement only - the graph carries one-thread clock-stamp kernels around
        ``ProjectLayer.get_voxel`` / ``get_voxel_zspectrum`` so that ``unprojection_us()`` reads the kernel's time INSIDE the replayed step (between
        the camera fetch before it and the convolutions behind it), not that of a stand-alone launch."""
        self.net = net
        self._stamps = None
        self.static_hms: List[torch.Tensor] = list(heatmaps)       # the caller writes new heat-maps into these
        dev = heatmaps[0].device
        pl = net.project_layer
        B = heatmaps[0].shape[0]
        self._batch, self._flip, self._meta, self._dev = B, flip_xcoords, meta, dev
        tab = torch.from_numpy(pack_cameras(meta, B, pl.img_size, flip_xcoords))
        self._ring = torch.empty((self.RING,) + tuple(tab.shape), dtype=torch.float32).pin_memory()
        self.blocking_events = bool(self.BLOCKING_EVENTS)
        self._events = [torch.cuda.Event(blocking=self.blocking_events) for _ in range(self.RING)]
        self._counter = torch.zeros(1, dtype=torch.int32, device=dev)      # the kernel's own launch count
        self._launches = 0                                                  # the host's count of fetch launches
        self.cam_dev = torch.empty(tab.shape, dtype=torch.float32, device=dev)
        with pl.static_camera_table(self.cam_dev):
            stream = torch.cuda.Stream(dev)
            stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(stream), torch.no_grad():
                for _ in range(warmup):
                    self._stage(meta)
                    _lib.fetch_ring(self._ring, self.cam_dev, self._counter)
                    net(self.static_hms, meta, flip_xcoords)
                    self._events[(self._launches - 1) % self.RING].record(stream)
            torch.cuda.current_stream(dev).wait_stream(stream)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            if time_unprojection:
                # three one-thread kernels that write the chip-wide 100 MHz clock: before and after get_voxel, and a third
                # right behind the second (stamp-to-stamp distance with nothing in between = what the marker nodes cost)
                import ctypes as C
                lib = _lib.load()
                lib.sp3d_debug_stamp.restype = C.c_int
                lib.sp3d_debug_stamp.argtypes = [C.c_void_p, C.c_void_p]
                self._stamps = torch.zeros(3, dtype=torch.int64, device=dev)
                def stamp(i):
                    _lib.check(lib.sp3d_debug_stamp(self._stamps[i:].data_ptr(), _lib._stream(dev)), "sp3d_debug_stamp")

                def timed(inner):
                    def call(*a, **k):
                        stamp(0)
                        r = inner(*a, **k)
                        stamp(1)
                        stamp(2)
                        return r
                    return call
                # instance attributes: shadow the methods while capturing only (get_voxel_zspectrum: the unprojection fused
                # with the opening conv's z pass, what the root net calls on the root grid since round 6)
                pl.get_voxel, pl.get_voxel_zspectrum = timed(pl.get_voxel), timed(pl.get_voxel_zspectrum)
            # ``copies`` > 1: the same step captured into several graph executables that are replayed in turn.  One executable
            # cannot overlap with itself - the runtime starts replay i + 1 of an executable only after replay i has
            # completed, which leaves the GPU idle for a launch latency between steps; with two executables the next
            # step's packets are queued while the current one runs.  Outputs alternate: ``__call__`` returns the tensors of
            # the copy it launched, valid until that copy is launched again (``copies`` steps later).
            self.copies = max(1, int(copies))
            self._graphs, self._outs = [], []
            try:
                for _ in range(self.copies):
                    gr = self.graph if not self._graphs else torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr), torch.no_grad():
                        _lib.fetch_ring(self._ring, self.cam_dev, self._counter)
                        out = net(self.static_hms, meta, flip_xcoords)
                    self._graphs.append(gr)
                    self._outs.append(out)
                self.out = self._outs[0]
            finally:
                if time_unprojection:
                    del pl.get_voxel, pl.get_voxel_zspectrum
        # the captured kernels read the folded inference plan's tensors by address: keep them alive even if the net
        # drops its plan later (train() / load_state_dict / invalidate_plan)
        self._plan = getattr(getattr(net, "v2v_net", None), "_plan", None)
        self._plan_tensors = dict(self._plan.t) if self._plan is not None else None     # incl. the padded FFT buffers

    def unprojection_us(self):
        """(stamp before -> stamp after get_voxel, stamp -> adjacent stamp) of the LAST replay, microseconds
        (``time_unprojection=True``); the kernel's time in the step is the first minus the second.  Synchronises."""
        if self._stamps is None:
            raise RuntimeError("GraphedRootNet was not built with time_unprojection=True")
        torch.cuda.synchronize(self._dev)
        t = self._stamps.cpu().tolist()
        return (t[1] - t[0]) / 100.0, (t[2] - t[1]) / 100.0

    def _stage(self, meta):
        """camera table of the NEXT fetch launch -> its ring slot (waits only if the GPU is RING launches behind)"""
        slot = self._launches % self.RING
        self._events[slot].synchronize()
        self._ring[slot].copy_(torch.from_numpy(pack_cameras(meta, self._batch, self.net.project_layer.img_size, self._flip)))
        self._launches += 1

    def __call__(self, meta=None):
        """one step: pack the (possibly new) camera table into the ring, replay"""
        self._stage(self._meta if meta is None else meta)
        k = (self._launches - 1) % self.copies
        self._graphs[k].replay()
        self._events[(self._launches - 1) % self.RING].record(torch.cuda.current_stream(self._dev))
        self.out = self._outs[k]
        return self.out


