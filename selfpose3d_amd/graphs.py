"""HIP-graph execution of the root-localisation forward (fixed shapes): heat-map re-tiling, unprojection, V2V convs +
fused epilogues, NMS/top-k are captured once and replayed, so ~65 kernel launches cost one graph launch.  Per step the
host packs the batch's camera table (calibration / crop data) into a pinned staging buffer and enqueues ONE async copy
into the static device table the captured kernels read, then replays."""
from __future__ import annotations

from typing import List, Sequence

import torch

from .camera_pack import pack_cameras


class GraphedRootNet:
    """The wrapped net is left exactly as it was: the static camera table is only in force while capturing
    (``ProjectLayer.static_camera_table``), so eager calls on the same net keep working with their own ``meta``."""

    RING = 3      # pinned staging buffers: step t+1 is packed while the copy of step t may still be in flight

    def __init__(self, net, heatmaps: Sequence[torch.Tensor], meta: Sequence[dict], flip_xcoords=None, warmup: int = 3):
        self.net = net
        self.static_hms: List[torch.Tensor] = list(heatmaps)       # the caller writes new heat-maps into these
        dev = heatmaps[0].device
        pl = net.project_layer
        B = heatmaps[0].shape[0]
        self._batch, self._flip, self._meta, self._dev = B, flip_xcoords, meta, dev
        tab = torch.from_numpy(pack_cameras(meta, B, pl.img_size, flip_xcoords))
        self._ring = [[torch.empty_like(tab).pin_memory(), torch.cuda.Event()] for _ in range(self.RING)]
        self._slot = 0
        self.cam_dev = tab.to(dev)
        with pl.static_camera_table(self.cam_dev):
            stream = torch.cuda.Stream(dev)
            stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(stream), torch.no_grad():
                for _ in range(warmup):
                    net(self.static_hms, meta, flip_xcoords)
            torch.cuda.current_stream(dev).wait_stream(stream)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph), torch.no_grad():
                self.out = net(self.static_hms, meta, flip_xcoords)
        # the captured kernels read the folded inference plan's tensors by address: keep them alive even if the net
        # drops its plan later (train() / load_state_dict / invalidate_plan)
        self._plan = getattr(getattr(net, "v2v_net", None), "_plan", None)
        self._plan_tensors = dict(self._plan.t) if self._plan is not None else None     # incl. the padded FFT buffers

    def __call__(self, meta=None):
        """one step: pack the (possibly new) camera table, upload it asynchronously, replay"""
        m = self._meta if meta is None else meta
        pinned, ev = self._ring[self._slot]
        ev.synchronize()                       # the copy that last read this staging buffer (RING steps ago) is done
        pinned.copy_(torch.from_numpy(pack_cameras(m, self._batch, self.net.project_layer.img_size, self._flip)))
        self.cam_dev.copy_(pinned, non_blocking=True)
        ev.record(torch.cuda.current_stream(self._dev))
        self._slot = (self._slot + 1) % self.RING
        self.graph.replay()
        return self.out
