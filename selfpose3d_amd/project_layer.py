"""Drop-in replacement for the reference's ``models.project_layer.ProjectLayer``.

Same constructor and call signature as /root/reference/lib/models/project_layer.py:15-106

    ProjectLayer(cfg)
    forward(heatmaps, meta, grid_size, grid_center, cube_size, flip_xcoords=None) -> (cubes, grids)

but the batch x view Python loop of ~90 tiny kernels per (sample, view) is replaced by one
host-side camera-table pack (cached while ``meta`` is unchanged) and one or two launches of
the hand-written gfx950 kernels behind the C ABI in include/sp3d.h.  There is no CPU path:
tensors must live on the GPU and libsp3d.so must be built, otherwise this raises.
"""
from __future__ import annotations

import contextlib
import os
from typing import Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .camera_pack import meta_cache_key, pack_cameras


# Re-tiled heat-maps are shared by every projection of one forward pass (1 coarse + up to
# MAX_PEOPLE_NUM fine calls read the same maps).  An entry is valid only while the very same
# tensor OBJECTS are alive and unmodified (weak references + version counters): a freed tensor
# whose address the allocator hands to the next batch can therefore never produce a false hit.
_PACK_CACHE: "list[tuple]" = []


def packed_heatmaps(hms: Sequence[torch.Tensor], jp: int, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    import weakref
    if any(h.is_inference() for h in hms):      # inference tensors carry no version counter: nothing to key on
        return _lib.pack_heatmaps([x.detach() for x in hms], jp=jp, out_dtype=dtype)
    for refs, versions, cjp, packed in _PACK_CACHE:
        if cjp == (jp, dtype) and len(refs) == len(hms) and \
                all(r() is h and v == h._version for r, v, h in zip(refs, versions, hms)):
            return packed
    packed = _lib.pack_heatmaps([x.detach() for x in hms], jp=jp, out_dtype=dtype)
    _PACK_CACHE.append((tuple(weakref.ref(h) for h in hms), tuple(h._version for h in hms), (jp, dtype), packed))
    while len(_PACK_CACHE) > 2:
        _PACK_CACHE.pop(0)
    return packed


def clear_pack_cache():
    _PACK_CACHE.clear()


def nhwc_heatmap_views(packed: torch.Tensor, num_joints: int) -> "list[torch.Tensor]":
    """``packed`` (V,B,h,w,Jp), channels >= num_joints ZERO  ->  list[V] of (B,J,h,w) tensors that are strided views
    of it (no copy).  This is how a producer that already works channels-last (``PoseResNet.forward_views``: the
    1x1 head emits 16 channels, the 16th filter being zero) hands its heat-maps over: every consumer sees the
    reference's ``list[V] of (B,J,h,w)`` and ``ProjectLayer`` recognises the views and skips its re-tiling pass -
    the kernel reads the producer's buffer directly (``SP3D_LAYOUT_NHWC`` takes per-view pointers)."""
    V, B, h, w, Jp = packed.shape
    if not packed.is_contiguous() or Jp != ProjectLayer.jp_for(num_joints):
        raise _lib.Sp3dError(f"nhwc_heatmap_views: need a contiguous (V,B,h,w,{ProjectLayer.jp_for(num_joints)}) buffer")
    views = []
    for c in range(V):
        t = packed[c].permute(0, 3, 1, 2)[:, :num_joints]
        t._sp3d_packed = (packed, c)
        views.append(t)
    return views


def _packed_source(hms: Sequence[torch.Tensor], jp: int, dtype: torch.dtype):
    """the (V,B,h,w,jp) buffer the heat-maps are views of (see nhwc_heatmap_views), or None"""
    tag = getattr(hms[0], "_sp3d_packed", None)
    if tag is None:
        return None
    base = tag[0]
    if base.dtype != dtype or base.shape[-1] != jp or base.shape[0] != len(hms):
        return None
    for c, h in enumerate(hms):
        t = getattr(h, "_sp3d_packed", None)
        if t is None or t[0] is not base or t[1] != c:
            return None
    return base


class _UnprojectFn(torch.autograd.Function):
    """autograd seam: gradient flows to the heat-maps only."""

    @staticmethod
    def forward(ctx, layer, cam, centers, valid, grid_size, cube_size, want_grids, mode, pad_channels, channels_last,
                sample_of, out, *heatmaps):
        _, J, h, w = heatmaps[0].shape
        B = int(centers.shape[0])                 # number of output cubes (== batch unless `sample_of` is given)
        hms = [x.detach() for x in heatmaps]
        io = layer.io_dtype
        source = _packed_source(heatmaps, layer.jp_for(J), io) if mode == "nhwc" else None
        if io == torch.float32 and source is None:
            hms = [x if (x.is_contiguous() and x.dtype == torch.float32) else x.contiguous().float() for x in hms]
        if mode == "nhwc":
            if source is not None:
                packed = source.detach()          # producer already emits (V,B,h,w,jp): no re-tiling pass
            elif layer.cache_packs:
                packed = packed_heatmaps(heatmaps, layer.jp_for(J), io)
            else:
                packed = _lib.pack_heatmaps(hms, jp=layer.jp_for(J), out_dtype=io)
            jp = packed.shape[-1]
            views = [packed[c] for c in range(len(hms))]
            # when a gradient will be asked for, let the kernel also emit the clamp pass mask: the backward
            # then runs the line-coalesced scatter without re-reading any heat-map
            need_grad = io == torch.float32 and any(ctx.needs_input_grad[12:]) and w >= 2 and h >= 2
            X, Y, Z = cube_size
            mask = torch.empty((B, X * Y * Z), dtype=torch.int16, device=cam.device) if need_grad else None
            # pad_channels: run the kernel over all jp channels - the padded ones are zero in `packed`,
            # so the extra output channels are exact zeros at no extra cost
            cubes, grids = _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, jp, cam, centers, valid, B,
                                              jp if pad_channels else J, h, w, cube_size, grid_size, layer.img_size,
                                              want_grids, channels_last=channels_last, sample_of=sample_of,
                                              out_dtype=io, pass_mask=mask, out=None if need_grad else out)
            ctx.packed_bwd = (mask, jp, int(heatmaps[0].shape[0]), len(hms), J, h, w) if need_grad else None
        else:
            ctx.packed_bwd = None
            cubes, grids = _lib.unproject_fwd(hms, _lib.LAYOUT_PLANAR, 0, cam, centers, valid, B, J, h, w, cube_size,
                                              grid_size, layer.img_size, want_grids, sample_of=sample_of)
        # float64 callers (the mixed-precision pins of tests/test_gpu_reference_pins_r3.py: a float64 model around THIS fp32
        # path): the kernels compute and store fp32, the results travel on in the caller's type
        ctx.wide = heatmaps[0].dtype == torch.float64
        if ctx.wide:
            cubes = cubes.double()
            grids = grids.double() if grids is not None else None
        ctx.layer = layer
        ctx.geom = (tuple(grid_size), tuple(cube_size))
        ctx.sample_of = sample_of
        if ctx.packed_bwd is not None:
            ctx.save_for_backward(cam, centers, valid)        # the packed backward reads no heat-map
        else:
            ctx.save_for_backward(cam, centers, valid, *hms)
        if grids is None:
            grids = torch.empty(0, device=cubes.device)
        ctx.mark_non_differentiable(grids)
        return cubes, grids

    @staticmethod
    def backward(ctx, grad_cubes, _grad_grids):
        cam, centers, valid, *hms = ctx.saved_tensors
        grid_size, cube_size = ctx.geom
        if ctx.wide:
            grad_cubes = grad_cubes.float()
            wide = lambda gs: tuple(x.double() for x in gs)
        else:
            wide = tuple
        if ctx.packed_bwd is not None:
            mask, jp, batch, nv, J, h, w = ctx.packed_bwd
            grads = _lib.unproject_bwd_packed(cam, centers, valid, grad_cubes, mask, batch, nv, J, jp, h, w, cube_size,
                                              grid_size, ctx.layer.img_size, sample_of=ctx.sample_of,
                                              deterministic=ctx.layer.deterministic_backward)
            return (None,) * 12 + wide(grads)
        grads = _lib.unproject_bwd(hms, cam, centers, valid, grad_cubes, cube_size, grid_size, ctx.layer.img_size,
                                   sample_of=ctx.sample_of)
        return (None,) * 12 + wide(grads)


class ProjectLayer(nn.Module):
    """See module docstring.  ``mode``: "nhwc" (re-tile + fast kernel, J <= 16), "planar"
    (direct kernel on the reference layout) or "auto"."""

    def __init__(self, cfg, mode: str = "auto", io_dtype: torch.dtype = torch.float32):
        super().__init__()
        # storage type of the re-tiled heat-maps and of the cubes: fp32 (reference) or bf16 ("mixed bf16",
        # BASELINE configs[4]); projection / interpolation / fusion arithmetic is fp32 either way
        self.io_dtype = io_dtype
        self.img_size = [int(v) for v in cfg.NETWORK.IMAGE_SIZE]        # project_layer.py:19
        self.heatmap_size = [int(v) for v in cfg.NETWORK.HEATMAP_SIZE]  # project_layer.py:20
        self.mode = mode
        self.cache_packs = True       # share the re-tiled heat-maps between the projections of one forward
        self._cam_key = None
        self._cam_dev = None
        self._static_cam = None       # see static_camera_table()
        # gradient scatter in 64-bit fixed point (bit-identical run to run) instead of fp32 atomics; SP3D_BWD_DETERMINISTIC=1
        self.deterministic_backward = os.environ.get("SP3D_BWD_DETERMINISTIC", "0") not in ("", "0")

    @contextlib.contextmanager
    def static_camera_table(self, table: torch.Tensor):
        """Inside the context every call uses ``table`` (a device tensor the caller refreshes itself) and re-tiles
        the heat-maps on every call: what HIP-graph capture of a forward needs (graphs.py).  Everything is restored
        on exit, so eager calls after a capture behave as before."""
        _lib._require_cam(table, "static camera table")
        prev = (self._static_cam, self.cache_packs)
        self._static_cam, self.cache_packs = table, False
        try:
            yield self
        finally:
            self._static_cam, self.cache_packs = prev

    @staticmethod
    def jp_for(J: int) -> int:
        return 4 if J <= 4 else (8 if J <= 8 else (12 if J <= 12 else 16))

    # -- host side -----------------------------------------------------------------------
    def camera_table(self, meta: Sequence[dict], batch: int, flip_xcoords, device) -> torch.Tensor:
        """(B,V,64) fp32 table on `device`; rebuilt only when `meta` / flip change.  The upload is one
        asynchronous copy from a small ring of pinned staging buffers (no host<->GPU synchronisation,
        unlike the ~6 blocking transfers per (sample, view) of the reference, transforms.py:67-72)."""
        if self._static_cam is not None:
            return self._static_cam
        key = (meta_cache_key(meta, flip_xcoords, self.img_size), batch, str(device))
        if key != self._cam_key:
            tab = torch.from_numpy(pack_cameras(meta, batch, self.img_size, flip_xcoords))
            if device.type == "cuda":
                ring = self.__dict__.setdefault("_cam_ring", [])
                slot = self.__dict__.get("_cam_slot", 0)
                if len(ring) < 4 or ring[slot][0].shape != tab.shape:
                    entry = [torch.empty_like(tab).pin_memory(), torch.cuda.Event()]
                    if len(ring) < 4:
                        ring.append(entry)
                        slot = len(ring) - 1
                    else:
                        ring[slot] = entry
                else:
                    ring[slot][1].synchronize()          # the copy that last used this staging buffer is long done
                pinned, ev = ring[slot]
                pinned.copy_(tab)
                dev_tab = torch.empty(tab.shape, dtype=torch.float32, device=device)
                dev_tab.copy_(pinned, non_blocking=True)
                ev.record(torch.cuda.current_stream(device))
                self.__dict__["_cam_slot"] = (slot + 1) % 4
                self._cam_dev = dev_tab
            else:
                self._cam_dev = tab.to(device)
            self._cam_key = key
        return self._cam_dev

    _CONST_CENTERS: "dict[tuple, tuple]" = {}

    @staticmethod
    def centers_valid(grid_center, batch: int, device):
        """(centers (B,3) fp32, valid (B) uint8) following project_layer.py:54,58-61.  A host-side list
        centre (the shared coarse-grid centre from the config) is uploaded once and reused - no per-call
        host->device copy, which also keeps the call HIP-graph capturable."""
        if isinstance(grid_center, torch.Tensor):
            gc = grid_center.to(device=device, dtype=torch.float32)
            if gc.dim() == 1:
                gc = gc[None]
        else:
            arr = np.asarray(grid_center, dtype=np.float32)
            key = (arr.tobytes(), arr.shape, int(batch), str(device))
            hit = ProjectLayer._CONST_CENTERS.get(key)
            if hit is not None:
                return hit
            gc = torch.as_tensor(arr, device=device)
        rows, cols = gc.shape
        if cols == 3:                       # `len(grid_center[0]) == 3`: always valid
            valid = torch.ones(batch, dtype=torch.uint8, device=device)
        else:
            flag = gc[:, 3] >= 0
            valid = (flag if rows == batch else flag[:1].expand(batch)).to(torch.uint8)
        centers = gc[:, :3]
        if rows == 1 and batch > 1:         # `len(grid_center) == 1`: shared centre
            centers = centers.expand(batch, 3)
        out = (centers.contiguous(), valid.contiguous())
        if not isinstance(grid_center, torch.Tensor):
            if len(ProjectLayer._CONST_CENTERS) > 64:
                ProjectLayer._CONST_CENTERS.clear()
            ProjectLayer._CONST_CENTERS[key] = out
        return out

    # -- reference API -------------------------------------------------------------------
    def get_voxel(self, heatmaps, meta, grid_size, grid_center, cube_size, flip_xcoords=None, want_grids=True,
                  pad_channels=False, channels_last=False, sample_of=None, out=None):
        """Reference semantics (project_layer.py:42-102).  Extras for in-repo callers only:
        ``want_grids=False`` skips the (B,N,3) grid output, ``pad_channels`` returns
        ceil4(J) channels (zeros beyond J) and ``channels_last`` returns torch.channels_last_3d
        strides - both let MIOpen's 3D convolutions run their fast paths without a copy;
        ``sample_of`` (int (P,)) with ``grid_center`` (P,5|3): P cubes, cube p read from sample
        sample_of[p] (all person proposals of a batch in one launch); ``out``: a (P,J,X,Y,Z) view of a larger buffer
        (z contiguous) that receives the planar result directly - no grids, inference only."""
        device = heatmaps[0].device
        if not heatmaps[0].is_cuda:
            raise _lib.Sp3dError("ProjectLayer: heat-maps must be on the GPU (no CPU fallback)")
        B, J, h, w = heatmaps[0].shape
        if [w, h] != self.heatmap_size:
            # the reference samples with cfg HEATMAP_SIZE (project_layer.py:50) whatever the tensor says
            raise _lib.Sp3dError(f"heat-map tensor is {w}x{h} but cfg.NETWORK.HEATMAP_SIZE is {self.heatmap_size}")
        cam = self.camera_table(meta, B, flip_xcoords, device)
        if sample_of is not None:
            sample_of = sample_of.to(device=device, dtype=torch.int32).contiguous()
            centers, valid = self.centers_valid(grid_center, int(sample_of.shape[0]), device)
        else:
            centers, valid = self.centers_valid(grid_center, B, device)
        if isinstance(cube_size, int):
            cube_size = [cube_size] * 3
        if isinstance(grid_size, (int, float)):
            grid_size = [grid_size] * 3
        mode = self.mode
        if mode == "auto":
            mode = "nhwc" if (J <= 16 and w >= 2 and h >= 2) else "planar"
        if self.io_dtype != torch.float32 and (mode != "nhwc" or self.jp_for(J) != 16):
            raise _lib.Sp3dError("bf16 storage needs the NHWC path with 13..16 joints")
        if mode != "nhwc":
            pad_channels = channels_last = False
        if channels_last and not pad_channels and (J & 3):
            channels_last = False
        if out is not None and (mode != "nhwc" or want_grids or pad_channels or channels_last):
            raise _lib.Sp3dError("get_voxel(out=...): planar NHWC-path result without grids only")
        cubes, grids = _UnprojectFn.apply(self, cam, centers, valid, [float(v) for v in grid_size],
                                          [int(v) for v in cube_size], bool(want_grids), mode, bool(pad_channels),
                                          bool(channels_last), sample_of, out, *heatmaps)
        return cubes, (grids if want_grids else None)

    def get_voxel_zspectrum(self, heatmaps, meta, grid_size, grid_center, cube_size, SZ: int, flip_xcoords=None):
        """Inference only, root grid only (round 6): ``get_voxel`` FUSED with the z pass of the V2V net's frequency-domain
        opening conv - the cubes are never written.  Returns the (B, J, SZ//2+1, X/4, Y/4, 16) complex64 z-spectrum of the
        cubes ``get_voxel`` would return (bit-identical to ``_lib.zdft_fwd_cl`` of its channels-last result), in the
        4 x 4-tiled layout ``_lib.cfft2d_88_tiled`` reads.  Same reference semantics (project_layer.py:42-102)."""
        if torch.is_grad_enabled() and any(h.requires_grad for h in heatmaps):
            raise _lib.Sp3dError("get_voxel_zspectrum is an inference path (no autograd); use get_voxel")
        device = heatmaps[0].device
        if not heatmaps[0].is_cuda:
            raise _lib.Sp3dError("ProjectLayer: heat-maps must be on the GPU (no CPU fallback)")
        B, J, h, w = heatmaps[0].shape
        if [w, h] != self.heatmap_size:
            raise _lib.Sp3dError(f"heat-map tensor is {w}x{h} but cfg.NETWORK.HEATMAP_SIZE is {self.heatmap_size}")
        if self.io_dtype != torch.float32 or self.jp_for(J) != 16 or self.mode not in ("auto", "nhwc") or w < 2 or h < 2:
            raise _lib.Sp3dError("get_voxel_zspectrum: fp32 NHWC path with 13..16 joints only")
        cam = self.camera_table(meta, B, flip_xcoords, device)
        centers, valid = self.centers_valid(grid_center, B, device)
        source = _packed_source(heatmaps, 16, torch.float32)
        if source is not None:
            packed = source.detach()
        elif self.cache_packs:
            packed = packed_heatmaps(heatmaps, 16, torch.float32)
        else:
            packed = _lib.pack_heatmaps([x.detach() for x in heatmaps], jp=16)
        return _lib.unproject_fwd_zdft([packed[c] for c in range(len(heatmaps))], 16, cam, centers, valid, B, J, h, w,
                                       [int(v) for v in cube_size], [float(v) for v in grid_size], self.img_size, int(SZ))

    def forward(self, heatmaps, meta, grid_size, grid_center, cube_size, flip_xcoords=None):
        return self.get_voxel(heatmaps, meta, grid_size, grid_center, cube_size, flip_xcoords=flip_xcoords)
