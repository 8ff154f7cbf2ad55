"""Grouped training-mode BatchNorm on channels-last tensors (``sp3d_gbn_forward`` / ``sp3d_gbn_backward``, include/sp3d.h).

The reference trains its pose net one candidate slot per call (/root/reference/lib/models/multi_person_posenet.py:84-88,
multi_person_posenet_ssv.py:354-383): every ``BatchNorm3d`` of ``V2VNet`` (/root/reference/lib/models/v2v_net.py:14,28,31,
38,64) then normalises with the statistics of THAT slot's valid cubes and updates its running statistics once per call.
``GroupedBatchNorm3d`` is ``nn.BatchNorm3d`` (same parameters, buffers, state_dict keys) that, while a ``GroupSpec`` is
attached in train mode, treats a batch as G groups of samples - statistics, normalisation, backward and the G sequential
running-statistics updates per group - so all slots run through the net as ONE batch with the loop's results.  The same
class with ``group_of[n] = n % V`` is the per-camera BatchNorm of a backbone that runs all V views at once
(``GroupedBatchNorm2d``).  Without a spec, or in eval mode, both are their parent class.

There is no CPU path: tensors must be on the GPU and libsp3d.so built, otherwise this raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import _lib


class GroupSpec:
    """which sample belongs to which BatchNorm group.  ``sizes``: samples per group, host ints, in running-statistics
    order; groups are contiguous runs unless ``group_of`` (host ints, one per sample) is given.  ``n_update``: the first
    n_update groups update the running statistics (padding groups behind them do not)."""

    def __init__(self, sizes: Sequence[int], device, group_of: Optional[Sequence[int]] = None, n_update: Optional[int] = None):
        self.sizes = [int(s) for s in sizes]
        self.G = len(self.sizes)
        self.N = sum(self.sizes)
        self.n_update = self.G if n_update is None else int(n_update)
        if group_of is None:
            group_of = [g for g, s in enumerate(self.sizes) for _ in range(s)]
        if len(group_of) != self.N:
            raise ValueError("GroupSpec: group_of must name a group for each of the sum(sizes) samples")
        counted = [0] * self.G
        for g in group_of:
            counted[int(g)] += 1
        if counted != self.sizes:
            raise ValueError(f"GroupSpec: group_of holds {counted} samples per group, sizes says {self.sizes}")
        self.group_of = torch.tensor(list(group_of), dtype=torch.int32, device=device)
        self.group_samples = torch.tensor(self.sizes, dtype=torch.int32, device=device)
        self._ws = {}

    def workspace(self, C_: int, device) -> torch.Tensor:
        """float64 (replicas, G, C, 2) accumulator, zero-filled ONCE per (spec, C, stream): every forward / backward call
        leaves it zero-filled again (include/sp3d.h), so all layers of that width share it in stream order"""
        key = (int(C_), str(device), torch.cuda.current_stream(device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None:
            n = int(_lib.load().sp3d_gbn_workspace_bytes(self.G, int(C_))) // 8
            ws = self._ws[key] = torch.zeros(n, dtype=torch.float64, device=device)
        return ws


_SPECS: "dict[tuple, GroupSpec]" = {}


def group_spec(sizes: Sequence[int], device, group_of: Optional[Sequence[int]] = None, n_update: Optional[int] = None) -> GroupSpec:
    """a cached GroupSpec: a training loop presents the same few groupings step after step (5 views x B images; the slots of
    a scene), and a spec owns device tensors and zero-filled workspaces that need not be rebuilt per step"""
    key = (tuple(int(s) for s in sizes), None if group_of is None else tuple(int(g) for g in group_of), n_update, str(device))
    spec = _SPECS.get(key)
    if spec is None:
        if len(_SPECS) >= 64:
            _SPECS.pop(next(iter(_SPECS)))
        spec = _SPECS[key] = GroupSpec(sizes, device, group_of=group_of, n_update=n_update)
    return spec


def _as_rows(x: torch.Tensor):
    """(N, C, *spatial) with channels-last strides -> (tensor whose memory is (N, S, C) contiguous, N, S, C)"""
    if x.dim() not in (4, 5):
        raise _lib.Sp3dError("grouped BatchNorm: (N,C,H,W) or (N,C,D,H,W) input")
    fmt = torch.channels_last if x.dim() == 4 else torch.channels_last_3d
    if not x.is_contiguous(memory_format=fmt):
        x = x.contiguous(memory_format=fmt)
    N, C_ = int(x.shape[0]), int(x.shape[1])
    S = 1
    for d in x.shape[2:]:
        S *= int(d)
    return x, N, S, C_


_DT = {torch.float32: 0, torch.float64: 1}
FUSE_TAIL = __import__("os").environ.get("SP3D_GBN_FUSE_TAIL", "1") != "0"


class _GroupedBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, spec: GroupSpec, eps: float, momentum: float, relu: bool,
                residual=None):
        lib = _lib.load()
        _lib._require_cuda(x, "x")
        if x.dtype not in _DT:
            raise _lib.Sp3dError(f"grouped BatchNorm: float32 or float64 input, got {x.dtype}")
        x, N, S, C_ = _as_rows(x)
        if N != spec.N:
            raise _lib.Sp3dError(f"grouped BatchNorm: batch of {N} samples, GroupSpec describes {spec.N}")
        mode = 2 if residual is not None else (1 if relu else 0)
        if residual is not None:
            if not relu or residual.shape != x.shape or residual.dtype != x.dtype:
                raise _lib.Sp3dError("grouped BatchNorm: a residual (same shape and dtype as x) goes with relu=True")
            residual = _as_rows(residual.detach())[0]
        dev = x.device
        y = torch.empty_like(x)                                   # keeps the channels-last strides
        stats = torch.empty((4, spec.G, C_), dtype=x.dtype, device=dev)      # mean, invstd, scale, shift
        w = weight.detach().to(x.dtype).contiguous() if weight is not None else None
        b = bias.detach().to(x.dtype).contiguous() if bias is not None else None
        rm = running_mean if (running_mean is not None and running_mean.dtype == x.dtype) else None
        rv = running_var if (running_var is not None and running_var.dtype == x.dtype) else None
        if (running_mean is not None) != (rm is not None):
            raise _lib.Sp3dError("grouped BatchNorm: running statistics must have the input's dtype")
        ws = spec.workspace(C_, dev)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        rc = lib.sp3d_gbn_forward(p(x), p(y), _DT[x.dtype], p(spec.group_of), p(spec.group_samples), N, C.c_int64(S), C_,
                                  spec.G, spec.n_update, p(w), p(b), p(rm), p(rv), C.c_double(eps), C.c_double(momentum),
                                  mode, p(residual), p(stats[0]), p(stats[1]), p(stats[2]), p(stats[3]), p(ws),
                                  C.c_void_p(_lib._stream(dev)))
        _lib.check(rc, "sp3d_gbn_forward")
        if mode == 2:
            ctx.save_for_backward(x, w, stats, y)                 # the block's output is what the next layer keeps anyway
        else:
            ctx.save_for_backward(x, w, stats)
        ctx.spec, ctx.mode, ctx.geom = spec, mode, (N, S, C_)
        ctx.has_affine = (weight is not None, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, stats = ctx.saved_tensors[:3]
        yout = ctx.saved_tensors[3] if ctx.mode == 2 else None
        spec, (N, S, C_) = ctx.spec, ctx.geom
        fmt = torch.channels_last if x.dim() == 4 else torch.channels_last_3d
        dy = dy.contiguous(memory_format=fmt)
        dev = x.device
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.mode == 2 else None
        gw = torch.empty(C_, dtype=x.dtype, device=dev)
        gb = torch.empty(C_, dtype=x.dtype, device=dev)
        k123 = torch.empty((3, spec.G, C_), dtype=x.dtype, device=dev)
        ws = spec.workspace(C_, dev)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        rc = lib.sp3d_gbn_backward(p(x), p(dy), p(dx), _DT[x.dtype], p(spec.group_of), p(spec.group_samples), N, C.c_int64(S),
                                   C_, spec.G, p(w), p(stats[0]), p(stats[1]), p(stats[2]), p(stats[3]), ctx.mode, p(yout),
                                   p(dres), p(gw), p(gb), p(k123), p(ws), C.c_void_p(_lib._stream(dev)))
        _lib.check(rc, "sp3d_gbn_backward")
        return (dx, (gw if ctx.has_affine[0] else None), (gb if ctx.has_affine[1] else None), None, None, None, None, None, None,
                dres)


def _declare(lib):
    if getattr(lib, "_gbn_declared", False):
        return
    I, P, L, D = C.c_int, C.c_void_p, C.c_int64, C.c_double
    lib.sp3d_gbn_workspace_bytes.restype = L
    lib.sp3d_gbn_workspace_bytes.argtypes = [I, I]
    lib.sp3d_gbn_forward.restype = I
    lib.sp3d_gbn_forward.argtypes = [P, P, I, P, P, I, L, I, I, I, P, P, P, P, D, D, I, P, P, P, P, P, P, P]
    lib.sp3d_gbn_backward.restype = I
    lib.sp3d_gbn_backward.argtypes = [P, P, P, I, P, P, I, L, I, I, P, P, P, P, P, I, P, P, P, P, P, P, P]
    lib._gbn_declared = True


class _GroupedMixin:
    """train mode + an attached GroupSpec: the grouped kernels; otherwise the parent BatchNorm"""
    groups: Optional[GroupSpec] = None

    def grouped_forward(self, x, relu: bool = False, residual=None):
        """BatchNorm [+ residual] [+ ReLU]; with a spec attached in train mode ONE pass of the grouped kernels"""
        spec = self.groups
        if residual is not None and not FUSE_TAIL and spec is not None and self.training:      # A/B switch (measurement)
            return torch.relu_(self.grouped_forward(x, False) + residual)
        if spec is None or not self.training:
            y = super().forward(x)
            if residual is not None:
                y = y + residual
            return torch.relu_(y) if relu else y
        if self.momentum is None or not self.track_running_stats:
            raise ValueError("grouped BatchNorm needs a numeric momentum and running statistics (as the reference's layers)")
        _declare(_lib.load())
        y = _GroupedBNFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, spec, float(self.eps),
                               float(self.momentum), bool(relu), residual)
        with torch.no_grad():
            self.num_batches_tracked += spec.n_update
        return y

    def forward(self, x):
        return self.grouped_forward(x, False)


class GroupedBatchNorm3d(_GroupedMixin, nn.BatchNorm3d):
    pass


class GroupedBatchNorm2d(_GroupedMixin, nn.BatchNorm2d):
    pass


@contextlib.contextmanager
def bn_groups(module: nn.Module, spec: Optional[GroupSpec]):
    """attach ``spec`` to every grouped BatchNorm below ``module`` for the duration of the block"""
    mods = [m for m in module.modules() if isinstance(m, _GroupedMixin)]
    prev = [m.groups for m in mods]
    for m in mods:
        m.groups = spec
    try:
        yield
    finally:
        for m, g in zip(mods, prev):
            m.groups = g
