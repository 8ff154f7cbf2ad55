"""ctypes/numpy front-end of the CPU oracle (oracle/sp3d_oracle.c).

TEST INFRASTRUCTURE ONLY - see the header of sp3d_oracle.c.  Import this from tests/,
``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg, never from the
``selfpose3d_amd`` package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsp3d_oracle.so")
_lib = None

CAM_STRIDE = 64


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "sp3d_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.sp3d_oracle_unproject_fwd.restype = C.c_int
        _lib.sp3d_oracle_unproject_bwd.restype = C.c_int
        _lib.sp3d_oracle_nms_topk.restype = C.c_int
    return _lib


def set_threads(n: int) -> int:
    """OpenMP threads of the oracle's voxel loops (results do not depend on it); returns the previous setting"""
    return int(lib().sp3d_oracle_set_threads(C.c_int(int(n))))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def linspace(L: float, n: int) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().sp3d_oracle_linspace(C.c_float(L), C.c_int(n), _fp(out))
    return out


def project_points(cam: np.ndarray, pts: np.ndarray) -> np.ndarray:
    cam = np.ascontiguousarray(cam, np.float32)
    pts = np.ascontiguousarray(pts, np.float32)
    out = np.empty((pts.shape[0], 2), np.float32)
    lib().sp3d_oracle_project_points(_fp(cam), _fp(pts), C.c_int(pts.shape[0]), _fp(out))
    return out


def affine(center, scale, rot, out_size) -> np.ndarray:
    center = np.ascontiguousarray(center, np.float64)
    scale = np.ascontiguousarray(scale, np.float32)
    out = np.empty(6, np.float64)
    lib().sp3d_oracle_affine(center.ctypes.data_as(C.POINTER(C.c_double)), _fp(scale), C.c_double(float(rot)),
                             C.c_int(int(out_size[0])), C.c_int(int(out_size[1])),
                             out.ctypes.data_as(C.POINTER(C.c_double)))
    return out.reshape(2, 3)


def _view_ptrs(hms, ctype):
    arr = (C.POINTER(ctype) * len(hms))()
    for i, h in enumerate(hms):
        arr[i] = h.ctypes.data_as(C.POINTER(ctype))
    return arr


def unproject_fwd(hms, cam, centers, valid, grid_size, cube_size, img_size, want_grids=True, want_bound_frac=False):
    """hms: list[V] of (B,J,h,w) float32 arrays; cam (B,V,32); centers (B,3); valid (B) uint8."""
    hms = [np.ascontiguousarray(h, np.float32) for h in hms]
    B, J, h, w = hms[0].shape
    V = len(hms)
    X, Y, Z = [int(c) for c in cube_size]
    N = X * Y * Z
    cam = np.ascontiguousarray(cam, np.float32).reshape(B, V, CAM_STRIDE)
    centers = np.ascontiguousarray(centers, np.float32).reshape(B, 3)
    valid = np.ascontiguousarray(valid, np.uint8).reshape(B)
    gs = np.ascontiguousarray(grid_size, np.float32)
    cubes = np.empty((B, J, X, Y, Z), np.float32)
    grids = np.empty((B, N, 3), np.float32) if want_grids else None
    bf = C.c_double(0.0)
    rc = lib().sp3d_oracle_unproject_fwd(
        _view_ptrs(hms, C.c_float), _fp(cam), _fp(centers), valid.ctypes.data_as(C.POINTER(C.c_uint8)),
        _fp(cubes), _fp(grids) if want_grids else None, B, V, J, h, w, X, Y, Z, _fp(gs),
        int(img_size[0]), int(img_size[1]), C.byref(bf))
    assert rc == 0
    if want_bound_frac:
        return cubes, grids, bf.value
    return cubes, grids


def unproject_bwd(hms, cam, centers, valid, grad_cubes, grid_size, cube_size, img_size):
    hms = [np.ascontiguousarray(h, np.float32) for h in hms]
    B, J, h, w = hms[0].shape
    V = len(hms)
    X, Y, Z = [int(c) for c in cube_size]
    cam = np.ascontiguousarray(cam, np.float32).reshape(B, V, CAM_STRIDE)
    centers = np.ascontiguousarray(centers, np.float32).reshape(B, 3)
    valid = np.ascontiguousarray(valid, np.uint8).reshape(B)
    gs = np.ascontiguousarray(grid_size, np.float32)
    gc = np.ascontiguousarray(grad_cubes, np.float32)
    grads = [np.zeros((B, J, h, w), np.float64) for _ in range(V)]
    rc = lib().sp3d_oracle_unproject_bwd(
        _view_ptrs(hms, C.c_float), _fp(cam), _fp(centers), valid.ctypes.data_as(C.POINTER(C.c_uint8)),
        _fp(gc), _view_ptrs(grads, C.c_double), B, V, J, h, w, X, Y, Z, _fp(gs), int(img_size[0]), int(img_size[1]))
    assert rc == 0
    return grads


def nms_topk(cube: np.ndarray, k: int):
    cube = np.ascontiguousarray(cube, np.float32)
    B, X, Y, Z = cube.shape
    vals = np.empty((B, k), np.float32)
    idx = np.empty((B, k, 3), np.int64)
    rc = lib().sp3d_oracle_nms_topk(_fp(cube), B, X, Y, Z, k, _fp(vals), idx.ctypes.data_as(C.POINTER(C.c_int64)))
    assert rc == 0
    return vals, idx


def soft_argmax(x: np.ndarray, grids: np.ndarray, beta: float) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    Bv, J = x.shape[:2]
    N = int(np.prod(x.shape[2:]))
    grids = np.ascontiguousarray(grids, np.float32).reshape(Bv, N, 3)
    out = np.empty((Bv, J, 3), np.float32)
    lib().sp3d_oracle_soft_argmax(_fp(x), _fp(grids), C.c_int(Bv), C.c_int(J), C.c_size_t(N), C.c_float(beta), _fp(out))
    return out
