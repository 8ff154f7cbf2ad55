"""ctypes binding of libsp3d.so (C ABI: include/sp3d.h).

There is NO CPU fallback behind this module: if the library is missing or a call fails,
an exception is raised.  torch is used only to obtain device pointers and the current HIP
stream (``torch.cuda.current_stream().cuda_stream``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB_PATH = os.path.join(_HERE, "libsp3d.so")
NOPK_LIB_PATH = os.path.join(_HERE, "libsp3d_nopk.so")       # the flavour without packed-fp32 instructions (shared GPUs)
# SP3D_LIB_PATH: a measurement / sanitizer build of the SAME sources instead of the shipped library (tools/sanitize_host.py)
LIB_PATH = os.environ.get("SP3D_LIB_PATH") or _DEFAULT_LIB_PATH

LAYOUT_PLANAR = 0
LAYOUT_NHWC = 1
OUT_CHANNELS_LAST = 0x100
HM_BF16 = 0x200
OUT_BF16 = 0x400
SCATTER_AUTO, SCATTER_PER_TAP, SCATTER_MERGE = 0, 2, 3      # include/sp3d.h: per-call scatter choice of unproject_bwd_packed
MAX_VIEWS = 16
MAX_TOPK = 32
ABI_VERSION = 3

EXPORTS = [
    "sp3d_abi_version", "sp3d_error_string", "sp3d_pack_heatmaps", "sp3d_unproject_fwd", "sp3d_unproject_bwd",
    "sp3d_nms_topk_workspace_bytes", "sp3d_nms_topk", "sp3d_nms_proposals", "sp3d_soft_argmax", "sp3d_unproject_fwd_indexed",
    "sp3d_unproject_bwd_indexed", "sp3d_unproject_fwd_strided", "sp3d_fetch_ring", "sp3d_maxpool2x_cl", "sp3d_crop_shift_act_cl", "sp3d_rfft3d", "sp3d_irfft3d", "sp3d_cfft2d", "sp3d_cfft2d_ex", "sp3d_zdft_fwd_cl", "sp3d_zdft_inv_cl", "sp3d_unproject_fwd_zdft", "sp3d_cfft2d_88_tiled", "sp3d_freq_contract_ty", "sp3d_soft_argmax_grid", "sp3d_soft_argmax_grid_train", "sp3d_soft_argmax_grid_bwd", "sp3d_channel_shift_act", "sp3d_pack_heatmaps_ex",
    "sp3d_unproject_fwd_train", "sp3d_unproject_bwd_packed", "sp3d_unproject_bwd_packed_det", "sp3d_fixed_to_float", "sp3d_gaussian_target_3d", "sp3d_render_root_heatmaps", "sp3d_freq_contract", "sp3d_freq_contract_ex", "sp3d_wino_input", "sp3d_wino_output", "sp3d_wino_fused", "sp3d_wino_fused_split", "sp3d_wino_fused_split64", "sp3d_conv3_split", "sp3d_camera_finish", "sp3d_upsample2x_scatter", "sp3d_upsample2x_scatter_head", "sp3d_render_joints_fwd", "sp3d_render_joints_bwd", "sp3d_gbn_workspace_bytes", "sp3d_gbn_forward", "sp3d_gbn_backward",
]

_lib = None


class Sp3dError(RuntimeError):
    pass


def shared_gpu() -> bool:
    """SP3D_SHARED_GPU=1 (or, when the variable is unset and NO *_VISIBLE_DEVICES variable narrows what this process sees,
    more local ranks than GPUs): this process is not alone on its GPU (another process, or a second stream of its own, may run
    kernels at the same time).  The conservative library flavour is loaded then, libsp3d_nopk.so - no packed-fp32
    instruction at all.  Background (profiles/r05_mfma_pk_hazard.md): ONE packed-fp32 instruction form is wrong on MI355X
    while another kernel's double-rate matrix instructions run on the same CU; the build removes that form from the default
    flavour too (selfpose3d_amd/pk_src1.py), and tests/test_gpu_shared_gpu.py asserts both flavours on two streams."""
    v = os.environ.get("SP3D_SHARED_GPU")
    if v is not None:
        return v.lower() in ("1", "true", "yes", "on")
    # not told: a launcher that starts more local ranks than there are GPUs (torchrun's LOCAL_WORLD_SIZE) makes ranks share.
    # Only when the device count is the NODE's: a launcher that hands every rank one GPU through HIP_/CUDA_/ROCR_VISIBLE_DEVICES
    # makes device_count() == 1 on a perfectly normal one-process-per-GPU job (round-5 advice) - nothing can be inferred then.
    if any(os.environ.get(k) not in (None, "") for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES",
                                                          "GPU_DEVICE_ORDINAL")):
        return False
    try:
        ranks = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
        if ranks > 1 and torch.cuda.is_available() and ranks > torch.cuda.device_count():
            import warnings
            warnings.warn(f"selfpose3d_amd: {ranks} local ranks on {torch.cuda.device_count()} GPU(s) - ranks share a GPU: loading "
                          "libsp3d_nopk.so (no packed-fp32 instructions; set SP3D_SHARED_GPU=0 to force the default flavour)")
            return True
    except ValueError:
        pass
    return False


def load():
    """dlopen libsp3d.so (building nothing: run `python -m selfpose3d_amd.build` first)."""
    global _lib, LIB_PATH
    if _lib is not None:
        return _lib
    if shared_gpu() and LIB_PATH == _DEFAULT_LIB_PATH:
        LIB_PATH = NOPK_LIB_PATH
    if not os.path.exists(LIB_PATH):
        raise Sp3dError(
            f"{LIB_PATH} not found - the HIP extension is not built. Run `python -m selfpose3d_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback for the unprojection path.")
    lib = C.CDLL(LIB_PATH)
    P, I, F, V = C.c_void_p, C.c_int, C.c_float, C.c_void_p
    lib.sp3d_abi_version.restype = I
    lib.sp3d_error_string.restype = C.c_char_p
    lib.sp3d_error_string.argtypes = [I]
    lib.sp3d_pack_heatmaps.restype = I
    lib.sp3d_pack_heatmaps.argtypes = [P, P, I, I, I, I, I, I, V]
    lib.sp3d_pack_heatmaps_ex.restype = I
    lib.sp3d_pack_heatmaps_ex.argtypes = [P, P, I, I, I, I, I, I, I, I, V]
    lib.sp3d_unproject_fwd.restype = I
    lib.sp3d_unproject_fwd.argtypes = [P, I, I, P, P, P, P, P, I, I, I, I, I, I, I, I, P, I, I, V]
    lib.sp3d_unproject_bwd.restype = I
    lib.sp3d_unproject_bwd.argtypes = [P, P, P, P, P, P, I, I, I, I, I, I, I, I, P, I, I, V]
    lib.sp3d_nms_topk_workspace_bytes.restype = C.c_int64
    lib.sp3d_nms_topk_workspace_bytes.argtypes = [I, I, I, I, I]
    lib.sp3d_nms_topk.restype = I
    lib.sp3d_nms_topk.argtypes = [P, I, I, I, I, I, P, P, P, P, P, P, V]
    lib.sp3d_soft_argmax.restype = I
    lib.sp3d_soft_argmax.argtypes = [P, P, P, I, I, C.c_int64, F, V]
    lib.sp3d_unproject_fwd_indexed.restype = I
    lib.sp3d_unproject_fwd_indexed.argtypes = [P, I, I, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P, I, I, V]
    lib.sp3d_unproject_bwd_indexed.restype = I
    lib.sp3d_unproject_bwd_indexed.argtypes = [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P, I, I, V]
    lib.sp3d_unproject_fwd_train.restype = I
    lib.sp3d_unproject_fwd_train.argtypes = [P, I, I, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P, I, I, V]
    lib.sp3d_unproject_bwd_packed.restype = I
    lib.sp3d_unproject_bwd_packed.argtypes = [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P, I, I, I, V]
    lib.sp3d_upsample2x_scatter.restype = I
    lib.sp3d_upsample2x_scatter.argtypes = [P, P, P, P, C.c_int64, I, I, I, I, V]
    lib.sp3d_render_joints_fwd.restype = I
    lib.sp3d_render_joints_fwd.argtypes = [P, P, I, I, I, I, I, F, P, V]
    lib.sp3d_render_joints_bwd.restype = I
    lib.sp3d_render_joints_bwd.argtypes = [P, P, P, I, I, I, I, I, F, P, V]
    lib.sp3d_wino_fused.restype = I
    lib.sp3d_wino_fused.argtypes = [P, P, P, P, P, I, I, I, I, I, I, I, V]
    lib.sp3d_wino_input.restype = I
    lib.sp3d_wino_input.argtypes = [P, P, I, I, I, I, I, V]
    lib.sp3d_wino_output.restype = I
    lib.sp3d_wino_output.argtypes = [P, P, P, P, I, I, I, I, I, I, V]
    lib.sp3d_freq_contract_ex.restype = I
    lib.sp3d_freq_contract_ex.argtypes = [P, P, P, I, I, I] + [C.c_int64] * 5 + [I, I, V]
    lib.sp3d_freq_contract.restype = I
    lib.sp3d_freq_contract.argtypes = [P, P, P, I, I, I, C.c_int64, V]
    lib.sp3d_gaussian_target_3d.restype = I
    lib.sp3d_gaussian_target_3d.argtypes = [P, I, I, P, P, P, I, I, I, F, P, V]
    lib.sp3d_render_root_heatmaps.restype = I
    lib.sp3d_render_root_heatmaps.argtypes = [P, I, I, P, I, I, I, F, P, V]
    lib.sp3d_soft_argmax_grid.restype = I
    lib.sp3d_soft_argmax_grid.argtypes = [P, P, P, I, I, I, P, I, I, F, V]
    lib.sp3d_channel_shift_act.restype = I
    lib.sp3d_channel_shift_act.argtypes = [P, P, P, I, C.c_int64, I, C.c_int64, I, V]
    lib.sp3d_unproject_fwd_strided.restype = I
    lib.sp3d_unproject_fwd_strided.argtypes = [P, I, I, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P, I, I, V]
    if hasattr(lib, "sp3d_unproject_fwd_variant"):
        lib.sp3d_unproject_fwd_variant.restype = I
        lib.sp3d_unproject_fwd_variant.argtypes = [P, I, P, P, P, P, P, I, I, I, I, I, I, I, I, P, I, I, I, V]
    if lib.sp3d_abi_version() != ABI_VERSION:
        raise Sp3dError(f"libsp3d.so ABI {lib.sp3d_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().sp3d_error_string(rc).decode()
        raise Sp3dError(f"{what} failed: {msg} (code {rc})")


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise Sp3dError(f"{name} must live on the GPU (got {t.device}); the unprojection path has no CPU fallback")


CAM_STRIDE = 64     # floats per (sample, view) record, ABI version 2 (include/sp3d.h: SP3D_CAM_STRIDE)


def _require_cam(cam: torch.Tensor, name: str = "cam"):
    """the kernels read 64-float records (fields 0..29 + the derived block 32..62 written by sp3d_camera_finish /
    camera_pack.finish): a legacy 32-float table, another dtype or a strided view would be read out of bounds or as
    garbage WITHOUT any error from the C ABI (it sees a pointer), so the binding refuses them"""
    _require_cuda(cam, name)
    if cam.dim() < 2 or cam.shape[-1] != CAM_STRIDE or cam.dtype != torch.float32 or not cam.is_contiguous():
        raise Sp3dError(f"{name} must be a contiguous float32 (..., {CAM_STRIDE}) camera table built by camera_pack.pack_cameras "
                        f"(+ camera_pack.finish after edits); got shape {tuple(cam.shape)}, {cam.dtype}, "
                        f"contiguous={cam.is_contiguous()}")


def _ptr_array(tensors: Sequence[torch.Tensor]):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def _f3(vals):
    return (C.c_float * 3)(float(vals[0]), float(vals[1]), float(vals[2]))


def pack_heatmaps(hms: Sequence[torch.Tensor], jp: int = 16, out: Optional[torch.Tensor] = None,
                  out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """list[V] of (B,J,h,w) fp32|bf16 contiguous -> (V,B,h,w,jp) channels-last (fp32|bf16), padded channels zero."""
    lib = load()
    h0 = hms[0]
    _require_cuda(h0, "heatmaps")
    B, J, h, w = h0.shape
    V = len(hms)
    in_dt = torch.bfloat16 if h0.dtype == torch.bfloat16 else torch.float32
    hms = [x if (x.is_contiguous() and x.dtype == in_dt) else x.contiguous().to(in_dt) for x in hms]
    out_dtype = out.dtype if out is not None else (out_dtype or torch.float32)
    if out is None:
        out = torch.empty((V, B, h, w, jp), dtype=out_dtype, device=h0.device)
    check(lib.sp3d_pack_heatmaps_ex(_ptr_array(hms), out.data_ptr(), int(in_dt == torch.bfloat16),
                                    int(out_dtype == torch.bfloat16), B, V, J, jp, h, w, _stream(h0.device)),
          "sp3d_pack_heatmaps")
    return out


def unproject_fwd(views: Sequence[torch.Tensor], layout: int, jp: int, cam: torch.Tensor, centers: torch.Tensor,
                  valid: torch.Tensor, B: int, J: int, h: int, w: int, cube_size, grid_size, img_size,
                  want_grids: bool = True, variant: Optional[int] = None, channels_last: bool = False,
                  sample_of: Optional[torch.Tensor] = None, out_dtype: torch.dtype = torch.float32,
                  pass_mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
    """-> (cubes (B,J,X,Y,Z), grids (B,N,3) | None).  With ``channels_last`` the cubes tensor has
    torch.channels_last_3d strides (memory (B,X,Y,Z,J), J % 4 == 0, NHWC input only).
    ``sample_of`` (int32, (B,)): output cube p reads heat-map/camera row sample_of[p] (B = #cubes).
    ``out``: a (B,J,X,Y,Z) VIEW of a larger buffer (z contiguous, e.g. the zero-padded FFT input of the opening
    conv): the planar result is written straight into it (sp3d_unproject_fwd_strided)."""
    lib = load()
    dev = cam.device
    _require_cam(cam)
    X, Y, Z = (int(c) for c in cube_size)
    V = len(views)
    if out is not None:
        assert layout == LAYOUT_NHWC and not channels_last and not want_grids and pass_mask is None and variant is None
        assert tuple(out.shape) == (B, J, X, Y, Z) and out.stride(4) == 1 and out.dtype == out_dtype
        flags = (HM_BF16 if views[0].dtype == torch.bfloat16 else 0) | (OUT_BF16 if out_dtype == torch.bfloat16 else 0)
        st = (C.c_int64 * 4)(*[int(v) for v in out.stride()[:4]])
        rc = lib.sp3d_unproject_fwd_strided(_ptr_array(views), layout | flags, jp, cam.data_ptr(),
                                            sample_of.data_ptr() if sample_of is not None else None, centers.data_ptr(),
                                            valid.data_ptr(), out.data_ptr(), st, B, V, J, h, w, X, Y, Z, _f3(grid_size),
                                            int(img_size[0]), int(img_size[1]), _stream(dev))
        check(rc, "sp3d_unproject_fwd_strided")
        return out, None
    if channels_last:
        cubes = torch.empty((B, X, Y, Z, J), dtype=out_dtype, device=dev).permute(0, 4, 1, 2, 3)
    else:
        cubes = torch.empty((B, J, X, Y, Z), dtype=out_dtype, device=dev)
    flags = (OUT_CHANNELS_LAST if channels_last else 0) | (HM_BF16 if views[0].dtype == torch.bfloat16 else 0) | \
        (OUT_BF16 if out_dtype == torch.bfloat16 else 0)
    grids = torch.empty((B, X * Y * Z, 3), dtype=torch.float32, device=dev) if want_grids else None
    gs = _f3(grid_size)
    if pass_mask is not None:
        rc = lib.sp3d_unproject_fwd_train(_ptr_array(views), layout | flags, jp, cam.data_ptr(),
                                          sample_of.data_ptr() if sample_of is not None else None, centers.data_ptr(),
                                          valid.data_ptr(), cubes.data_ptr(), grids.data_ptr() if want_grids else None,
                                          pass_mask.data_ptr(), B, V, J, h, w, X, Y, Z, gs, int(img_size[0]),
                                          int(img_size[1]), _stream(dev))
    elif variant is None:
        rc = lib.sp3d_unproject_fwd_indexed(_ptr_array(views), layout | flags,
                                            jp, cam.data_ptr(), sample_of.data_ptr() if sample_of is not None else None,
                                            centers.data_ptr(), valid.data_ptr(), cubes.data_ptr(),
                                            grids.data_ptr() if want_grids else None, B, V, J, h, w, X, Y, Z, gs,
                                            int(img_size[0]), int(img_size[1]), _stream(dev))
    else:
        assert layout == LAYOUT_NHWC
        rc = lib.sp3d_unproject_fwd_variant(_ptr_array(views), jp, cam.data_ptr(), centers.data_ptr(),
                                            valid.data_ptr(), cubes.data_ptr(),
                                            grids.data_ptr() if want_grids else None, B, V, J, h, w, X, Y, Z, gs,
                                            int(img_size[0]), int(img_size[1]),
                                            int(variant) | (0x1000000 if channels_last else 0), _stream(dev))
    check(rc, "sp3d_unproject_fwd")
    return cubes, grids


def unproject_bwd(hms: Sequence[torch.Tensor], cam, centers, valid, grad_cubes: torch.Tensor, cube_size, grid_size,
                  img_size, sample_of: Optional[torch.Tensor] = None):
    lib = load()
    dev = cam.device
    _require_cam(cam)
    B, J, h, w = hms[0].shape
    P = int(grad_cubes.shape[0])
    X, Y, Z = (int(c) for c in cube_size)
    V = len(hms)
    hms = [x if x.dtype == torch.float32 else x.float() for x in hms]
    grad_cubes = grad_cubes[:, :J].float().contiguous()
    grads = torch.zeros((V, B, J, h, w), dtype=torch.float32, device=dev)
    gviews = [grads[c] for c in range(V)]
    rc = lib.sp3d_unproject_bwd_indexed(_ptr_array(hms), cam.data_ptr(),
                                        sample_of.data_ptr() if sample_of is not None else None, centers.data_ptr(),
                                        valid.data_ptr(), grad_cubes.data_ptr(), _ptr_array(gviews), P, V, J, h, w, X,
                                        Y, Z, _f3(grid_size), int(img_size[0]), int(img_size[1]), _stream(dev))
    check(rc, "sp3d_unproject_bwd")
    return gviews


def nms_proposals(root_cubes: torch.Tensor, k: int, grid_size, grid_center, threshold: float) -> torch.Tensor:
    """(B,X,Y,Z) -> grid_centers (B,k,5) = [x,y,z mm, (score > threshold) - 1, score]: NMS, top-k, index -> mm and the
    eval-mode proposal flags in the same two launches (no torch glue kernels)."""
    lib = load()
    lib.sp3d_nms_proposals.restype = C.c_int
    lib.sp3d_nms_proposals.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_float] + [C.c_void_p] * 6
    _require_cuda(root_cubes, "root_cubes")
    rc_ = root_cubes.contiguous().float()
    B, X, Y, Z = rc_.shape
    dev = rc_.device
    out = torch.empty((B, k, 5), dtype=torch.float32, device=dev)
    scratch = torch.empty((B * k * (1 + 3) * 4 + B * k * 3 * 8,), dtype=torch.uint8, device=dev)
    vals = scratch[:B * k * 4].view(torch.float32)
    locs = scratch[B * k * 4:B * k * 16].view(torch.float32)
    idx = scratch[B * k * 16:].view(torch.int64)
    nbytes = lib.sp3d_nms_topk_workspace_bytes(B, X, Y, Z, k)
    ws = torch.empty((max(int(nbytes), 8),), dtype=torch.uint8, device=dev)
    rc = lib.sp3d_nms_proposals(rc_.data_ptr(), B, X, Y, Z, k, _f3(grid_size), _f3(grid_center), C.c_float(float(threshold)),
                                vals.data_ptr(), idx.data_ptr(), locs.data_ptr(), out.data_ptr(), ws.data_ptr(), _stream(dev))
    check(rc, "sp3d_nms_proposals")
    return out


def nms_topk(root_cubes: torch.Tensor, k: int, grid_size=None, grid_center=None):
    """(B,X,Y,Z) -> vals (B,k) fp32, idx (B,k,3) int64, locs (B,k,3) fp32 mm (None without grid_size)."""
    lib = load()
    _require_cuda(root_cubes, "root_cubes")
    rc_ = root_cubes.contiguous().float()
    B, X, Y, Z = rc_.shape
    dev = rc_.device
    vals = torch.empty((B, k), dtype=torch.float32, device=dev)
    idx = torch.empty((B, k, 3), dtype=torch.int64, device=dev)
    locs = torch.empty((B, k, 3), dtype=torch.float32, device=dev) if grid_size is not None else None
    nbytes = lib.sp3d_nms_topk_workspace_bytes(B, X, Y, Z, k)
    ws = torch.empty((max(int(nbytes), 8),), dtype=torch.uint8, device=dev)
    rc = lib.sp3d_nms_topk(rc_.data_ptr(), B, X, Y, Z, k, _f3(grid_size) if grid_size is not None else None,
                           _f3(grid_center) if grid_center is not None else None, vals.data_ptr(), idx.data_ptr(),
                           locs.data_ptr() if locs is not None else None, ws.data_ptr(), _stream(dev))
    check(rc, "sp3d_nms_topk")
    return vals, idx, locs


def soft_argmax(x: torch.Tensor, grids: torch.Tensor, beta: float) -> torch.Tensor:
    """x (Bv,J,X,Y,Z) or (Bv,J,N); grids (Bv,N,3) -> (Bv,J,3)"""
    lib = load()
    _require_cuda(x, "x")
    Bv, J = x.shape[:2]
    xc = x.contiguous().float().reshape(Bv, J, -1)
    N = xc.shape[2]
    gc = grids.contiguous().float()
    out = torch.empty((Bv, J, 3), dtype=torch.float32, device=x.device)
    if Bv == 0:
        return out
    check(lib.sp3d_soft_argmax(xc.data_ptr(), gc.data_ptr(), out.data_ptr(), Bv, J, N, float(beta), _stream(x.device)),
          "sp3d_soft_argmax")
    return out


def soft_argmax_grid(x: torch.Tensor, centers: torch.Tensor, grid_size, cube_size, beta: float) -> torch.Tensor:
    """soft-argmax with voxel centres regenerated in-kernel: x (P,J,X,Y,Z), centers (P,3) -> (P,J,3)"""
    lib = load()
    _require_cuda(x, "x")
    P, J = x.shape[:2]
    X, Y, Z = (int(c) for c in cube_size)
    xc = x.contiguous().float()
    cc = centers.contiguous().float()
    out = torch.empty((P, J, 3), dtype=torch.float32, device=x.device)
    if P == 0:
        return out
    check(lib.sp3d_soft_argmax_grid(xc.data_ptr(), cc.data_ptr(), _f3(grid_size), X, Y, Z, out.data_ptr(), P, J,
                                    float(beta), _stream(x.device)), "sp3d_soft_argmax_grid")
    return out


class _SoftArgmaxGridFn(torch.autograd.Function):
    """soft-argmax with in-kernel voxel centres under autograd (training pose net): forward = sp3d_soft_argmax_grid_train (also
    keeps max and sum per row), backward = ONE elementwise pass (sp3d_soft_argmax_grid_bwd) instead of the softmax / product /
    sum graph over a (P,J,N,3) temporary"""

    @staticmethod
    def forward(ctx, x, centers, grid_size, cube_size, beta):
        lib = load()
        _require_cuda(x, "x")
        P, J = int(x.shape[0]), int(x.shape[1])
        X, Y, Z = (int(c) for c in cube_size)
        xc = x.detach().contiguous().float()                      # planar (P,J,N)
        cc = centers.detach().contiguous().float()
        out = torch.empty((P, J, 3), dtype=torch.float32, device=x.device)
        stats = torch.empty((P, J, 2), dtype=torch.float32, device=x.device)
        lib.sp3d_soft_argmax_grid_train.restype = C.c_int
        lib.sp3d_soft_argmax_grid_train.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                    C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
        check(lib.sp3d_soft_argmax_grid_train(xc.data_ptr(), cc.data_ptr(), _f3(grid_size), X, Y, Z, out.data_ptr(),
                                              stats.data_ptr(), P, J, float(beta), _stream(x.device)), "sp3d_soft_argmax_grid_train")
        ctx.save_for_backward(xc, cc, out, stats)
        ctx.geom = (tuple(float(v) for v in grid_size), (X, Y, Z), float(beta), x.is_contiguous(memory_format=torch.channels_last_3d)
                    and not x.is_contiguous())
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load()
        xc, cc, out, stats = ctx.saved_tensors
        grid_size, (X, Y, Z), beta, cl = ctx.geom
        P, J = int(xc.shape[0]), int(xc.shape[1])
        dx = torch.empty_like(xc)
        gc = g.contiguous().float()
        lib.sp3d_soft_argmax_grid_bwd.restype = C.c_int
        lib.sp3d_soft_argmax_grid_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
        check(lib.sp3d_soft_argmax_grid_bwd(xc.data_ptr(), cc.data_ptr(), _f3(grid_size), X, Y, Z, out.data_ptr(), stats.data_ptr(),
                                            gc.data_ptr(), dx.data_ptr(), P, J, beta, _stream(xc.device)), "sp3d_soft_argmax_grid_bwd")
        if cl:
            dx = dx.contiguous(memory_format=torch.channels_last_3d)
        return dx, None, None, None, None


def soft_argmax_grid_autograd(x: torch.Tensor, centers: torch.Tensor, grid_size, cube_size, beta: float) -> torch.Tensor:
    """x (P,J,X,Y,Z) fp32 (gradient flows to it), centers (P,3) -> (P,J,3)"""
    if int(x.shape[0]) == 0:
        return x.new_zeros((0, int(x.shape[1]), 3))
    return _SoftArgmaxGridFn.apply(x, centers, grid_size, cube_size, beta)


def fetch_ring(ring: torch.Tensor, dst: torch.Tensor, counter: torch.Tensor):
    """graph-capturable: dst <- ring[counter % R] (ring: PINNED host tensor (R, ...)), counter += 1 (device int32)"""
    lib = load()
    lib.sp3d_fetch_ring.restype = C.c_int
    lib.sp3d_fetch_ring.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    assert ring.is_pinned() and ring.dtype == torch.float32 and dst.is_cuda and counter.is_cuda
    R = int(ring.shape[0])
    n = ring[0].numel()
    assert dst.numel() == n and dst.is_contiguous() and ring.is_contiguous()
    check(lib.sp3d_fetch_ring(ring.data_ptr(), dst.data_ptr(), counter.data_ptr(), R, n, _stream(dst.device)),
          "sp3d_fetch_ring")


def maxpool2x(x: torch.Tensor) -> torch.Tensor:
    """MaxPool3d(2,2) of a channels_last_3d tensor (B,C,X,Y,Z) -> channels_last_3d (B,C,X/2,Y/2,Z/2)"""
    lib = load()
    lib.sp3d_maxpool2x_cl.restype = C.c_int
    lib.sp3d_maxpool2x_cl.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
    B, Cc, X, Y, Z = (int(v) for v in x.shape)
    y = torch.empty((B, X // 2, Y // 2, Z // 2, Cc), dtype=torch.float32, device=x.device).permute(0, 4, 1, 2, 3)
    check(lib.sp3d_maxpool2x_cl(x.data_ptr(), y.data_ptr(), B, X, Y, Z, Cc, _stream(x.device)), "sp3d_maxpool2x_cl")
    return y


def rfft3d(x: torch.Tensor) -> torch.Tensor:
    """unnormalised rFFT over the last three dims of a dense fp32 (..., SX,SY,SZ) tensor -> complex64 (..., SX,SY,SZ//2+1);
    x is only read (no defensive clone, unlike torch.fft.rfftn on ROCm)"""
    lib = load()
    _require_cuda(x, "x")
    if not x.is_contiguous() or x.dtype != torch.float32 or x.dim() < 3:
        raise Sp3dError("rfft3d: x must be a dense fp32 tensor with >= 3 dims")
    lib.sp3d_rfft3d.restype = C.c_int
    lib.sp3d_rfft3d.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
    SX, SY, SZ = (int(v) for v in x.shape[-3:])
    batch = int(x.numel() // (SX * SY * SZ))
    out = torch.empty(tuple(x.shape[:-1]) + (SZ // 2 + 1,), dtype=torch.complex64, device=x.device)
    check(lib.sp3d_rfft3d(x.data_ptr(), out.data_ptr(), batch, SX, SY, SZ, _stream(x.device)), "sp3d_rfft3d")
    return out


def irfft3d_(spec: torch.Tensor, SZ: int) -> torch.Tensor:
    """unnormalised inverse of rfft3d: complex64 (..., SX,SY,SZ//2+1) -> fp32 (..., SX,SY,SZ).  `spec` is scratch
    afterwards (the C2R plan may overwrite it)"""
    lib = load()
    _require_cuda(spec, "spec")
    if not spec.is_contiguous() or spec.dtype != torch.complex64 or spec.dim() < 3 or spec.shape[-1] != SZ // 2 + 1:
        raise Sp3dError("irfft3d_: spec must be a dense complex64 (..., SX,SY,SZ//2+1) tensor")
    lib.sp3d_irfft3d.restype = C.c_int
    lib.sp3d_irfft3d.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
    SX, SY = (int(v) for v in spec.shape[-3:-1])
    batch = int(spec.numel() // (SX * SY * (SZ // 2 + 1)))
    out = torch.empty(tuple(spec.shape[:-1]) + (int(SZ),), dtype=torch.float32, device=spec.device)
    check(lib.sp3d_irfft3d(spec.data_ptr(), out.data_ptr(), batch, SX, SY, int(SZ), _stream(spec.device)), "sp3d_irfft3d")
    return out


ZDFT_SHAPES = {(20, 28, 16)}       # (Z, SZ, channels) the direct z-DFT kernels are built for


def cfft2d_(spec: torch.Tensor, inverse: bool, rows_in: Optional[int] = None, rows_out: Optional[int] = None,
             library: bool = False) -> torch.Tensor:
    """in-place unnormalised 2-D complex FFT over the last two dims of a dense complex64 tensor.  rows_in: only the first
    rows_in rows of each input plane are non-zero; rows_out: only the first rows_out rows of each result are needed (the
    rest is unspecified).  library=True forces the hipFFT plan (whole planes)."""
    lib = load()
    _require_cuda(spec, "spec")
    if not spec.is_contiguous() or spec.dtype != torch.complex64 or spec.dim() < 2:
        raise Sp3dError("cfft2d_: dense complex64 tensor with >= 2 dims expected")
    lib.sp3d_cfft2d.restype = C.c_int
    lib.sp3d_cfft2d.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
    lib.sp3d_cfft2d_ex.restype = C.c_int
    lib.sp3d_cfft2d_ex.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]
    SX, SY = int(spec.shape[-2]), int(spec.shape[-1])
    batch = int(spec.numel() // (SX * SY))
    if library:
        check(lib.sp3d_cfft2d(spec.data_ptr(), batch, SX, SY, 1 if inverse else 0, _stream(spec.device)), "sp3d_cfft2d")
    else:
        check(lib.sp3d_cfft2d_ex(spec.data_ptr(), batch, SX, SY, 1 if inverse else 0, SX if rows_in is None else int(rows_in),
                                 SX if rows_out is None else int(rows_out), _stream(spec.device)), "sp3d_cfft2d_ex")
    return spec


def zdft_fwd_cl(x: torch.Tensor, cout: int, S) -> torch.Tensor:
    """channels_last_3d (B,C,X,Y,Z) real -> (B,cout,SZ//2+1,SX,SY) complex64: z-DFT of the rows zero-padded to S"""
    lib = load()
    _require_cuda(x, "x")
    B, Cc, X, Y, Z = (int(v) for v in x.shape)
    if x.dtype != torch.float32 or not x.permute(0, 2, 3, 4, 1).is_contiguous():
        raise Sp3dError("zdft_fwd_cl: dense fp32 channels_last_3d cubes expected")
    lib.sp3d_zdft_fwd_cl.restype = C.c_int
    lib.sp3d_zdft_fwd_cl.argtypes = [C.c_void_p] * 2 + [C.c_int] * 9 + [C.c_void_p]
    SX, SY, SZ = (int(v) for v in S)
    spec = torch.empty((B, int(cout), SZ // 2 + 1, SX, SY), dtype=torch.complex64, device=x.device)
    check(lib.sp3d_zdft_fwd_cl(x.data_ptr(), spec.data_ptr(), B, Cc, int(cout), X, Y, Z, SX, SY, SZ, _stream(x.device)),
          "sp3d_zdft_fwd_cl")
    return spec


def unproject_fwd_zdft(views: Sequence[torch.Tensor], jp: int, cam: torch.Tensor, centers: torch.Tensor, valid: torch.Tensor,
                       batch: int, J: int, h: int, w: int, cube_size, grid_size, img_size, SZ: int) -> torch.Tensor:
    """root grid, inference: the unprojection's result as the z-spectrum of its cubes (include/sp3d.h,
    sp3d_unproject_fwd_zdft) -> (B, J, SZ//2+1, X/4, Y/4, 16) complex64, 4 x 4-tiled planes without padding"""
    lib = load()
    _require_cam(cam)
    X, Y, Z = (int(c) for c in cube_size)
    for v in views:
        _require_cuda(v, "heat-map view")
        if v.dtype != torch.float32 or not v.is_contiguous() or tuple(v.shape) != (batch, h, w, jp):
            raise Sp3dError("unproject_fwd_zdft: dense fp32 (B,h,w,16) heat-map views expected")
    if (Z, int(SZ), jp) not in ZDFT_SHAPES or X % 4 or Y % 4:
        raise Sp3dError(f"unproject_fwd_zdft: built for (Z, SZ, channels) in {sorted(ZDFT_SHAPES)} and X, Y multiples of 4")
    spec = torch.empty((batch, J, int(SZ) // 2 + 1, X // 4, Y // 4, 16), dtype=torch.complex64, device=cam.device)
    lib.sp3d_unproject_fwd_zdft.restype = C.c_int
    lib.sp3d_unproject_fwd_zdft.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 8 + [C.c_void_p, C.c_int, C.c_int,
                                                                                                   C.c_int, C.c_void_p]
    check(lib.sp3d_unproject_fwd_zdft(_ptr_array(views), jp, cam.data_ptr(), centers.data_ptr(), valid.data_ptr(),
                                      spec.data_ptr(), batch, len(views), J, h, w, X, Y, Z, _f3(grid_size), int(img_size[0]),
                                      int(img_size[1]), int(SZ), _stream(cam.device)), "sp3d_unproject_fwd_zdft")
    return spec


def cfft2d_88_tiled(spec: torch.Tensor, X: int, Y: int) -> torch.Tensor:
    """(..., X/4, Y/4, 16) complex64 tiled planes -> (..., 88, 88) complex64: forward x,y transform of the zero-padded planes"""
    lib = load()
    _require_cuda(spec, "spec")
    if spec.dtype != torch.complex64 or not spec.is_contiguous() or tuple(spec.shape[-3:]) != (X // 4, Y // 4, 16):
        raise Sp3dError("cfft2d_88_tiled: dense complex64 (..., X/4, Y/4, 16) expected")
    out = torch.empty(tuple(spec.shape[:-3]) + (88, 88), dtype=torch.complex64, device=spec.device)
    batch = out.numel() // (88 * 88)
    lib.sp3d_cfft2d_88_tiled.restype = C.c_int
    lib.sp3d_cfft2d_88_tiled.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    check(lib.sp3d_cfft2d_88_tiled(spec.data_ptr(), out.data_ptr(), batch, int(X), int(Y), _stream(spec.device)),
          "sp3d_cfft2d_88_tiled")
    return out


def zdft_inv_cl(spec: torch.Tensor, X: int, Y: int, Z: int, SZ: int, shift: torch.Tensor, relu: bool = True) -> torch.Tensor:
    """(B,O,SZ//2+1,SX,SY) complex64 -> channels_last_3d (B,O,X,Y,Z) = act(shift[o] + unnormalised C2R along z)"""
    lib = load()
    _require_cuda(spec, "spec")
    if not spec.is_contiguous() or spec.dtype != torch.complex64 or spec.dim() != 5 or spec.shape[2] != SZ // 2 + 1:
        raise Sp3dError("zdft_inv_cl: dense complex64 (B,O,SZ//2+1,SX,SY) spectrum expected")
    lib.sp3d_zdft_inv_cl.restype = C.c_int
    lib.sp3d_zdft_inv_cl.argtypes = [C.c_void_p] * 3 + [C.c_int] * 9 + [C.c_void_p]
    B, O, _, SX, SY = (int(v) for v in spec.shape)
    y = torch.empty((B, X, Y, Z, O), dtype=torch.float32, device=spec.device).permute(0, 4, 1, 2, 3)
    check(lib.sp3d_zdft_inv_cl(spec.data_ptr(), y.data_ptr(), shift.data_ptr(), B, O, X, Y, Z, SX, SY, int(SZ),
                               1 if relu else 0, _stream(spec.device)), "sp3d_zdft_inv_cl")
    return y


def crop_shift_act_cl(src: torch.Tensor, X: int, Y: int, Z: int, shift: torch.Tensor, relu: bool = True) -> torch.Tensor:
    """planar (B,C,SX,SY,SZ) -> channels_last_3d (B,C,X,Y,Z) = act(src[:, :, :X, :Y, :Z] + shift[c]) in one pass"""
    lib = load()
    _require_cuda(src, "src")
    lib.sp3d_crop_shift_act_cl.restype = C.c_int
    lib.sp3d_crop_shift_act_cl.argtypes = [C.c_void_p] * 3 + [C.c_int] * 9 + [C.c_void_p]
    if not src.is_contiguous() or src.dtype != torch.float32:
        raise Sp3dError("crop_shift_act_cl: src must be a dense fp32 (B,C,SX,SY,SZ) tensor")
    B, Cc, SX, SY, SZ = (int(v) for v in src.shape)
    y = torch.empty((B, X, Y, Z, Cc), dtype=torch.float32, device=src.device).permute(0, 4, 1, 2, 3)
    check(lib.sp3d_crop_shift_act_cl(src.data_ptr(), y.data_ptr(), shift.data_ptr(), B, Cc, X, Y, Z, SX, SY, SZ,
                                     1 if relu else 0, _stream(src.device)), "sp3d_crop_shift_act_cl")
    return y


def channel_shift_act_(y: torch.Tensor, shift: torch.Tensor, mode: int, residual: Optional[torch.Tensor] = None):
    """in place on a 5-D activation (B,C,D,H,W) that is dense in NCDHW or channels_last_3d order:
    mode 0 y+=shift[c]; 1 relu(y+shift); 2 relu(y+shift+residual); 3 relu(y+shift)+residual"""
    lib = load()
    _require_cuda(y, "y")
    B, Cc = int(y.shape[0]), int(y.shape[1])
    inner = int(y.numel() // max(1, B * Cc))
    if y.is_contiguous():
        cl = 0
    elif y.is_contiguous(memory_format=torch.channels_last_3d):
        cl = 1
    else:
        raise Sp3dError("channel_shift_act_: activation must be dense (NCDHW or channels_last_3d)")
    if residual is not None:
        if residual.shape != y.shape or residual.stride() != y.stride():
            residual = residual.contiguous(memory_format=torch.channels_last_3d if cl else torch.contiguous_format)
    check(lib.sp3d_channel_shift_act(y.data_ptr(), shift.data_ptr(), residual.data_ptr() if residual is not None else None,
                                     int(mode), B, Cc, inner, cl, _stream(y.device)), "sp3d_channel_shift_act")
    return y


def unproject_bwd_packed(cam, centers, valid, grad_cubes: torch.Tensor, pass_mask: torch.Tensor, batch: int,
                         num_views: int, J: int, jp: int, h: int, w: int, cube_size, grid_size, img_size,
                         sample_of: Optional[torch.Tensor] = None, deterministic: bool = False, return_packed: bool = False,
                         scatter: int = SCATTER_AUTO):
    """line-coalesced scatter: -> list[V] of (B,J,h,w) gradient views into one (V,B,h,w,jp) channels-last buffer
    (``return_packed``: that buffer itself, pad channels zero).
    ``deterministic``: accumulate in 64-bit fixed point (integer atomics): bit-identical run to run.
    ``scatter``: SCATTER_AUTO (by voxel pitch) / SCATTER_PER_TAP / SCATTER_MERGE - a per-call choice (include/sp3d.h)."""
    lib = load()
    dev = cam.device
    _require_cam(cam)
    P = int(grad_cubes.shape[0])
    X, Y, Z = (int(c) for c in cube_size)
    gc = grad_cubes[:, :J].float().contiguous()
    if deterministic:
        I, Pp, V_ = C.c_int, C.c_void_p, C.c_void_p
        lib.sp3d_unproject_bwd_packed_det.restype = I
        lib.sp3d_unproject_bwd_packed_det.argtypes = [Pp] * 8 + [I] * 10 + [Pp, I, I, I, V_]
        lib.sp3d_fixed_to_float.restype = I
        lib.sp3d_fixed_to_float.argtypes = [Pp, Pp, Pp, C.c_int64, V_]
        # scale = 2^(40 - ceil(log2 max|g|)): computed on the device, no host synchronisation
        gmax = gc.abs().amax().clamp_min(1e-30)
        scale = torch.exp2(40.0 - torch.ceil(torch.log2(gmax))).to(torch.float32).reshape(1)
        fixed = torch.zeros((num_views, batch, h, w, jp), dtype=torch.int64, device=dev)
        rc = lib.sp3d_unproject_bwd_packed_det(cam.data_ptr(), sample_of.data_ptr() if sample_of is not None else None,
                                               centers.data_ptr(), valid.data_ptr(), gc.data_ptr(), pass_mask.data_ptr(),
                                               fixed.data_ptr(), scale.data_ptr(), int(batch), P, num_views, J, jp, h, w,
                                               X, Y, Z, _f3(grid_size), int(img_size[0]), int(img_size[1]), int(scatter),
                                               _stream(dev))
        check(rc, "sp3d_unproject_bwd_packed_det")
        packed = torch.empty((num_views, batch, h, w, jp), dtype=torch.float32, device=dev)
        check(lib.sp3d_fixed_to_float(fixed.data_ptr(), packed.data_ptr(), scale.data_ptr(), fixed.numel(), _stream(dev)),
              "sp3d_fixed_to_float")
        return packed if return_packed else [packed[c].permute(0, 3, 1, 2)[:, :J] for c in range(num_views)]
    packed = torch.zeros((num_views, batch, h, w, jp), dtype=torch.float32, device=dev)
    rc = lib.sp3d_unproject_bwd_packed(cam.data_ptr(), sample_of.data_ptr() if sample_of is not None else None,
                                       centers.data_ptr(), valid.data_ptr(), gc.data_ptr(), pass_mask.data_ptr(),
                                       packed.data_ptr(), int(batch), P, num_views, J, jp, h, w, X, Y, Z,
                                       _f3(grid_size), int(img_size[0]), int(img_size[1]), int(scatter), _stream(dev))
    check(rc, "sp3d_unproject_bwd_packed")
    return packed if return_packed else [packed[c].permute(0, 3, 1, 2)[:, :J] for c in range(num_views)]


def gaussian_target_3d(roots: torch.Tensor, gx: torch.Tensor, gy: torch.Tensor, gz: torch.Tensor, sigma: float):
    """roots (B,R,3), per-axis voxel centres -> (B,X,Y,Z) max-of-Gaussians target"""
    lib = load()
    _require_cuda(roots, "roots")
    B, R = roots.shape[:2]
    X, Y, Z = gx.numel(), gy.numel(), gz.numel()
    r = roots.contiguous().float()
    out = torch.empty((B, X, Y, Z), dtype=torch.float32, device=roots.device)
    check(lib.sp3d_gaussian_target_3d(r.data_ptr(), B, R, gx.data_ptr(), gy.data_ptr(), gz.data_ptr(), X, Y, Z, float(sigma),
                                      out.data_ptr(), _stream(roots.device)), "sp3d_gaussian_target_3d")
    return out


def render_root_heatmaps(roots: torch.Tensor, cam: torch.Tensor, h: int, w: int, stride: float):
    """roots (B,R,3), cam (B,V,64) -> (V,B,1,h,w) clipped sum of sigma-3 Gaussians at the projected roots"""
    lib = load()
    _require_cuda(roots, "roots")
    _require_cam(cam)
    B, R = roots.shape[:2]
    V = cam.shape[1]
    r = roots.contiguous().float()
    out = torch.empty((V, B, 1, h, w), dtype=torch.float32, device=roots.device)
    check(lib.sp3d_render_root_heatmaps(r.data_ptr(), B, R, cam.data_ptr(), V, h, w, float(stride), out.data_ptr(),
                                        _stream(roots.device)), "sp3d_render_root_heatmaps")
    return out


def freq_contract(Xf: torch.Tensor, Wf: torch.Tensor) -> torch.Tensor:
    """Xf (B,C,*F) complex64, Wf (O,C,*F) complex64 (contiguous) -> Yf (B,O,*F) = sum_c Xf * Wf"""
    lib = load()
    _require_cuda(Xf, "Xf")
    if Xf.dtype != torch.complex64 or Wf.dtype != torch.complex64 or Xf.shape[2:] != Wf.shape[2:] or Xf.shape[1] != Wf.shape[1]:
        raise Sp3dError("freq_contract: complex64 (B,C,*F) x (O,C,*F) expected")
    Xf, Wf = Xf.resolve_conj().contiguous(), Wf.resolve_conj().contiguous()      # data_ptr() ignores a lazy conj bit
    B, Cc = int(Xf.shape[0]), int(Xf.shape[1])
    O = int(Wf.shape[0])
    Fn = 1
    for d in Xf.shape[2:]:
        Fn *= int(d)
    Yf = torch.empty((B, O) + tuple(Xf.shape[2:]), dtype=torch.complex64, device=Xf.device)
    check(lib.sp3d_freq_contract(Xf.data_ptr(), Wf.data_ptr(), Yf.data_ptr(), B, Cc, O, Fn, _stream(Xf.device)),
          "sp3d_freq_contract")
    return Yf


def freq_contract_ty(Xf: torch.Tensor, T: torch.Tensor, tw: torch.Tensor) -> torch.Tensor:
    """Xf (B,C,KZ,SX,SY) complex64, T (KZ*SX, O, C, 14) fp32 (v2v_net._FoldedV2V._weights_ty), tw (SY,3,2) fp32
    -> (B,O,KZ,SX,SY) complex64: the forward contraction with W^ rebuilt per bin along y (include/sp3d.h)"""
    lib = load()
    _require_cuda(Xf, "Xf")
    B, Cc, KZ, SX, SY = (int(v) for v in Xf.shape)
    rows, O = int(T.shape[0]), int(T.shape[1])
    if Xf.dtype != torch.complex64 or not Xf.is_contiguous() or T.dtype != torch.float32 or not T.is_contiguous() or \
            tuple(T.shape) != (KZ * SX, O, Cc, 14) or tuple(tw.shape) != (SY, 3, 2) or not tw.is_contiguous():
        raise Sp3dError("freq_contract_ty: (B,C,KZ,SX,SY) complex64 x (KZ*SX,O,C,14) fp32 table expected")
    Yf = torch.empty((B, O, KZ, SX, SY), dtype=torch.complex64, device=Xf.device)
    lib.sp3d_freq_contract_ty.restype = C.c_int
    lib.sp3d_freq_contract_ty.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]
    check(lib.sp3d_freq_contract_ty(Xf.data_ptr(), T.data_ptr(), tw.data_ptr(), Yf.data_ptr(), B, Cc, O, rows, SY,
                                    _stream(Xf.device)), "sp3d_freq_contract_ty")
    return Yf


def freq_contract_ex(Pf: torch.Tensor, Qf: torch.Tensor, mode: str) -> torch.Tensor:
    """the three products of the frequency-domain conv on contiguous complex64 (A,B,*F) operands:
    'fwd'  Pf = X (B,C,*F), Qf = W^ (O,C,*F)  -> (B,O,*F) = sum_c X conj(W^)
    'dx'   Pf = Gy (B,O,*F), Qf = W^ (O,C,*F) -> (B,C,*F) = sum_o Gy W^
    'dw'   Pf = Gy (B,O,*F), Qf = X (B,C,*F)  -> (O,C,*F) = sum_b conj(Gy) X"""
    lib = load()
    _require_cuda(Pf, "Pf")
    Pf, Qf = Pf.resolve_conj().contiguous(), Qf.resolve_conj().contiguous()
    if Pf.dtype != torch.complex64 or Qf.dtype != torch.complex64 or Pf.shape[2:] != Qf.shape[2:]:
        raise Sp3dError("freq_contract_ex: complex64 operands with equal frequency shape expected")
    Fn = 1
    for d in Pf.shape[2:]:
        Fn *= int(d)
    p0, p1, q0, q1 = int(Pf.shape[0]), int(Pf.shape[1]), int(Qf.shape[0]), int(Qf.shape[1])
    if mode == "fwd":       # i=b, k=c, j=o
        I, K, J = p0, p1, q0
        sPi, sPk, sQj, sQk, cp, cq = p1 * Fn, Fn, q1 * Fn, Fn, 0, 1
        ok = q1 == p1
    elif mode == "dx":      # i=b, k=o, j=c
        I, K, J = p0, p1, q1
        sPi, sPk, sQj, sQk, cp, cq = p1 * Fn, Fn, Fn, q1 * Fn, 0, 0
        ok = q0 == p1
    elif mode == "dw":      # i=o, k=b, j=c
        I, K, J = p1, p0, q1
        sPi, sPk, sQj, sQk, cp, cq = Fn, p1 * Fn, Fn, q1 * Fn, 1, 0
        ok = q0 == p0
    else:
        raise Sp3dError(f"freq_contract_ex: unknown mode {mode}")
    if not ok:
        raise Sp3dError("freq_contract_ex: contracted dimensions differ")
    Y = torch.empty((I, J) + tuple(Pf.shape[2:]), dtype=torch.complex64, device=Pf.device)
    check(lib.sp3d_freq_contract_ex(Pf.data_ptr(), Qf.data_ptr(), Y.data_ptr(), I, J, K, Fn, sPi, sPk, sQj, sQk, cp, cq,
                                    _stream(Pf.device)), "sp3d_freq_contract_ex")
    return Y


_WINO_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))


def wino_weights(w: torch.Tensor) -> torch.Tensor:
    """(O,C,3,3,3) conv weights -> U (64, C, O) = G g G^T along each axis (Winograd F(2,3))"""
    G = torch.tensor(_WINO_G, dtype=torch.float64, device=w.device)
    u = torch.einsum("ai,bj,ck,ozijk->abczo", G, G, G, w.double())
    return u.reshape(64, w.shape[1], w.shape[0]).float().contiguous()


def wino_weights_split(U: torch.Tensor, chunk: int = 8) -> torch.Tensor:
    """U (64,C,O) fp32 -> the bf16 three-piece records the split kernels read: (64, C/chunk, chunk/4, O, 3, 4) bfloat16
    with pieces ordered (mid, hi, lo) and channel = chunk*c + 4*group + q; hi + mid + lo == U exactly (24 mantissa bits).
    chunk = 8: sp3d_wino_fused_split (lane halves), chunk = 16: sp3d_wino_fused_split64 (lane quarters)"""
    P, Cc, O = (int(v) for v in U.shape)
    U = U.float()
    hi = U.bfloat16()
    r = U - hi.float()
    mid = r.bfloat16()
    lo = (r - mid.float()).bfloat16()
    pieces = torch.stack([mid, hi, lo], 0).reshape(3, P, Cc // chunk, chunk // 4, 4, O)   # [piece, p, chunk, group, q, o]
    return pieces.permute(1, 2, 3, 5, 0, 4).contiguous()                                  # [p, chunk, group, o, piece, q]


def conv_weights_split(w: torch.Tensor) -> torch.Tensor:
    """(O,C,3,3,3) conv weights -> the bf16 operand records sp3d_conv3_split reads: (27, C/8, 2, O, 6, 4) bfloat16 = the
    three B operands {hi,lo} {hi,hi} {mid,mid} of 4 channels each, tap = kz*9 + ky*3 + kx, channel = 8*chunk + 4*half + q"""
    O, Cc = int(w.shape[0]), int(w.shape[1])
    pc = wino_weights_split(w.float().permute(4, 3, 2, 1, 0).reshape(27, Cc, O).contiguous(), 8)    # [.., piece (mid,hi,lo), 4]
    mid, hi, lo = pc[..., 0, :], pc[..., 1, :], pc[..., 2, :]
    return torch.stack([hi, lo, hi, hi, mid, mid], -2).contiguous()        # the B operands {bh,bl} {bh,bh} {bm,bm}


def conv3_split_(x: torch.Tensor, W3: torch.Tensor, shift: torch.Tensor, mode: int,
                 residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3x3x3 stride-1 'same' conv of channels_last_3d x with the fused epilogue, direct (implicit GEMM) on the bf16 matrix
    pipe with exact three-piece splits; W3 = conv_weights_split(w).  Returns the channels_last_3d result (B,O,X,Y,Z).
    (Round 3's chained form - split activations handed from layer to layer - measured no gain and was removed in round 4.)"""
    lib = load()
    _require_cuda(x, "x")
    B, Cc, X, Y, Z = (int(v) for v in x.shape)
    if not x.is_contiguous(memory_format=torch.channels_last_3d) or x.dtype != torch.float32:
        raise Sp3dError("conv3_split_: float32 channels_last_3d activations expected")
    O = int(W3.shape[3])
    dev = x.device
    y = torch.empty((B, X, Y, Z, O), dtype=torch.float32, device=dev).permute(0, 4, 1, 2, 3)
    if residual is not None and (tuple(residual.shape) != (B, O, X, Y, Z) or
                                 not residual.is_contiguous(memory_format=torch.channels_last_3d)):
        residual = residual.contiguous(memory_format=torch.channels_last_3d)
    lib.sp3d_conv3_split.restype = C.c_int
    lib.sp3d_conv3_split.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_void_p]
    check(lib.sp3d_conv3_split(x.data_ptr(), W3.data_ptr(), y.data_ptr(), shift.data_ptr(),
                               residual.data_ptr() if residual is not None else None, int(mode), B, X, Y, Z, Cc, O,
                               _stream(dev)), "sp3d_conv3_split")
    return y


def wino_conv3d_(x: torch.Tensor, U: torch.Tensor, shift: torch.Tensor, mode: int,
                 residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3x3x3 stride-1 'same' conv of channels-last x (B,C,X,Y,Z as torch.channels_last_3d) with pre-transformed weights
    U (64,C,O), fused with the layer epilogue; returns a channels_last_3d tensor (B,O,X,Y,Z)."""
    lib = load()
    _require_cuda(x, "x")
    B, Cc, X, Y, Z = (int(v) for v in x.shape)
    if not x.is_contiguous(memory_format=torch.channels_last_3d) or x.dtype != torch.float32:
        raise Sp3dError("wino_conv3d_: float32 channels_last_3d activations expected")
    O = int(U.shape[2])
    T = B * ((X + 1) // 2) * ((Y + 1) // 2) * ((Z + 1) // 2)
    V = torch.empty((64, T, Cc), dtype=torch.float32, device=x.device)
    check(lib.sp3d_wino_input(x.data_ptr(), V.data_ptr(), B, X, Y, Z, Cc, _stream(x.device)), "sp3d_wino_input")
    M = torch.bmm(V, U)         # the 64 products: the library's fp32 batched GEMM (an own split-bf16 GEMM was not faster: round 3)
    y = torch.empty((B, X, Y, Z, O), dtype=torch.float32, device=x.device).permute(0, 4, 1, 2, 3)
    if residual is not None and (residual.shape != y.shape or residual.stride() != y.stride()):
        residual = residual.contiguous(memory_format=torch.channels_last_3d)
    check(lib.sp3d_wino_output(M.data_ptr(), y.data_ptr(), shift.data_ptr(),
                               residual.data_ptr() if residual is not None else None, int(mode), B, X, Y, Z, O,
                               _stream(x.device)), "sp3d_wino_output")
    return y


def wino_fused_conv3d_(x: torch.Tensor, U: torch.Tensor, shift: torch.Tensor, mode: int,
                       residual: Optional[torch.Tensor] = None, U3: Optional[torch.Tensor] = None) -> torch.Tensor:
    """one-launch Winograd 3x3x3 conv (C = 16 | 32 -> O = 32; with U3 also C = 32 | 64 -> O = 64) of channels_last_3d x
    with the fused epilogue; with U3 (wino_weights_split(U, 8 | 16)) the products run as exact three-piece bf16 splits
    on the bf16 matrix pipe"""
    lib = load()
    _require_cuda(x, "x")
    B, Cc, X, Y, Z = (int(v) for v in x.shape)
    if not x.is_contiguous(memory_format=torch.channels_last_3d) or x.dtype != torch.float32:
        raise Sp3dError("wino_fused_conv3d_: float32 channels_last_3d activations expected")
    O = int(U.shape[2])
    y = torch.empty((B, X, Y, Z, O), dtype=torch.float32, device=x.device).permute(0, 4, 1, 2, 3)
    if residual is not None and (residual.shape != y.shape or residual.stride() != y.stride()):
        residual = residual.contiguous(memory_format=torch.channels_last_3d)
    if U3 is not None and O == 64:
        lib.sp3d_wino_fused_split64.restype = C.c_int
        lib.sp3d_wino_fused_split64.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_void_p]
        check(lib.sp3d_wino_fused_split64(x.data_ptr(), U3.data_ptr(), y.data_ptr(), shift.data_ptr(),
                                          residual.data_ptr() if residual is not None else None, int(mode), B, X, Y, Z, Cc, O,
                                          _stream(x.device)), "sp3d_wino_fused_split64")
        return y
    if U3 is not None:
        lib.sp3d_wino_fused_split.restype = C.c_int
        lib.sp3d_wino_fused_split.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_void_p]
        check(lib.sp3d_wino_fused_split(x.data_ptr(), U3.data_ptr(), y.data_ptr(), shift.data_ptr(),
                                        residual.data_ptr() if residual is not None else None, int(mode), B, X, Y, Z, Cc, O,
                                        _stream(x.device)), "sp3d_wino_fused_split")
        return y
    check(lib.sp3d_wino_fused(x.data_ptr(), U.data_ptr(), y.data_ptr(), shift.data_ptr(),
                              residual.data_ptr() if residual is not None else None, int(mode), B, X, Y, Z, Cc, O,
                              _stream(x.device)), "sp3d_wino_fused")
    return y


def upsample2x_(x: torch.Tensor, w_gemm: torch.Tensor, shift: torch.Tensor, skip: torch.Tensor) -> torch.Tensor:
    """ConvTranspose3d(k=2,s=2) + shift + ReLU + skip on channels_last_3d activations: one rocBLAS GEMM on the
    (voxels, Cin) view with w_gemm (Cin, 8*O) [columns (i,j,k,o)] + sp3d_upsample2x_scatter."""
    lib = load()
    _require_cuda(x, "x")
    B, Cc, X, Y, Z = (int(v) for v in x.shape)
    if not x.is_contiguous(memory_format=torch.channels_last_3d) or x.dtype != torch.float32:
        raise Sp3dError("upsample2x_: float32 channels_last_3d activations expected")
    O = int(w_gemm.shape[1]) // 8
    G = torch.matmul(x.permute(0, 2, 3, 4, 1).reshape(-1, Cc), w_gemm)
    out = torch.empty((B, 2 * X, 2 * Y, 2 * Z, O), dtype=torch.float32, device=x.device).permute(0, 4, 1, 2, 3)
    if skip.shape != out.shape or skip.stride() != out.stride():
        skip = skip.contiguous(memory_format=torch.channels_last_3d)
    check(lib.sp3d_upsample2x_scatter(G.data_ptr(), out.data_ptr(), shift.data_ptr(), skip.data_ptr(), B, X, Y, Z, O,
                                      _stream(x.device)), "sp3d_upsample2x_scatter")
    return out


def upsample2x_head_(x: torch.Tensor, w_gemm: torch.Tensor, shift: torch.Tensor, skip: torch.Tensor, w_out: torch.Tensor,
                     b_out: torch.Tensor) -> torch.Tensor:
    """upsample2x_ fused with the 1x1x1 output conv that is its only consumer: returns (B,J,2X,2Y,2Z) as a permuted view
    of a (B,2X,2Y,2Z,J) tensor (for J = 1 that is also the dense NCDHW tensor)"""
    lib = load()
    _require_cuda(x, "x")
    lib.sp3d_upsample2x_scatter_head.restype = C.c_int
    lib.sp3d_upsample2x_scatter_head.argtypes = [C.c_void_p] * 6 + [C.c_int64] + [C.c_int] * 5 + [C.c_void_p]
    B, Cc, X, Y, Z = (int(v) for v in x.shape)
    if not x.is_contiguous(memory_format=torch.channels_last_3d) or x.dtype != torch.float32:
        raise Sp3dError("upsample2x_head_: float32 channels_last_3d activations expected")
    O = int(w_gemm.shape[1]) // 8
    J = int(w_out.shape[0])
    G = torch.matmul(x.permute(0, 2, 3, 4, 1).reshape(-1, Cc), w_gemm)
    if tuple(skip.shape) != (B, O, 2 * X, 2 * Y, 2 * Z) or not skip.is_contiguous(memory_format=torch.channels_last_3d):
        skip = skip.contiguous(memory_format=torch.channels_last_3d)
    wo = w_out.reshape(J, O).contiguous().float()
    bo = (b_out if b_out is not None else torch.zeros(J, device=x.device)).contiguous().float()
    head = torch.empty((B, 2 * X, 2 * Y, 2 * Z, J), dtype=torch.float32, device=x.device)
    check(lib.sp3d_upsample2x_scatter_head(G.data_ptr(), head.data_ptr(), shift.data_ptr(), skip.data_ptr(), wo.data_ptr(),
                                           bo.data_ptr(), B, X, Y, Z, O, J, _stream(x.device)), "sp3d_upsample2x_scatter_head")
    return head.permute(0, 4, 1, 2, 3)


class _RenderJoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kps, count, h, w, sigma):
        lib = load()
        _require_cuda(kps, "kps")
        N, Pn, J = (int(v) for v in kps.shape[:3])
        k = kps.contiguous().float()
        cnt = None if count is None else count.to(device=kps.device, dtype=torch.int32).contiguous()
        out = torch.empty((N, J, h, w), dtype=torch.float32, device=kps.device)
        check(lib.sp3d_render_joints_fwd(k.data_ptr(), cnt.data_ptr() if cnt is not None else None, N, Pn, J, h, w,
                                         float(sigma), out.data_ptr(), _stream(kps.device)), "sp3d_render_joints_fwd")
        ctx.save_for_backward(k, cnt if cnt is not None else torch.empty(0, device=kps.device))
        ctx.geom = (h, w, float(sigma), cnt is not None, kps.dtype)
        return out if kps.dtype == torch.float32 else out.to(kps.dtype)      # float64 callers: fp32 kernel, caller's type

    @staticmethod
    def backward(ctx, gout):
        lib = load()
        k, cnt = ctx.saved_tensors
        h, w, sigma, has_cnt, in_dtype = ctx.geom
        N, Pn, J = (int(v) for v in k.shape[:3])
        g = gout.contiguous().float()
        gk = torch.empty_like(k)
        check(lib.sp3d_render_joints_bwd(k.data_ptr(), cnt.data_ptr() if has_cnt else None, g.data_ptr(), N, Pn, J, h, w,
                                         sigma, gk.data_ptr(), _stream(k.device)), "sp3d_render_joints_bwd")
        return gk.to(in_dtype), None, None, None, None


def render_joint_heatmaps(kps: torch.Tensor, count: Optional[torch.Tensor], h: int, w: int, sigma: float = 3.0):
    """kps (N,P,J,2) heat-map pixels, count (N,) people per entry or None -> (N,J,h,w), differentiable in kps"""
    return _RenderJoints.apply(kps, count, int(h), int(w), float(sigma))
