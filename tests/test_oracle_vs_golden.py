"""Pins the CPU oracle (oracle/sp3d_oracle.c) against the reference's own outputs (tests/golden)."""
import numpy as np
import pytest

from oracle import oracle
from selfpose3d_amd import synthetic as syn
from selfpose3d_amd.camera_pack import get_affine_transform_batch, pack_cameras
from tests import golden_io as gio

VOX_TOL = 5e-7      # north_star allows 1e-4; the restatement is ~1 ulp from torch-CPU


def test_linspace_matches_torch():
    import torch
    for L, n in ((8000.0, 80), (2000.0, 20), (2000.0, 64), (8000.0, 160), (2000.0, 40), (8000.0, 5), (2000.0, 3),
                 (2000.0, 16), (8000.0, 12), (8000.0, 7), (2000.0, 1)):
        ref = torch.linspace(-L / 2, L / 2, n).numpy()
        assert np.array_equal(oracle.linspace(L, n), ref), (L, n)


def test_affine_golden():
    g = gio.load("affine")
    n = len(g["rot"])
    got_c = np.stack([oracle.affine(g["center"][i], g["scale"][i], g["rot"][i], g["img"][i]) for i in range(n)])
    assert np.abs(got_c - g["trans"]).max() < 1e-9
    # the product's vectorised host twin, per output size
    for img in np.unique(g["img"], axis=0):
        sel = np.all(g["img"] == img, axis=1)
        got = get_affine_transform_batch(g["center"][sel], g["scale"][sel], g["rot"][sel], img)
        assert np.abs(got - g["trans"][sel]).max() < 1e-9
        assert np.array_equal(got.astype(np.float32), g["trans"][sel].astype(np.float32))


def test_project_pose_golden_bit_exact():
    g = gio.load("project_pose")
    meta = syn.make_meta(1, 5, (960, 512))
    cam = pack_cameras(meta, 1, (960, 512))
    for c in range(5):
        got = oracle.project_points(cam[0, c], g["pts"])
        ref = g["px"][c]
        assert np.array_equal(got, ref, equal_nan=True), c


@pytest.mark.parametrize("name", gio.SMALL_CASES + gio.FULL_CASES)
def test_unproject_fwd_golden(name):
    case = gio.Case(name)
    cubes, grids = oracle.unproject_fwd([h.numpy() for h in case.hms], case.cam, case.centers, case.valid,
                                        case.grid_size, case.cube, case.img)
    exp_c, exp_g, idx = case.expected()
    got_c = cubes.reshape(case.B, case.J, case.N)
    if idx is not None:
        got_c, got_g = got_c[:, :, idx], grids[:, idx]
    else:
        got_g = grids
    assert np.array_equal(got_g, exp_g), "grids must be bit-exact"
    assert np.abs(got_c - exp_c).max() <= VOX_TOL
    # whole-volume checksums (float64 sums over every voxel, also for sub-sampled goldens)
    assert abs(cubes.astype(np.float64).sum() - float(case.g["cubes_sum"])) <= 1e-7 * cubes.size
    assert np.allclose(cubes.astype(np.float64).sum(axis=(0, 2, 3, 4)), case.g["cubes_sum_per_joint"],
                       rtol=0, atol=1e-7 * cubes.size)
    assert np.allclose(grids.astype(np.float64).sum(axis=(0, 1)), case.g["grids_sum"], rtol=1e-12, atol=1e-3)


@pytest.mark.parametrize("name", ["unproj_grad_small", "unproj_grad_fine_aug"])
def test_unproject_bwd_golden(name):
    case = gio.Case(name)
    g = case.g
    wgt = np.random.default_rng(int(g["grad_seed"])).standard_normal(
        (case.B, case.J, *case.cube)).astype(np.float32)
    grads = oracle.unproject_bwd([h.numpy() for h in case.hms], case.cam, case.centers, case.valid, wgt,
                                 case.grid_size, case.cube, case.img)
    ref = g["grad_hm"]
    for c in range(case.V):
        scale = max(1.0, float(np.abs(ref[c]).max()))
        assert np.abs(grads[c] - ref[c]).max() <= 2e-5 * scale, c


def test_nms_golden_indices_bit_exact():
    g = gio.load("nms")
    case = gio.Case("unproj_people_coarse")
    cubes, _ = oracle.unproject_fwd([h.numpy() for h in case.hms], case.cam, case.centers, case.valid,
                                    case.grid_size, case.cube, case.img, want_grids=False)
    vals, idx = oracle.nms_topk(np.ascontiguousarray(cubes[:, 2]), 10)
    pos = g["people_vals"] > 0
    assert pos.sum() >= 4
    assert np.abs(vals - g["people_vals"]).max() <= VOX_TOL
    assert np.array_equal(idx[pos], g["people_idx"][pos])
    rnd = np.random.default_rng(int(g["rnd_seed"])).random(tuple(g["rnd_shape"]), dtype=np.float32)
    vals, idx = oracle.nms_topk(rnd, 10)
    assert np.array_equal(vals, g["rnd_vals"])
    assert np.array_equal(idx, g["rnd_idx"])


def test_posenet_full_cubes_golden():
    """the fine-grid unprojection at the pose stage's full size (5 views, 240x128, J=15, 64^3) against the cubes the
    reference PoseRegressionNet handed to its V2V (tests/golden/make_goldens_r4.py): 3 valid proposals + 1 skipped"""
    g = gio.load("posenet_full")
    img, hm = [int(v) for v in g["img"]], [int(v) for v in g["hm"]]
    V, J, B = int(g["V"]), int(g["J"]), int(g["B"])
    cube = [int(v) for v in g["fine_cube"]]
    N = cube[0] * cube[1] * cube[2]
    meta = syn.make_meta(B, V, img)
    hms, _ = syn.people_heatmaps(B, V, J, hm[1], hm[0], img, seed=int(g["hm_seed"]))
    cam = pack_cameras(meta, B, img)
    gc = g["grid_centers"]
    for k in range(gc.shape[1]):
        valid = (gc[:, k, 3] >= 0).astype(np.uint8)
        cubes, _ = oracle.unproject_fwd([h.numpy() for h in hms], cam, np.ascontiguousarray(gc[:, k, :3]), valid,
                                        syn.FINE_GRID_SIZE, cube, img, want_grids=False)
        got = cubes.reshape(B, J, N)[valid.astype(bool)]
        assert np.abs(got[:, :, g["sub_idx"]] - g[f"cube_sub_{k}"]).max() <= VOX_TOL
        assert np.allclose(got.astype(np.float64).sum(axis=2), g[f"cube_sum_{k}"], rtol=0, atol=1e-7 * N)
        assert np.count_nonzero(cubes[~valid.astype(bool)]) == 0


# ---- round 6: the backward (and the B=4 forward) at the sizes the kernels run at ------------------------------------------
def grad_full_check(got, g, tol_rel=2e-5, what=""):
    """(V,B,J,h,w) gradient against a full-size golden of tests/golden/make_goldens_r6.py: the stored sub-sample and two
    whole planes element by element, float64 sums / position-weighted sums per (view, sample, joint), and the exact set
    of touched pixels per plane.  Shared by tests/test_gpu_bwd_full_size.py."""
    V, B, J, h, w = got.shape
    g64 = got.astype(np.float64)
    scale = max(1.0, float(np.abs(g["grad_sub"]).max()))
    sub = got.reshape(-1)[::int(g["grad_stride"])]
    assert np.abs(sub - g["grad_sub"]).max() <= tol_rel * scale, what
    assert np.abs(got[0, 0, 2] - g["grad_plane_v0_b0_j2"]).max() <= tol_rel * scale, what
    assert np.abs(got[V - 1, B - 1, J - 1] - g["grad_plane_vl_bl_jl"]).max() <= tol_rel * scale, what
    # sums over a 30 720-pixel plane of terms each good to tol_rel * scale: bound by the plane's own mass
    mass = g["grad_abs_sum"]
    assert np.all(np.abs(g64.sum(axis=(3, 4)) - g["grad_sum"]) <= tol_rel * (mass + 1.0)), what
    pw = np.random.default_rng(int(g["pos_seed"])).standard_normal((h, w))
    assert np.all(np.abs((g64 * pw).sum(axis=(3, 4)) - g["grad_pos_sum"]) <= tol_rel * (mass + 1.0)), what
    assert np.all(np.abs(np.abs(g64).sum(axis=(3, 4)) - mass) <= tol_rel * (mass + 1.0)), what
    assert np.all(np.abs(np.abs(got).max(axis=(3, 4)) - g["grad_absmax"]) <= tol_rel * scale), what


@pytest.mark.parametrize("name", gio.GRAD_FULL_CASES)
def test_unproject_bwd_golden_full_size(name):
    """oracle.unproject_bwd (double accumulation, joints split over threads) against the reference's autograd at B=4,
    80x80x20 and at four 64^3 cubes: values, sums and the EXACT touched-pixel count of every plane"""
    case = gio.Case(name)
    g = case.g
    wgt = np.random.default_rng(int(g["grad_seed"])).standard_normal((case.B, case.J, *case.cube)).astype(np.float32)
    grads = np.stack(oracle.unproject_bwd([h.numpy() for h in case.hms], case.cam, case.centers, case.valid, wgt,
                                          case.grid_size, case.cube, case.img))
    grad_full_check(grads.astype(np.float32), g, what=name)
    assert np.array_equal((grads != 0).sum(axis=(3, 4)), g["grad_nonzero"])
    if not bool(g["center_is_list"]):
        inv = np.flatnonzero(case.valid == 0)
        assert len(inv) and not grads[:, inv].any()                  # the skipped sample gets no gradient
    # forward of the same case: clamp populations equal the reference's voxel for voxel (what the pass mask encodes)
    cubes, _ = oracle.unproject_fwd([h.numpy() for h in case.hms], case.cam, case.centers, case.valid, case.grid_size,
                                    case.cube, case.img)
    assert np.array_equal(((cubes > 0) & (cubes < 1)).sum(axis=(2, 3, 4)), g["cubes_interior"])
    assert np.array_equal((cubes == 1).sum(axis=(2, 3, 4)), g["cubes_at_one"])
    assert np.abs(cubes.reshape(case.B, case.J, -1)[:, :, g["sub_idx"]] - g["cubes_sub"]).max() <= VOX_TOL


def test_oracle_threads_change_no_bit_of_the_backward():
    case = gio.Case("unproj_grad_fine_aug")
    wgt = np.random.default_rng(3).standard_normal((case.B, case.J, *case.cube)).astype(np.float32)
    args = ([h.numpy() for h in case.hms], case.cam, case.centers, case.valid, wgt, case.grid_size, case.cube, case.img)
    prev = oracle.set_threads(1)
    try:
        one = np.stack(oracle.unproject_bwd(*args))
        oracle.set_threads(max(2, prev))
        many = np.stack(oracle.unproject_bwd(*args))
    finally:
        oracle.set_threads(prev)
    assert np.array_equal(one, many)
