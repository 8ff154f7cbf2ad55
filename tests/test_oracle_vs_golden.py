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
