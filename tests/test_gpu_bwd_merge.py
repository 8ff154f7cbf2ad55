"""The block-merge backward scatter (unproject_bwd3_kernel, round 4: an 8x8x4 block of voxels adds its taps in an LDS patch in
64-bit fixed point; every touched pixel leaves the CU once) on DENSE grids, against
  * the oracle's backward (autograd of lib/models/project_layer.py:93-99 restated on the CPU),
  * the per-tap scatter (unproject_bwd2_kernel) on the same inputs - the deterministic forms BIT for bit (integer sums),
on grids that are not multiples of the block, with one window per view and with several (a large footprint), 4 / 8 / 16
channel slots, indexed cubes (sample_of) and an invalid cube.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

IMG = (960, 512)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _setup(dev, cube, grid_size, hm, J, jp, B=2, V=5, seed=0, centers=None, sample_of=None, valid=None):
    from selfpose3d_amd import _lib, synthetic as syn
    from selfpose3d_amd.camera_pack import pack_cameras
    w, h = hm
    meta = syn.make_meta(B, V, IMG)
    cam_np = pack_cameras(meta, B, IMG)
    hms = syn.random_heatmaps(B, V, J, h, w, seed=seed)
    rng = np.random.default_rng(seed)
    if centers is None:
        centers = np.stack([rng.uniform(-600, 600, B), rng.uniform(-900, 100, B), rng.uniform(700, 1000, B)], 1).astype(np.float32)
    P = centers.shape[0]
    valid = np.ones(P, np.uint8) if valid is None else valid
    wgt = rng.standard_normal((P, J, *cube)).astype(np.float32)
    wgt *= np.exp(rng.uniform(-6, 3, (P, J, 1, 1, 1))).astype(np.float32)       # channels of very different magnitude
    d = dict(cam=torch.from_numpy(cam_np).to(dev), cen=torch.from_numpy(centers).to(dev), val=torch.from_numpy(valid).to(dev),
             so=None if sample_of is None else torch.from_numpy(sample_of).to(dev), wgt=torch.from_numpy(wgt).to(dev))
    packed = _lib.pack_heatmaps([x.to(dev) for x in hms], jp=jp)
    N = cube[0] * cube[1] * cube[2]
    mask = torch.empty((P, N), dtype=torch.int16, device=dev)
    _lib.unproject_fwd([packed[c] for c in range(V)], _lib.LAYOUT_NHWC, jp, d["cam"], d["cen"], d["val"], P, J, h, w, cube,
                       grid_size, IMG, False, sample_of=d["so"], pass_mask=mask)
    return d, mask, hms, cam_np, centers, valid, wgt


def _run(which, d, mask, B, V, J, jp, hm, cube, grid_size, deterministic=False):
    from selfpose3d_amd import _lib
    out = _lib.unproject_bwd_packed(d["cam"], d["cen"], d["val"], d["wgt"], mask, B, V, J, jp, hm[1], hm[0], cube, grid_size,
                                    IMG, sample_of=d["so"], deterministic=deterministic, scatter=which)
    return torch.stack([o.contiguous() for o in out])


CASES = {
    # name: cube, grid (mm), heat-map (w, h), J, jp
    "person_cube_ragged": ((20, 18, 13), (620.0, 560.0, 390.0), (240, 128), 15, 16),      # 32 mm pitch, ~1.7 px: one window
    "several_windows": ((16, 16, 8), (750.0, 750.0, 350.0), (480, 256), 15, 16),          # 50 mm at twice the resolution
    "four_slots": ((9, 17, 6), (300.0, 560.0, 180.0), (240, 128), 3, 4),
    "eight_slots": ((8, 8, 4), (250.0, 250.0, 110.0), (120, 64), 7, 8),
    "twelve_slots": ((11, 5, 9), (330.0, 140.0, 260.0), (240, 128), 10, 12),
}


@pytest.mark.parametrize("name", list(CASES))
def test_block_merge_scatter_vs_oracle_and_per_tap(dev, name):
    from oracle import oracle
    cube, grid_size, hm, J, jp = CASES[name]
    B, V = 2, 5
    d, mask, hms, cam_np, centers, valid, wgt = _setup(dev, cube, grid_size, hm, J, jp, B=B, V=V, seed=len(name))
    ref = oracle.unproject_bwd([x.numpy() for x in hms], cam_np, centers, valid, wgt, list(grid_size), list(cube), IMG)
    got3 = _run(3, d, mask, B, V, J, jp, hm, cube, grid_size).cpu().numpy()
    got2 = _run(2, d, mask, B, V, J, jp, hm, cube, grid_size).cpu().numpy()
    auto = _run(0, d, mask, B, V, J, jp, hm, cube, grid_size).cpu().numpy()
    assert np.count_nonzero(got3) > 500
    for c in range(V):
        # per channel: the block's fixed-point scale comes from the largest |g| of ALL its channels; the small channels
        # must come out as accurately as the per-tap fp32 scatter's
        for j in range(J):
            scale = max(1e-30, float(np.abs(ref[c][:, j]).max()))
            e3 = float(np.abs(got3[c][:, j] - ref[c][:, j]).max()) / scale
            e2 = float(np.abs(got2[c][:, j] - ref[c][:, j]).max()) / scale
            assert e3 <= 2e-5, (name, c, j, e3, e2)
            assert e3 <= max(2.0 * e2, 2e-6), (name, c, j, e3, e2)
    assert np.abs(auto - got3).max() <= 1e-5 * np.abs(got3).max()        # dense grid: the library picks the merge kernel
    # deterministic forms: the same integers, whatever the kernel and the run
    det3 = _run(3, d, mask, B, V, J, jp, hm, cube, grid_size, deterministic=True)
    det2 = _run(2, d, mask, B, V, J, jp, hm, cube, grid_size, deterministic=True)
    assert torch.equal(det3, det2)
    assert torch.equal(det3, _run(3, d, mask, B, V, J, jp, hm, cube, grid_size, deterministic=True))
    assert np.abs(det3.cpu().numpy() - np.stack(ref)).max() <= 2e-6 * max(1.0, float(np.abs(np.stack(ref)).max()))


def test_block_merge_scatter_indexed_cubes_and_invalid_cube(dev):
    """P = 5 cubes over B = 2 samples (sample_of), cube 3 invalid: merge == per tap (deterministic: the same bits)"""
    cube, grid_size, hm, J, jp = (24, 16, 12), (760.0, 500.0, 370.0), (240, 128), 15, 16
    rng = np.random.default_rng(5)
    P = 5
    centers = np.stack([rng.uniform(-900, 900, P), rng.uniform(-1200, 300, P), rng.uniform(600, 1100, P)], 1).astype(np.float32)
    sample_of = np.array([0, 1, 1, 0, 1], np.int32)
    valid = np.array([1, 1, 1, 0, 1], np.uint8)
    d, mask, *_ = _setup(dev, cube, grid_size, hm, J, jp, centers=centers, sample_of=sample_of, valid=valid, seed=9)
    d3 = _run(3, d, mask, 2, 5, J, jp, hm, cube, grid_size, deterministic=True)
    d2 = _run(2, d, mask, 2, 5, J, jp, hm, cube, grid_size, deterministic=True)
    assert torch.equal(d3, d2) and torch.count_nonzero(d3) > 0
    f3 = _run(3, d, mask, 2, 5, J, jp, hm, cube, grid_size)
    assert (f3 - d3).abs().max() <= 2e-6 * d3.abs().max()
    # the invalid cube contributed nothing: with all five valid (same gradients) the sums differ
    da, maska, *_ = _setup(dev, cube, grid_size, hm, J, jp, centers=centers, sample_of=sample_of, valid=np.ones(P, np.uint8), seed=9)
    assert torch.equal(da["wgt"], d["wgt"])
    assert not torch.equal(_run(3, da, maska, 2, 5, J, jp, hm, cube, grid_size, deterministic=True), d3)


def test_block_merge_scatter_zero_and_nonfinite_gradients(dev):
    """an all-zero gradient scatters nothing; an Inf in the gradient reaches the pixels its voxel touches and no others
    (the block with the non-finite value takes the per-tap path: no fixed-point scale exists for it)"""
    cube, grid_size, hm, J, jp = (16, 16, 8), (500.0, 500.0, 230.0), (240, 128), 15, 16
    d, mask, *_ = _setup(dev, cube, grid_size, hm, J, jp, seed=3)
    base = _run(3, d, mask, 2, 5, J, jp, hm, cube, grid_size)
    keep = d["wgt"]
    d["wgt"] = torch.zeros_like(keep)
    assert torch.count_nonzero(_run(3, d, mask, 2, 5, J, jp, hm, cube, grid_size)) == 0
    w = keep.clone()
    w[0, 2, 5, 6, 3] = float("inf")
    d["wgt"] = w
    got = _run(3, d, mask, 2, 5, J, jp, hm, cube, grid_size)
    ref = _run(2, d, mask, 2, 5, J, jp, hm, cube, grid_size)
    bad3, bad2 = ~torch.isfinite(got), ~torch.isfinite(ref)
    assert torch.equal(bad3, bad2) and 0 < int(bad3.sum()) <= 5 * 4      # the 2x2 taps of one voxel in each view, channel 2
    ok = ~bad3
    assert (got[ok] - base[ok]).abs().max() <= 1e-4 * base.abs().max()


@pytest.mark.parametrize("layout", ["planar", "nhwc"])
def test_registered_op_backward_takes_the_fast_scatter(dev, layout):
    """torch.ops.selfpose3d_mi.unproject_bwd with J <= 16 = pass-mask forward + packed scatter (here: the block merge, a
    dense grid): the same gradient as the first-generation planar scatter it used to call, in both input layouts"""
    import selfpose3d_amd.torch_ops  # noqa: F401
    from selfpose3d_amd import _lib
    cube, grid_size, hm, J, jp = CASES["person_cube_ragged"]
    d, mask, hms, cam_np, centers, valid, wgt = _setup(dev, cube, grid_size, hm, J, jp, seed=21)
    hmd = [x.to(dev) for x in hms]
    ref = torch.stack(_lib.unproject_bwd(hmd, d["cam"], d["cen"], d["val"], d["wgt"], list(cube), list(grid_size), IMG))
    inp = torch.stack(hmd, 0) if layout == "planar" else _lib.pack_heatmaps(hmd, jp=16)
    got = torch.ops.selfpose3d_mi.unproject_bwd(d["wgt"], inp, d["cam"], d["cen"], d["val"], list(grid_size), list(cube),
                                                list(IMG), list(hm), J)
    assert got.shape == inp.shape
    if layout == "nhwc":
        assert float(got[..., J:].abs().max()) == 0.0
        got = got[..., :J].permute(0, 1, 4, 2, 3)
    for j in range(J):
        scale = float(ref[:, :, j].abs().max())
        assert float((got[:, :, j] - ref[:, :, j]).abs().max()) <= 2e-5 * scale, j
