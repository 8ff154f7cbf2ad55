"""Randomised parity sweep: HIP (all three forward kernels, through the C ABI) vs the CPU oracle on random
rigs / shapes, bit for bit.  Covers ragged and extreme shapes: V = 1..16, J = 1..20, 2-pixel heat-maps, cubes
whose voxel count is not a multiple of 4 / 64, negative and > 1 heat-map values, rotated / scaled / flipped
crops, skipped samples, per-sample centres."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(2024)
    out = []
    fixed = [  # (B, V, J, (w,h) heat-map, cube)
        (1, 1, 1, (2, 2), (1, 1, 1)), (2, 16, 16, (9, 7), (5, 3, 7)), (1, 3, 17, (16, 12), (8, 8, 4)),
        (3, 2, 20, (33, 17), (6, 6, 6)), (2, 5, 15, (96, 72), (17, 13, 9)), (1, 10, 15, (48, 36), (32, 32, 8)),
        (4, 4, 13, (24, 18), (16, 4, 4)), (2, 7, 4, (5, 64), (4, 4, 33)),
    ]
    for f in fixed:
        out.append(f)
    for _ in range(16):
        out.append((int(rng.integers(1, 5)), int(rng.integers(1, 9)), int(rng.integers(1, 17)),
                    (int(rng.integers(2, 80)), int(rng.integers(2, 60))),
                    (int(rng.integers(1, 20)), int(rng.integers(1, 20)), int(rng.integers(1, 24)))))
    return out


@pytest.mark.parametrize("idx,case", list(enumerate(_cases())))
def test_random_case_bit_exact(idx, case):
    from oracle import oracle
    from selfpose3d_amd import _lib, synthetic as syn
    from selfpose3d_amd.camera_pack import pack_cameras
    dev = torch.device("cuda:0")
    B, V, J, (w, h), cube = case
    rng = np.random.default_rng(1000 + idx)
    img = (w * 4, h * 4)
    meta = syn.random_meta(B, V, img, seed=idx, augment=(idx % 2 == 0), ssv_style=(idx % 3 == 0))
    flip = torch.from_numpy(rng.random(B) < 0.4) if idx % 2 == 0 else None
    cam = pack_cameras(meta, B, img, flip)
    hms = [torch.from_numpy((rng.random((B, J, h, w), dtype=np.float32) * 1.6 - 0.3)) for _ in range(V)]
    fine = idx % 2 == 1
    if fine:
        centers = np.stack([rng.uniform(-2500, 2500, B), rng.uniform(-3000, 2000, B), rng.uniform(0, 1800, B)], 1).astype(np.float32)
        gs = [float(rng.uniform(500, 3000))] * 3
    else:
        centers = np.repeat(np.asarray([syn.SPACE_CENTER], np.float32), B, 0)
        gs = list(syn.SPACE_SIZE)
    valid = (rng.random(B) < 0.8).astype(np.uint8)
    valid[0] = 1
    ref_c, ref_g = oracle.unproject_fwd([x.numpy() for x in hms], cam, centers, valid, gs, cube, img)
    d_h = [x.to(dev) for x in hms]
    camd, cen, val = torch.from_numpy(cam).to(dev), torch.from_numpy(centers).to(dev), torch.from_numpy(valid).to(dev)
    got, grids = _lib.unproject_fwd(d_h, _lib.LAYOUT_PLANAR, 0, camd, cen, val, B, J, h, w, cube, gs, img, True)
    assert np.array_equal(grids.cpu().numpy(), ref_g)
    assert np.array_equal(got.cpu().numpy(), ref_c), float(np.abs(got.cpu().numpy() - ref_c).max())
    if J <= 16:
        jp = 4 if J <= 4 else (8 if J <= 8 else (12 if J <= 12 else 16))
        packed = _lib.pack_heatmaps(d_h, jp=jp)
        views = [packed[c] for c in range(V)]
        for variant in (None, 1, 8, 24, 56, 120):  # default, block-synchronous, pipelined 4 waves/WG, 1 wave/WG, 4x4x4 bricks
            got, grids = _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, jp, camd, cen, val, B, J, h, w, cube, gs, img, True,
                                            variant=variant)
            assert np.array_equal(grids.cpu().numpy(), ref_g)
            assert np.array_equal(got.cpu().numpy(), ref_c), (variant, float(np.abs(got.cpu().numpy() - ref_c).max()))
