"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed
reference goldens.  Run on the MI355X box: python -m pytest tests -m gpu

Tolerances: north_star asks <=1e-4 on voxel values and bit-exact integer indices.  The kernels
follow the oracle's fp32 operation order exactly, so the assertions below are much tighter
(VOX_TOL); grids and NMS indices must be bit-exact.
"""
import numpy as np
import pytest
import torch

from tests import golden_io as gio

pytestmark = pytest.mark.gpu

VOX_TOL = 1e-6


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _oracle_fwd(case):
    from oracle import oracle
    return oracle.unproject_fwd([h.numpy() for h in case.hms], case.cam, case.centers, case.valid, case.grid_size,
                                case.cube, case.img)


def _hip_fwd(case, dev, layout, want_grids=True, variant=None):
    from selfpose3d_amd import _lib
    hms = [h.to(dev) for h in case.hms]
    cam = torch.from_numpy(case.cam).to(dev)
    centers = torch.from_numpy(case.centers).to(dev)
    valid = torch.from_numpy(case.valid).to(dev)
    w, h = case.hm
    if layout == "planar":
        return _lib.unproject_fwd(hms, _lib.LAYOUT_PLANAR, 0, cam, centers, valid, case.B, case.J, h, w, case.cube,
                                  case.grid_size, case.img, want_grids)
    jp = 4 if case.J <= 4 else (8 if case.J <= 8 else (12 if case.J <= 12 else 16))
    packed = _lib.pack_heatmaps(hms, jp=jp)
    views = [packed[c] for c in range(case.V)]
    return _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, jp, cam, centers, valid, case.B, case.J, h, w, case.cube,
                              case.grid_size, case.img, want_grids, variant=variant)


def test_pack_heatmaps(dev):
    from selfpose3d_amd import _lib
    rng = np.random.default_rng(0)
    for (B, V, J, h, w, jp) in [(2, 3, 15, 18, 24, 16), (1, 1, 1, 7, 5, 4), (3, 5, 7, 33, 65, 8), (1, 2, 12, 16, 16, 12)]:
        hms = [torch.from_numpy(rng.standard_normal((B, J, h, w)).astype(np.float32)).to(dev) for _ in range(V)]
        packed = _lib.pack_heatmaps(hms, jp=jp).cpu()
        assert packed.shape == (V, B, h, w, jp)
        for c in range(V):
            exp = hms[c].cpu().permute(0, 2, 3, 1)
            assert torch.equal(packed[c, ..., :J], exp)
            assert torch.count_nonzero(packed[c, ..., J:]) == 0


@pytest.mark.parametrize("layout", ["planar", "nhwc"])
@pytest.mark.parametrize("name", gio.SMALL_CASES + gio.FULL_CASES)
def test_unproject_fwd_vs_oracle_and_golden(dev, name, layout):
    case = gio.Case(name)
    cubes, grids = _hip_fwd(case, dev, layout)
    cubes = cubes.cpu().numpy()
    grids = grids.cpu().numpy()
    ref_c, ref_g = _oracle_fwd(case)
    assert np.array_equal(grids, ref_g), "grids must be bit-exact vs oracle"
    d = np.abs(cubes - ref_c)
    assert d.max() <= VOX_TOL, (name, layout, float(d.max()))
    # committed reference goldens (sub-sampled for the full-size cases) + whole-volume checksums
    exp_c, exp_g, idx = case.expected()
    got_c = cubes.reshape(case.B, case.J, case.N)
    got_g = grids
    if idx is not None:
        got_c, got_g = got_c[:, :, idx], grids[:, idx]
    assert np.array_equal(got_g, exp_g)
    assert np.abs(got_c - exp_c).max() <= VOX_TOL
    assert abs(cubes.astype(np.float64).sum() - float(case.g["cubes_sum"])) <= 1e-7 * cubes.size
    print(f"{name}/{layout}: max|d| vs oracle {d.max():.3e}, bit-equal {float((cubes == ref_c).mean()):.6f}")


@pytest.mark.parametrize("variant", [0, 1, 2, 4, 5, 6, 8, 12, 24, 28, 24 | (1 << 21), 24 | (1 << 17), 24 | (7 << 17),
                                     56, 56 | (1 << 21), 56 | (1 << 17), 56 | (5 << 17), 120, 120 | (1 << 17)])
def test_nhwc_variants_bit_identical(dev, variant):
    case = gio.Case("unproj_coarse_full_96x72")
    base, _ = _hip_fwd(case, dev, "planar")
    got, _ = _hip_fwd(case, dev, "nhwc", variant=variant)
    assert torch.equal(base, got)


def test_no_grids_and_invalid_rows(dev):
    case = gio.Case("unproj_fine_small")          # row 1 is invalid (flag < 0)
    for layout in ("planar", "nhwc"):
        cubes, grids = _hip_fwd(case, dev, layout, want_grids=False)
        assert grids is None
        assert torch.count_nonzero(cubes[1]) == 0
        cubes2, grids2 = _hip_fwd(case, dev, layout, want_grids=True)
        assert torch.equal(cubes, cubes2)
        assert torch.count_nonzero(grids2[1]) == 0


def test_properties_full_size(dev):
    """size-independent properties at BASELINE sizes (B=4, 5 views, 240x128, 80x80x20)."""
    from selfpose3d_amd import _lib, synthetic as syn
    from selfpose3d_amd.camera_pack import pack_cameras
    B, V, J, h, w = 4, 5, 15, 128, 240
    img = (960, 512)
    meta = syn.make_meta(B, V, img)
    cam = torch.from_numpy(pack_cameras(meta, B, img)).to(dev)
    centers = torch.tensor([syn.SPACE_CENTER] * B, dtype=torch.float32, device=dev)
    valid = torch.ones(B, dtype=torch.uint8, device=dev)
    cube, gs = syn.INITIAL_CUBE_SIZE, syn.SPACE_SIZE

    def run(hms, layout="nhwc"):
        if layout == "planar":
            return _lib.unproject_fwd(hms, _lib.LAYOUT_PLANAR, 0, cam, centers, valid, B, J, h, w, cube, gs, img, False)[0]
        p = _lib.pack_heatmaps(hms, jp=16)
        return _lib.unproject_fwd([p[c] for c in range(V)], _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, J, h, w,
                                  cube, gs, img, False)[0]

    a = [0.5 * x.to(dev) for x in syn.random_heatmaps(B, V, J, h, w, seed=100)]
    b = [0.5 * x.to(dev) for x in syn.random_heatmaps(B, V, J, h, w, seed=101)]
    ua, ub = run(a), run(b)
    # 1. the two independent kernels agree bit for bit
    assert torch.equal(ua, run(a, "planar"))
    # 2. linearity while the clamp is inactive (values < 1): u(a+b) == u(a)+u(b) up to fp32 rounding
    uab = run([x + y for x, y in zip(a, b)])
    assert float((uab - (ua + ub)).abs().max()) <= 5e-6
    # 3. zero in, zero out; constant c in => c wherever all taps are inside (never above c)
    z = [torch.zeros_like(x) for x in a]
    assert torch.count_nonzero(run(z)) == 0
    c = [torch.full_like(x, 0.25) for x in a]
    uc = run(c)
    assert float(uc.max()) <= 0.25 + 1e-6
    assert float((uc - 0.25).abs().median()) <= 1e-6
    # 4. channels are independent: channel j of the J=15 result == the J=1 run on channel j alone
    j = 7
    a1 = [x[:, j:j + 1].contiguous() for x in a]
    p1 = _lib.pack_heatmaps(a1, jp=4)
    u1 = _lib.unproject_fwd([p1[c] for c in range(V)], _lib.LAYOUT_NHWC, 4, cam, centers, valid, B, 1, h, w, cube, gs,
                            img, False)[0]
    assert torch.equal(u1[:, 0], ua[:, j])
    # 5. samples are independent: permuting the batch permutes the output
    perm = [2, 0, 3, 1]
    up = run([x[perm].contiguous() for x in a])
    assert torch.equal(up, ua[perm])
    # 6. range
    assert float(ua.min()) >= 0.0 and float(ua.max()) <= 1.0


@pytest.mark.parametrize("name", ["unproj_grad_small", "unproj_grad_fine_aug"])
def test_unproject_bwd(dev, name):
    from oracle import oracle
    from selfpose3d_amd import _lib
    case = gio.Case(name)
    g = case.g
    wgt = np.random.default_rng(int(g["grad_seed"])).standard_normal((case.B, case.J, *case.cube)).astype(np.float32)
    hms = [h.to(dev) for h in case.hms]
    grads = _lib.unproject_bwd(hms, torch.from_numpy(case.cam).to(dev), torch.from_numpy(case.centers).to(dev),
                               torch.from_numpy(case.valid).to(dev), torch.from_numpy(wgt).to(dev), case.cube,
                               case.grid_size, case.img)
    ref = oracle.unproject_bwd([h.numpy() for h in case.hms], case.cam, case.centers, case.valid, wgt, case.grid_size,
                               case.cube, case.img)
    gold = g["grad_hm"]
    for c in range(case.V):
        got = grads[c].cpu().numpy()
        scale = max(1.0, float(np.abs(ref[c]).max()))
        assert np.abs(got - ref[c]).max() <= 2e-5 * scale
        assert np.abs(got - gold[c]).max() <= 2e-5 * scale


def test_autograd_through_project_layer(dev):
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.project_layer import ProjectLayer
    case = gio.Case("unproj_grad_fine_aug")
    g = case.g
    cfg = load_config(None, NETWORK__IMAGE_SIZE=case.img, NETWORK__HEATMAP_SIZE=case.hm)
    layer = ProjectLayer(cfg)
    hms = [h.to(dev).requires_grad_(True) for h in case.hms]
    cubes, grids = layer(hms, case.meta, case.grid_size, case.grid_center.to(dev), case.cube,
                         flip_xcoords=case.flip)
    assert not grids.requires_grad
    wgt = torch.from_numpy(np.random.default_rng(int(g["grad_seed"])).standard_normal(
        tuple(cubes.shape)).astype(np.float32)).to(dev)
    (cubes * wgt).sum().backward()
    exp_c, exp_g, _ = case.expected()
    assert np.abs(cubes.detach().cpu().numpy().reshape(exp_c.shape) - exp_c).max() <= VOX_TOL
    for c in range(case.V):
        scale = max(1.0, float(np.abs(g["grad_hm"][c]).max()))
        assert np.abs(hms[c].grad.cpu().numpy() - g["grad_hm"][c]).max() <= 2e-5 * scale


@pytest.mark.parametrize("name", ["unproj_coarse_small", "unproj_coarse_aug", "unproj_fine_small"])
@pytest.mark.parametrize("mode", ["planar", "nhwc"])
def test_project_layer_module_reference_api(dev, name, mode):
    """through the reference's own call signature: list of heat-maps + collated meta dicts"""
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.project_layer import ProjectLayer
    case = gio.Case(name)
    cfg = load_config(None, NETWORK__IMAGE_SIZE=case.img, NETWORK__HEATMAP_SIZE=case.hm)
    layer = ProjectLayer(cfg, mode=mode)
    gc = case.grid_center if isinstance(case.grid_center, list) else case.grid_center.to(dev)
    cubes, grids = layer([h.to(dev) for h in case.hms], case.meta, case.grid_size, gc, case.cube,
                         flip_xcoords=case.flip)
    exp_c, exp_g, _ = case.expected()
    assert cubes.shape == (case.B, case.J, *case.cube) and grids.shape == (case.B, case.N, 3)
    assert np.array_equal(grids.cpu().numpy(), exp_g)
    assert np.abs(cubes.cpu().numpy().reshape(exp_c.shape) - exp_c).max() <= VOX_TOL


def test_nms_topk(dev):
    from oracle import oracle
    from selfpose3d_amd import _lib, synthetic as syn
    g = gio.load("nms")
    rnd = np.random.default_rng(int(g["rnd_seed"])).random(tuple(g["rnd_shape"]), dtype=np.float32)
    vals, idx, locs = _lib.nms_topk(torch.from_numpy(rnd).to(dev), 10, [8000.0, 8000.0, 2000.0], [0.0, -500.0, 800.0])
    assert np.array_equal(vals.cpu().numpy(), g["rnd_vals"])
    assert np.array_equal(idx.cpu().numpy(), g["rnd_idx"])
    # people case at full coarse size, against the oracle and the reference golden
    case = gio.Case("unproj_people_coarse")
    cubes, _ = _hip_fwd(case, dev, "nhwc", want_grids=False)
    root = cubes[:, 2].contiguous()
    vals, idx, locs = _lib.nms_topk(root, 10, syn.SPACE_SIZE, syn.SPACE_CENTER)
    rv, ri = oracle.nms_topk(root.cpu().numpy(), 10)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)
    pos = g["people_vals"] > 0
    assert np.array_equal(idx.cpu().numpy()[pos], g["people_idx"][pos])
    assert np.abs(vals.cpu().numpy() - g["people_vals"]).max() <= VOX_TOL
    # index -> mm (cuboid_proposal_net.py:42-52) in fp32
    cs = torch.tensor(case.cube, dtype=torch.float32)
    gs = torch.tensor(syn.SPACE_SIZE)
    gcen = torch.tensor(syn.SPACE_CENTER)
    exp_loc = idx.cpu().float() / (cs - 1) * gs + gcen - gs / 2.0
    assert torch.equal(locs.cpu(), exp_loc)
    # ties: constant volume -> every voxel is a local max, lowest flat indices win in order
    flat = torch.full((1, 6, 5, 4), 0.5, device=dev)
    vals, idx, _ = _lib.nms_topk(flat, 7)
    exp = torch.tensor([[n // 20, (n % 20) // 4, n % 4] for n in range(7)])
    assert torch.equal(idx.cpu()[0], exp) and torch.all(vals == 0.5)
    # empty-ish: all zeros and k larger than the number of positive peaks
    zero = torch.zeros((2, 8, 8, 4), device=dev)
    vals, idx, _ = _lib.nms_topk(zero, 10)
    assert torch.count_nonzero(vals) == 0


def test_nms_batches_merge_passes_and_graph_replay(dev):
    """several samples, a volume whose candidates need several merge passes, k = 32, many tiny samples, repeated calls, and
    replay from a HIP graph - all bit-equal to the oracle"""
    from oracle import oracle
    from selfpose3d_amd import _lib
    rng = np.random.default_rng(11)
    for shape, k in (((3, 80, 80, 20), 10), ((2, 160, 160, 40), 10), ((1, 13, 9, 70), 32), ((260, 8, 8, 4), 5)):
        x = rng.random(shape, dtype=np.float32)
        x[x < 0.7] = 0.0
        rv, ri = oracle.nms_topk(x, k)
        xd = torch.from_numpy(x).to(dev)
        for _ in range(3):
            vals, idx, _ = _lib.nms_topk(xd, k)
        assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv), shape
    x = rng.random((2, 80, 80, 20), dtype=np.float32)
    xd = torch.from_numpy(x).to(dev)
    ref = _lib.nms_proposals(xd, 10, [8000.0, 8000.0, 2000.0], [0.0, -500.0, 800.0], 0.3).clone()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        _lib.nms_proposals(xd, 10, [8000.0, 8000.0, 2000.0], [0.0, -500.0, 800.0], 0.3)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            out = _lib.nms_proposals(xd, 10, [8000.0, 8000.0, 2000.0], [0.0, -500.0, 800.0], 0.3)
    for _ in range(5):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
    rv, ri = oracle.nms_topk(x, 10)
    assert np.array_equal(out[:, :, 4].cpu().numpy(), rv)


def test_soft_argmax(dev):
    from oracle import oracle
    from selfpose3d_amd import _lib
    rng = np.random.default_rng(5)
    for (Bv, J, n) in [(2, 3, 16), (1, 15, 64)]:
        x = rng.random((Bv, J, n, n, n), dtype=np.float32) * 0.3
        for b in range(Bv):
            for j in range(J):
                x[b, j, rng.integers(n), rng.integers(n), rng.integers(n)] = 0.9
        grids = (rng.random((Bv, n ** 3, 3), dtype=np.float32) - 0.5) * 2000.0
        out = _lib.soft_argmax(torch.from_numpy(x).to(dev), torch.from_numpy(grids).to(dev), 100.0).cpu().numpy()
        ref = oracle.soft_argmax(x, grids, 100.0)
        assert np.abs(out - ref).max() <= 2e-2, float(np.abs(out - ref).max())   # mm, on +-1000 mm coordinates


def test_rootnet_posenet_vs_reference_golden(dev):
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.cuboid_proposal_net import CuboidProposalNet
    from selfpose3d_amd.pose_regression_net import PoseRegressionNet
    g = gio.load("rootnet_posenet")
    B, V, J = int(g["B"]), int(g["V"]), int(g["J"])
    img, hm = [int(v) for v in g["img"]], [int(v) for v in g["hm"]]
    cfg = load_config(None, NETWORK__IMAGE_SIZE=img, NETWORK__HEATMAP_SIZE=hm, NETWORK__NUM_JOINTS=J,
                      MULTI_PERSON__INITIAL_CUBE_SIZE=[int(v) for v in g["cube"]],
                      PICT_STRUCT__CUBE_SIZE=[int(v) for v in g["fine_cube"]],
                      MULTI_PERSON__THRESHOLD=float(g["threshold"]))
    meta = syn.make_meta(B, V, img)
    hms, _ = syn.people_heatmaps(B, V, J, hm[1], hm[0], img, seed=int(g["hm_seed"]))
    hms = [h.to(dev) for h in hms]
    rootnet = CuboidProposalNet(cfg)
    assert sorted(rootnet.state_dict().keys()) == list(g["root_keys"])
    syn.fill_parameters_deterministic(rootnet, seed=int(g["root_seed"]), scale=float(g["param_scale"]))
    rootnet.eval().to(dev)
    posenet = PoseRegressionNet(cfg)
    assert sorted(posenet.state_dict().keys()) == list(g["pose_keys"])
    syn.fill_parameters_deterministic(posenet, seed=int(g["pose_seed"]), scale=float(g["param_scale"]))
    posenet.eval().to(dev)
    prev = torch.backends.cudnn.allow_tf32
    with torch.no_grad():
        root_cubes, grid_centers = rootnet(hms, meta)
        assert float((root_cubes.cpu() - torch.from_numpy(g["root_cubes"])).abs().max()) <= 2e-4
        gc_ref = torch.from_numpy(g["grid_centers"])
        # compare proposal slots whose score is separated from its neighbours (MIOpen vs CPU conv rounding)
        sc = gc_ref[:, :, 4]
        for b in range(B):
            for k in range(sc.shape[1]):
                gap = min(abs(float(sc[b, k] - sc[b, k - 1])) if k > 0 else 1.0,
                          abs(float(sc[b, k] - sc[b, k + 1])) if k + 1 < sc.shape[1] else 1.0)
                if gap > 1e-3:
                    assert torch.equal(grid_centers[b, k, :4].cpu(), gc_ref[b, k, :4]), (b, k)
                    assert abs(float(grid_centers[b, k, 4].cpu() - gc_ref[b, k, 4])) <= 2e-4
        for n in range(g["preds"].shape[0]):
            pred = posenet(hms, meta, gc_ref[:, n].to(dev))
            assert float((pred.cpu() - torch.from_numpy(g["preds"][n])).abs().max()) <= 0.5   # mm
    torch.backends.cudnn.allow_tf32 = prev


def test_padded_and_channels_last_outputs(dev):
    """pad_channels / channels_last are layout-only options: same voxel bits, zeros in the padding"""
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.project_layer import ProjectLayer
    for name in ("unproj_coarse_full_96x72", "unproj_fine_small", "unproj_coarse_aug"):
        case = gio.Case(name)
        cfg = load_config(None, NETWORK__IMAGE_SIZE=case.img, NETWORK__HEATMAP_SIZE=case.hm)
        layer = ProjectLayer(cfg)
        gc = case.grid_center if isinstance(case.grid_center, list) else case.grid_center.to(dev)
        hms = [h.to(dev) for h in case.hms]
        base, _ = layer.get_voxel(hms, case.meta, case.grid_size, gc, case.cube, flip_xcoords=case.flip)
        for pad, cl in ((True, False), (True, True)):
            got, _ = layer.get_voxel(hms, case.meta, case.grid_size, gc, case.cube, flip_xcoords=case.flip,
                                     want_grids=False, pad_channels=pad, channels_last=cl)
            jp = ProjectLayer.jp_for(case.J)
            assert got.shape == (case.B, jp, *case.cube)
            if cl:
                assert got.is_contiguous(memory_format=torch.channels_last_3d)
            assert torch.equal(got[:, :case.J], base)
            assert torch.count_nonzero(got[:, case.J:]) == 0


def test_rootnet_channels_last_matches_default(dev):
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.cuboid_proposal_net import CuboidProposalNet
    cfg = load_config(None, NETWORK__IMAGE_SIZE=[384, 288], NETWORK__HEATMAP_SIZE=[96, 72],
                      MULTI_PERSON__INITIAL_CUBE_SIZE=[24, 24, 8])
    B, V, J = 2, 5, 15
    meta = syn.make_meta(B, V, (384, 288))
    hms, _ = syn.people_heatmaps(B, V, J, 72, 96, (384, 288), seed=5)
    hms = [h.to(dev) for h in hms]
    net = CuboidProposalNet(cfg)
    syn.fill_parameters_deterministic(net, seed=3, scale=0.05)
    net.eval().to(dev)
    with torch.no_grad():
        rc0, gc0 = net(hms, meta)
        net.use_channels_last(True)
        rc1, gc1 = net(hms, meta)
    assert float((rc0 - rc1).abs().max()) <= 2e-4
    assert torch.equal(gc0[:, :3, :3], gc1[:, :3, :3])


def test_indexed_unprojection_and_batched_posenet(dev):
    """all proposals of a batch in one launch == the reference's per-candidate loop"""
    from selfpose3d_amd import _lib, synthetic as syn
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.pose_regression_net import PoseRegressionNet
    from selfpose3d_amd.project_layer import ProjectLayer, clear_pack_cache
    img, hm, J, B, V, K = [384, 288], [96, 72], 15, 3, 5, 4
    cfg = load_config(None, NETWORK__IMAGE_SIZE=img, NETWORK__HEATMAP_SIZE=hm, PICT_STRUCT__CUBE_SIZE=[16, 16, 16])
    meta = syn.make_meta(B, V, img, rotations=[0.0, 10.0, -15.0], scale_mults=[1.0, 1.1, 0.9], ssv_style=True)
    flip = torch.tensor([False, True, False])
    hms, _ = syn.people_heatmaps(B, V, J, hm[1], hm[0], img, seed=77)
    hms = [h.to(dev) for h in hms]
    rng = np.random.default_rng(3)
    gc = np.zeros((B, K, 5), np.float32)
    gc[:, :, 0] = rng.uniform(-1500, 1500, (B, K)); gc[:, :, 1] = rng.uniform(-2000, 1000, (B, K))
    gc[:, :, 2] = rng.uniform(700, 1100, (B, K)); gc[:, :, 3] = rng.integers(-1, 2, (B, K)); gc[:, :, 4] = 0.9
    gc[0, 0, 3] = 0; gc[1, :, 3] = -1                                  # sample 1 has no valid proposal
    gct = torch.from_numpy(gc).to(dev)
    layer = ProjectLayer(cfg)
    pairs = torch.nonzero(gct[:, :, 3] >= 0)
    cubes_idx, grids_idx = layer.get_voxel(hms, meta, syn.FINE_GRID_SIZE, gct[pairs[:, 0], pairs[:, 1], :3].contiguous(),
                                           [16, 16, 16], flip_xcoords=flip, sample_of=pairs[:, 0])
    for p, (b, k) in enumerate(pairs.tolist()):
        ref_c, ref_g = layer.get_voxel(hms, meta, syn.FINE_GRID_SIZE, gct[:, k], [16, 16, 16], flip_xcoords=flip)
        assert torch.equal(cubes_idx[p], ref_c[b]) and torch.equal(grids_idx[p], ref_g[b])
    # whole pose net: batched == loop
    net = PoseRegressionNet(cfg)
    syn.fill_parameters_deterministic(net, seed=9, scale=0.05)
    net.eval().to(dev)
    with torch.no_grad():
        loop = torch.stack([net(hms, meta, gct[:, k], flip_xcoords=flip) for k in range(K)], 1)
        batched = net.forward_batched(hms, meta, gct, flip_xcoords=flip, max_cubes_per_call=3)
    assert float((loop - batched).abs().max()) <= 0.5           # mm on +-2000 mm coords; MIOpen picks per-batch-size conv algos
    assert torch.count_nonzero(batched[1]) == 0
    # soft-argmax with in-kernel grids == with materialised grids
    x = torch.rand(2, J, 16, 16, 16, device=dev)
    cen = gct[pairs[:2, 0], pairs[:2, 1], :3].contiguous()
    _, grids = layer.get_voxel(hms, meta, syn.FINE_GRID_SIZE, cen, [16, 16, 16], sample_of=pairs[:2, 0])
    a = _lib.soft_argmax(x, grids, 100.0)
    b2 = _lib.soft_argmax_grid(x, cen, syn.FINE_GRID_SIZE, [16, 16, 16], 100.0)
    assert torch.equal(a, b2)
    clear_pack_cache()


def test_pack_cache_never_serves_stale_data(dev):
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.project_layer import ProjectLayer, _PACK_CACHE
    case = gio.Case("unproj_coarse_small")
    cfg = load_config(None, NETWORK__IMAGE_SIZE=case.img, NETWORK__HEATMAP_SIZE=case.hm)
    layer = ProjectLayer(cfg)
    hms = [h.to(dev) for h in case.hms]
    a, _ = layer(hms, case.meta, case.grid_size, case.grid_center, case.cube)
    n = len(_PACK_CACHE)
    b, _ = layer(hms, case.meta, case.grid_size, case.grid_center, case.cube)     # hit
    assert len(_PACK_CACHE) == n and torch.equal(a, b)
    hms[0].mul_(0.5)                                                              # in-place edit: version bump
    c, _ = layer(hms, case.meta, case.grid_size, case.grid_center, case.cube)
    assert not torch.equal(a, c)
    new = [h.clone() for h in hms]                                                # same values, new objects
    d, _ = layer(new, case.meta, case.grid_size, case.grid_center, case.cube)
    assert torch.equal(c, d)


@pytest.mark.parametrize("cl", [False, True])
def test_v2v_fused_inference_plan_matches_module(dev, cl):
    """BatchNorm-folded convs + fused HIP epilogues == the plain module (fp32 rounding only)"""
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.v2v_net import V2VNet
    for cin, cout, shape in ((15, 1, (2, 15, 16, 16, 8)), (15, 15, (1, 16, 16, 16, 16)), (1, 1, (2, 1, 8, 8, 8))):
        net = V2VNet(cin, cout)
        syn.fill_parameters_deterministic(net, seed=21, scale=0.05)
        net.eval().to(dev)
        x = torch.rand(shape, device=dev)
        if cl:
            net.to(memory_format=torch.channels_last_3d)
            x = x.contiguous(memory_format=torch.channels_last_3d)
        with torch.no_grad():
            net.fused_inference = False
            ref = net(x)
            net.fused_inference = True
            got = net(x)
            assert net._plan is not None and net._plan.key is not None
            assert float((ref - got).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
            # parameters change -> the folded plan is rebuilt
            net.output_layer.bias.add_(1.0)
            net.front_layers[0].block[1].running_mean.add_(0.1)
            net.fused_inference = False
            ref2 = net(x)
            net.fused_inference = True
            got2 = net(x)
            assert float((ref2 - got2).abs().max()) <= 1e-4 * max(1.0, float(ref2.abs().max()))
            assert float((ref2 - ref).abs().max()) > 1e-3


def test_channel_shift_act_modes(dev):
    from selfpose3d_amd import _lib
    for cl in (False, True):
        y = torch.randn(2, 8, 4, 4, 4, device=dev)
        r = torch.randn_like(y)
        if cl:
            y = y.contiguous(memory_format=torch.channels_last_3d)
            r = r.contiguous(memory_format=torch.channels_last_3d)
        s = torch.randn(8, device=dev)
        sv = s.view(1, -1, 1, 1, 1)
        exp = [y + sv, torch.relu(y + sv), torch.relu(y + sv + r), torch.relu(y + sv) + r]
        for mode in range(4):
            got = _lib.channel_shift_act_(y.clone(memory_format=torch.preserve_format), s, mode, r if mode >= 2 else None)
            assert torch.equal(got, exp[mode]), (cl, mode)


@pytest.mark.parametrize("V", [3, 4])
def test_bf16_storage_config(dev, V):
    """BASELINE configs[4]: 3-4 views, fine 64^3 per-person voxels, heat-maps / cubes stored as bf16, math fp32.
    Parity definition: oracle run on the bf16-rounded heat-maps, its fp32 result rounded to bf16 (RNE)."""
    from oracle import oracle
    from selfpose3d_amd import _lib, synthetic as syn
    from selfpose3d_amd.camera_pack import pack_cameras
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.project_layer import ProjectLayer
    B, J, img, hm, cube = 2, 15, (384, 288), (96, 72), (64, 64, 64)
    cfg = load_config(None, NETWORK__IMAGE_SIZE=list(img), NETWORK__HEATMAP_SIZE=list(hm))
    meta = syn.make_meta(B, V, img)
    hms32 = syn.random_heatmaps(B, V, J, hm[1], hm[0], seed=50 + V)
    hms16 = [h.to(torch.bfloat16) for h in hms32]
    gc = torch.tensor([[300.0, -800.0, 900.0, 0.0, 0.9], [-700.0, 100.0, 1000.0, 1.0, 0.8]])
    cam = pack_cameras(meta, B, img)
    ref, ref_g = oracle.unproject_fwd([h.float().numpy() for h in hms16], cam, gc[:, :3].numpy(), np.ones(B, np.uint8),
                                      syn.FINE_GRID_SIZE, cube, img)
    ref16 = torch.from_numpy(ref).to(torch.bfloat16)
    layer = ProjectLayer(cfg, io_dtype=torch.bfloat16)
    for src in (hms16, hms32):                    # bf16 producer, or fp32 producer rounded inside the pack kernel
        cubes, grids = layer([h.to(dev) for h in src], meta, syn.FINE_GRID_SIZE, gc.to(dev), list(cube))
        assert cubes.dtype == torch.bfloat16 and grids.dtype == torch.float32
        assert torch.equal(cubes.cpu(), ref16)
        assert np.array_equal(grids.cpu().numpy(), ref_g)
    # bf16 in, fp32 out and channels-last padded bf16 out
    packed = _lib.pack_heatmaps([h.to(dev) for h in hms16], jp=16, out_dtype=torch.bfloat16)
    camd, cen, val = torch.from_numpy(cam).to(dev), gc[:, :3].contiguous().to(dev), torch.ones(B, dtype=torch.uint8, device=dev)
    c32, _ = _lib.unproject_fwd([packed[c] for c in range(V)], _lib.LAYOUT_NHWC, 16, camd, cen, val, B, J, hm[1], hm[0],
                                cube, syn.FINE_GRID_SIZE, img, False)
    assert torch.equal(c32.cpu(), torch.from_numpy(ref))
    ccl, _ = _lib.unproject_fwd([packed[c] for c in range(V)], _lib.LAYOUT_NHWC, 16, camd, cen, val, B, 16, hm[1], hm[0],
                                cube, syn.FINE_GRID_SIZE, img, False, channels_last=True, out_dtype=torch.bfloat16)
    assert ccl.is_contiguous(memory_format=torch.channels_last_3d)
    assert torch.equal(ccl[:, :J].cpu(), ref16) and torch.count_nonzero(ccl[:, J:]) == 0


@pytest.mark.parametrize("name", ["unproj_grad_small", "unproj_grad_fine_aug"])
def test_unproject_bwd_packed_line_coalesced(dev, name):
    """pass mask from the forward kernel + channels-last scatter == oracle / reference autograd"""
    from oracle import oracle
    from selfpose3d_amd import _lib
    case = gio.Case(name)
    g = case.g
    wgt = np.random.default_rng(int(g["grad_seed"])).standard_normal((case.B, case.J, *case.cube)).astype(np.float32)
    hms = [h.to(dev) for h in case.hms]
    cam, cen = torch.from_numpy(case.cam).to(dev), torch.from_numpy(case.centers).to(dev)
    val = torch.from_numpy(case.valid).to(dev)
    w, h = case.hm
    jp = 4 if case.J <= 4 else 16
    packed = _lib.pack_heatmaps(hms, jp=jp)
    mask = torch.empty((case.B, case.N), dtype=torch.int16, device=dev)
    cubes, _ = _lib.unproject_fwd([packed[c] for c in range(case.V)], _lib.LAYOUT_NHWC, jp, cam, cen, val, case.B, case.J,
                                  h, w, case.cube, case.grid_size, case.img, False, pass_mask=mask)
    exp_c, _, _ = case.expected()
    assert np.abs(cubes.cpu().numpy().reshape(exp_c.shape) - exp_c).max() <= VOX_TOL
    grads = _lib.unproject_bwd_packed(cam, cen, val, torch.from_numpy(wgt).to(dev), mask, case.B, case.V, case.J, jp, h, w,
                                      case.cube, case.grid_size, case.img)
    ref = oracle.unproject_bwd([x.numpy() for x in case.hms], case.cam, case.centers, case.valid, wgt, case.grid_size,
                               case.cube, case.img)
    for c in range(case.V):
        got = grads[c].cpu().numpy()
        scale = max(1.0, float(np.abs(ref[c]).max()))
        assert got.shape == ref[c].shape
        assert np.abs(got - ref[c]).max() <= 2e-5 * scale
        assert np.abs(got - g["grad_hm"][c]).max() <= 2e-5 * scale
    # the pass mask agrees with the pre-clamp values: clamped-at-1 voxels block the gradient
    m = mask.cpu().numpy().astype(np.uint16)
    o = cubes.cpu().numpy().reshape(case.B, case.J, case.N)
    for j in range(case.J):
        inside = (o[:, j] > 0) & (o[:, j] < 1)
        assert np.all(((m >> j) & 1)[inside] == 1)


@pytest.mark.parametrize("layout", ["planar", "nhwc"])
def test_registered_torch_op_forward_and_autograd(dev, layout):
    """torch.ops.selfpose3d_mi.unproject_fwd/bwd: same bits as the oracle, gradients = reference
    autograd golden, for the stacked planar (V,B,J,h,w) and the channels-last (V,B,h,w,16) input forms."""
    import selfpose3d_amd.torch_ops  # noqa: F401  (registers the ops)
    from selfpose3d_amd import _lib
    case = gio.Case("unproj_grad_fine_aug")
    g = case.g
    hms = [h.to(dev) for h in case.hms]
    if layout == "planar":
        hm = torch.stack(hms, 0)
    else:
        hm = _lib.pack_heatmaps(hms, jp=16)
    hm = hm.clone().requires_grad_(True)
    cam = torch.from_numpy(case.cam).to(dev)
    centers = torch.from_numpy(case.centers).to(dev)
    valid = torch.from_numpy(case.valid).to(dev)
    cubes, grids = torch.ops.selfpose3d_mi.unproject_fwd(hm, cam, centers, valid, [float(v) for v in case.grid_size],
                                                         list(case.cube), list(case.img), list(case.hm), case.J)
    o_c, o_g = _oracle_fwd(case)
    assert np.array_equal(cubes.detach().cpu().numpy().reshape(o_c.shape), o_c)
    assert np.array_equal(grids.cpu().numpy(), o_g) and not grids.requires_grad
    wgt = torch.from_numpy(np.random.default_rng(int(g["grad_seed"])).standard_normal(
        tuple(cubes.shape)).astype(np.float32)).to(dev)
    (cubes * wgt).sum().backward()
    assert hm.grad.shape == hm.shape
    for c in range(case.V):
        got = hm.grad[c] if layout == "planar" else hm.grad[c][..., :case.J].permute(0, 3, 1, 2)
        scale = max(1.0, float(np.abs(g["grad_hm"][c]).max()))
        assert np.abs(got.cpu().numpy() - g["grad_hm"][c]).max() <= 2e-5 * scale
    if layout == "nhwc":
        assert float(hm.grad[..., case.J:].abs().max()) == 0.0


@pytest.mark.parametrize("field", ["affine_inf", "affine_nan", "focal_nan"])
def test_non_finite_camera_rows_follow_the_reference(dev, field):
    """A non-finite camera / crop row makes every sample position of that view NaN; the reference then zeroes the
    voxel (project_layer.py:98).  The wave-level early-outs of the HIP kernel must not hide that."""
    from selfpose3d_amd.camera_pack import CAM_A, CAM_F, finish
    case = gio.Case("unproj_coarse_small")
    cam = case.cam.copy()
    if field == "affine_inf":
        cam[0, 1, CAM_A] = np.inf
    elif field == "affine_nan":
        cam[0, 1, CAM_A + 5] = np.nan
    else:
        cam[0, 0, CAM_F] = np.nan
    case.cam = finish(cam)                  # a record edited by hand: its derived fields follow (include/sp3d.h)
    ref_c, ref_g = _oracle_fwd(case)
    for layout in ("planar", "nhwc"):
        cubes, grids = _hip_fwd(case, dev, layout)
        assert np.array_equal(cubes.cpu().numpy().reshape(ref_c.shape), ref_c), layout
    assert float(np.abs(ref_c[0]).max()) == 0.0 or field == "affine_inf"


@pytest.mark.parametrize("shape", [(4, 15, 16, (12, 10, 8)), (1, 3, 5, (6, 6, 4)), (2, 16, 16, (8, 8, 8)), (7, 15, 16, (10, 6, 6))])
def test_freq_contract_and_fft_opening_conv(dev, shape):
    """sp3d_freq_contract == the complex einsum it replaces, and rFFT -> contraction -> irFFT == the direct 7x7x7
    convolution it stands for (float64 as the referee: the FFT route must not be further from it than the direct
    fp32 convolution is, plus a small margin)."""
    import torch.nn.functional as F
    from selfpose3d_amd import _lib
    from selfpose3d_amd.v2v_net import _FoldedV2V
    B, C, O, (X, Y, Z) = shape
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.rand((B, C, X, Y, Z), generator=g).to(dev)
    w = (torch.randn((O, C, 7, 7, 7), generator=g) * 0.05).to(dev)
    S = tuple(_FoldedV2V._fft_len(n + 6) for n in (X, Y, Z))
    Wf = torch.conj(torch.fft.rfftn(w, s=S, dim=(2, 3, 4))).resolve_conj().contiguous()
    xp = F.pad(x, (3, S[2] - Z - 3, 3, S[1] - Y - 3, 3, S[0] - X - 3))
    Xf = torch.fft.rfftn(xp, dim=(2, 3, 4))
    Yf = _lib.freq_contract(Xf, Wf)
    ref = torch.einsum("bcxyz,ocxyz->boxyz", Xf, Wf)
    assert float((Yf - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    # a lazily conjugated view must be honoured (data_ptr() alone would ignore the conj bit)
    Yf2 = _lib.freq_contract(Xf, torch.conj(torch.fft.rfftn(w, s=S, dim=(2, 3, 4))))
    assert torch.equal(Yf2, Yf)
    y = torch.fft.irfftn(Yf, s=S, dim=(2, 3, 4))[..., :X, :Y, :Z]
    ref64 = F.conv3d(x.double(), w.double(), padding=3)
    direct = F.conv3d(x, w, padding=3)
    err_fft = float((y.double() - ref64).abs().max())
    err_direct = float((direct.double() - ref64).abs().max())
    assert err_fft <= max(2.0 * err_direct, 2e-5), (err_fft, err_direct)


@pytest.mark.parametrize("shape", [(2, 15, 16, (12, 8, 8), True), (3, 4, 6, (8, 8, 4), False), (1, 16, 16, (8, 12, 8), True)])
def test_freq_domain_conv_autograd(dev, shape):
    """_FreqConv3d (the 7x7x7 opening conv, forward and backward in the frequency domain) against float64 autograd
    of the direct convolution: output, grad input, grad weight, grad bias."""
    import torch.nn.functional as F
    from selfpose3d_amd.v2v_net import _FreqConv3d
    B, C, O, (X, Y, Z), with_bias = shape
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.rand((B, C, X, Y, Z), generator=g).to(dev).requires_grad_(True)
    w = (torch.randn((O, C, 7, 7, 7), generator=g) * 0.05).to(dev).requires_grad_(True)
    b = (torch.randn((O,), generator=g)).to(dev).requires_grad_(True) if with_bias else None
    gy = torch.randn((B, O, X, Y, Z), generator=g).to(dev)
    y = _FreqConv3d.apply(x, w, b)
    y.backward(gy)
    x64, w64 = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    b64 = b.detach().double().requires_grad_(True) if with_bias else None
    y64 = F.conv3d(x64, w64, b64, padding=3)
    y64.backward(gy.double())
    tol = lambda ref: 2e-5 * max(1.0, float(ref.abs().max()))
    assert float((y.detach().double() - y64.detach()).abs().max()) <= tol(y64.detach())
    assert float((x.grad.double() - x64.grad).abs().max()) <= tol(x64.grad)
    assert float((w.grad.double() - w64.grad).abs().max()) <= tol(w64.grad)
    if with_bias:
        assert float((b.grad.double() - b64.grad).abs().max()) <= tol(b64.grad)
    # channels-last activations come back channels-last
    xcl = x.detach().contiguous(memory_format=torch.channels_last_3d)
    ycl = _FreqConv3d.apply(xcl, w.detach(), None)
    assert ycl.is_contiguous(memory_format=torch.channels_last_3d) or C == 1


@pytest.mark.parametrize("shape", [(4, 128, 128, (20, 20, 5), 1), (2, 64, 128, (6, 5, 3), 2), (1, 8, 4, (4, 4, 4), 0), (3, 16, 32, (7, 9, 2), 3)])
def test_winograd_conv3d_with_fused_epilogue(dev, shape):
    """sp3d_wino_input -> bmm -> sp3d_wino_output == conv3d(3x3x3, pad 1) + shift (+ residual) (+ ReLU), odd grid sizes
    included; float64 is the referee."""
    import torch.nn.functional as F
    from selfpose3d_amd import _lib
    B, C, O, (X, Y, Z), mode = shape
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn((B, C, X, Y, Z), generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn((O, C, 3, 3, 3), generator=g) * 0.05).to(dev)
    shift = torch.randn((O,), generator=g).to(dev)
    res = torch.randn((B, O, X, Y, Z), generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    U = _lib.wino_weights(w)
    y = _lib.wino_conv3d_(x, U, shift, mode, res if mode >= 2 else None)
    ref = F.conv3d(x.double(), w.double(), padding=1) + shift.double().view(1, O, 1, 1, 1)
    if mode == 2:
        ref = ref + res.double()
    if mode >= 1:
        ref = ref.clamp_min(0)
    if mode == 3:
        ref = ref + res.double()
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last_3d)
    assert float((y.double() - ref).abs().max()) <= 5e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("shape", [(2, 32, (16, 16, 8), 2), (1, 16, (8, 8, 4), 1), (3, 32, (9, 7, 5), 0), (1, 32, (20, 12, 6), 3), (4, 16, (10, 17, 3), 1),
                                   (2, 32, (9, 7, 5), 2)])          # ragged grid + residual: clamped residual addresses (round 5)
def test_winograd_fused_kernel(dev, shape):
    """sp3d_wino_fused (one launch: on-the-fly transforms + v_mfma_f32_32x32x2_f32 + epilogue) == conv3d + epilogue,
    block-edge and odd sizes included; float64 referee."""
    import torch.nn.functional as F
    from selfpose3d_amd import _lib
    B, C, (X, Y, Z), mode = shape
    O = 32
    g = torch.Generator(device="cpu").manual_seed(9)
    x = torch.randn((B, C, X, Y, Z), generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn((O, C, 3, 3, 3), generator=g) * 0.05).to(dev)
    shift = torch.randn((O,), generator=g).to(dev)
    res = torch.randn((B, O, X, Y, Z), generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    y = _lib.wino_fused_conv3d_(x, _lib.wino_weights(w), shift, mode, res if mode >= 2 else None)
    ref = F.conv3d(x.double(), w.double(), padding=1) + shift.double().view(1, O, 1, 1, 1)
    if mode == 2:
        ref = ref + res.double()
    if mode >= 1:
        ref = ref.clamp_min(0)
    if mode == 3:
        ref = ref + res.double()
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last_3d)
    assert float((y.double() - ref).abs().max()) <= 5e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("shape", [(4, 128, 64, (5, 5, 2)), (2, 64, 32, (4, 6, 3)), (1, 8, 4, (2, 2, 2))])
def test_upsample2x_gemm_scatter(dev, shape):
    """ConvTranspose3d(2, stride 2) as one GEMM + sp3d_upsample2x_scatter == conv_transpose3d + shift + ReLU + skip"""
    import torch.nn.functional as F
    from selfpose3d_amd import _lib
    B, C, O, (X, Y, Z) = shape
    g = torch.Generator(device="cpu").manual_seed(13)
    x = torch.randn((B, C, X, Y, Z), generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn((C, O, 2, 2, 2), generator=g) * 0.1).to(dev)
    shift = torch.randn((O,), generator=g).to(dev)
    skip = torch.randn((B, O, 2 * X, 2 * Y, 2 * Z), generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    wg = w.permute(0, 2, 3, 4, 1).reshape(C, 8 * O).contiguous()
    y = _lib.upsample2x_(x, wg, shift, skip)
    ref = (F.conv_transpose3d(x.double(), w.double(), stride=2) + shift.double().view(1, O, 1, 1, 1)).clamp_min(0) + skip.double()
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last_3d)
    assert float((y.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_nhwc_heatmap_views_skip_the_retiling_pass(dev):
    """heat-maps handed over as (B,J,h,w) VIEWS of a channels-last (V,B,h,w,16) buffer (what PoseResNet.forward_views
    emits) are unprojected straight from that buffer: same bits as the planar hand-over, no pack kernel, gradients too"""
    from selfpose3d_amd import _lib
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.project_layer import ProjectLayer, clear_pack_cache, nhwc_heatmap_views
    for name in ("unproj_coarse_full_96x72", "unproj_fine_small", "unproj_coarse_aug"):
        case = gio.Case(name)
        cfg = load_config(None, NETWORK__IMAGE_SIZE=case.img, NETWORK__HEATMAP_SIZE=case.hm)
        layer = ProjectLayer(cfg)
        gc = case.grid_center if isinstance(case.grid_center, list) else case.grid_center.to(dev)
        planar = [h.to(dev) for h in case.hms]
        base, grids = layer(planar, case.meta, case.grid_size, gc, case.cube, flip_xcoords=case.flip)
        packed = _lib.pack_heatmaps(planar, jp=ProjectLayer.jp_for(case.J))
        views = nhwc_heatmap_views(packed, case.J)
        assert all(torch.equal(v, p) for v, p in zip(views, planar)) and not views[0].is_contiguous()
        clear_pack_cache()
        calls = []
        orig = _lib.pack_heatmaps
        _lib.pack_heatmaps = lambda *a, **k: calls.append(1) or orig(*a, **k)
        try:
            got, g2 = layer(views, case.meta, case.grid_size, gc, case.cube, flip_xcoords=case.flip)
            padded, _ = layer.get_voxel(views, case.meta, case.grid_size, gc, case.cube, flip_xcoords=case.flip,
                                        want_grids=False, pad_channels=True, channels_last=True)
        finally:
            _lib.pack_heatmaps = orig
        assert not calls, "the re-tiling pass must not run for NHWC views"
        assert torch.equal(got, base) and torch.equal(g2, grids)
        assert torch.equal(padded[:, :case.J], base) and torch.count_nonzero(padded[:, case.J:]) == 0
    # gradients: the same scatter result whichever way the maps were handed over
    case = gio.Case("unproj_grad_small")
    cfg = load_config(None, NETWORK__IMAGE_SIZE=case.img, NETWORK__HEATMAP_SIZE=case.hm)
    layer = ProjectLayer(cfg)
    wgt = torch.from_numpy(np.random.default_rng(int(case.g["grad_seed"])).standard_normal(
        (case.B, case.J, *case.cube)).astype(np.float32)).to(dev)
    packed = _lib.pack_heatmaps([h.to(dev) for h in case.hms], jp=ProjectLayer.jp_for(case.J)).requires_grad_(True)
    views = nhwc_heatmap_views(packed, case.J)
    cubes, _ = layer(views, case.meta, case.grid_size, case.grid_center, case.cube)
    (cubes * wgt).sum().backward()
    gp = packed.grad.permute(0, 1, 4, 2, 3)[:, :, :case.J].cpu().numpy()            # (V,B,J,h,w)
    ref = case.g["grad_hm"]
    assert np.abs(gp - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max()))
    assert float(packed.grad[..., case.J:].abs().max()) == 0.0 if packed.shape[-1] > case.J else True


def test_backbone_emits_unprojection_ready_heatmaps(dev):
    """PoseResNet.forward_views on the GPU: 16-channel channels-last head, per-view (B,15,h,w) views, 16th channel zero,
    values equal to the plain per-view forward up to conv rounding"""
    from selfpose3d_amd import pose_resnet
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.project_layer import _packed_source
    cfg = load_config(None, POSE_RESNET__NUM_LAYERS=18)
    net = pose_resnet.get_pose_net(cfg, is_train=False)
    torch.manual_seed(0)
    for m in net.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            torch.nn.init.kaiming_normal_(m.weight)
    net.eval().to(dev).to(memory_format=torch.channels_last)
    views = [torch.randn(2, 3, 64, 96, device=dev) for _ in range(3)]
    with torch.no_grad():
        outs = net.forward_views(views)
        ref = [net(v) for v in views]
    src = _packed_source(outs, 16, torch.float32)
    assert src is not None and src.shape == (3, 2, 16, 24, 16)
    assert torch.count_nonzero(src[..., 15]) == 0
    for o, r in zip(outs, ref):
        assert o.shape == r.shape == (2, 15, 16, 24)
        assert float((o - r).abs().max()) <= 1e-4 * float(r.abs().max())


@pytest.mark.parametrize("name", ["unproj_coarse_full_96x72", "unproj_fine_small", "unproj_coarse_aug", "unproj_coarse_j1_v1",
                                  "unproj_fine_full_240x128"])
def test_strided_result_into_a_larger_buffer(dev, name):
    """sp3d_unproject_fwd_strided: the planar result lands inside a larger buffer (the FFT conv's zero-padded input),
    bit-identical to the dense result, nothing outside the addressed elements is touched; aligned and unaligned rows"""
    from selfpose3d_amd import _lib
    case = gio.Case(name)
    base, _ = _hip_fwd(case, dev, "nhwc", want_grids=False)
    hms = [h.to(dev) for h in case.hms]
    cam = torch.from_numpy(case.cam).to(dev)
    centers = torch.from_numpy(case.centers).to(dev)
    valid = torch.from_numpy(case.valid).to(dev)
    w, h = case.hm
    jp = 4 if case.J <= 4 else (8 if case.J <= 8 else (12 if case.J <= 12 else 16))
    packed = _lib.pack_heatmaps(hms, jp=jp)
    views = [packed[c] for c in range(case.V)]
    X, Y, Z = case.cube
    for pad in ((6, 8, 8), (3, 5, 6), (0, 0, 0)):            # 16-byte aligned rows / unaligned rows / dense strides
        buf = torch.full((case.B, case.J + 1, X + pad[0], Y + pad[1], Z + pad[2]), -7.0, device=dev)
        view = buf[:, :case.J, :X, :Y, :Z]
        got, _ = _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, jp, cam, centers, valid, case.B, case.J, h, w, case.cube,
                                    case.grid_size, case.img, False, out=view)
        assert got.data_ptr() == view.data_ptr()
        assert torch.equal(view, base), pad
        mask = torch.ones_like(buf, dtype=torch.bool)
        mask[:, :case.J, :X, :Y, :Z] = False
        assert torch.all(buf[mask] == -7.0)


def test_deterministic_backward_is_bit_reproducible(dev):
    """SP3D_BWD_DETERMINISTIC: 64-bit fixed-point accumulation - the same bits on every run (the fp32-atomic scatter is
    only reproducible up to rounding order), and within 1e-6 of the float64 gradient of the reference"""
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.project_layer import ProjectLayer
    for name in ("unproj_grad_small", "unproj_grad_fine_aug"):
        case = gio.Case(name)
        cfg = load_config(None, NETWORK__IMAGE_SIZE=case.img, NETWORK__HEATMAP_SIZE=case.hm)
        layer = ProjectLayer(cfg)
        layer.deterministic_backward = True
        gc = case.grid_center if isinstance(case.grid_center, list) else case.grid_center.to(dev)
        wgt = torch.from_numpy(np.random.default_rng(int(case.g["grad_seed"])).standard_normal(
            (case.B, case.J, *case.cube)).astype(np.float32)).to(dev)
        runs = []
        for _ in range(3):
            hms = [h.to(dev).requires_grad_(True) for h in case.hms]
            cubes, _ = layer(hms, case.meta, case.grid_size, gc, case.cube, flip_xcoords=case.flip)
            (cubes * wgt).sum().backward()
            runs.append(torch.stack([h.grad for h in hms]))
        assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
        ref = case.g["grad_hm"]
        assert np.abs(runs[0].cpu().numpy() - ref).max() <= 2e-6 * max(1.0, float(np.abs(ref).max()))


def test_maxpool2x_channels_last_equals_torch(dev):
    from selfpose3d_amd import _lib
    import torch.nn.functional as F
    g = torch.Generator(device="cpu").manual_seed(3)
    for shape in ((2, 32, 16, 12, 8), (1, 64, 40, 40, 10), (3, 4, 2, 2, 2)):
        x0 = torch.randn(shape, generator=g)
        x0.view(-1)[::97] = float("nan")                                  # NaN propagates like torch.max_pool3d
        x = x0.to(dev).contiguous(memory_format=torch.channels_last_3d)
        got = _lib.maxpool2x(x)
        ref = F.max_pool3d(x, 2, 2)
        assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last_3d)
        assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(ref, nan=-7.0))
        assert torch.equal(torch.isnan(got), torch.isnan(ref))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 16, 24, 24, 12, 8, 20, 8), (1, 32, 9, 11, 64, 7, 10, 64), (3, 4, 6, 5, 4, 6, 5, 4)])
def test_crop_shift_act_channels_last_equals_torch(shape):
    """sp3d_crop_shift_act_cl == relu(src[..., :X,:Y,:Z] + shift) in channels_last_3d, bit for bit"""
    from selfpose3d_amd import _lib
    B, C, SX, SY, SZ, X, Y, Z = shape
    g = torch.Generator().manual_seed(5)
    src = torch.randn(B, C, SX, SY, SZ, generator=g).cuda()
    src[0, 1, 0, 0, 0] = float("nan")
    shift = torch.randn(C, generator=g).cuda()
    for relu in (True, False):
        got = _lib.crop_shift_act_cl(src, X, Y, Z, shift, relu)
        want = src[:, :, :X, :Y, :Z] + shift.view(1, C, 1, 1, 1)
        if relu:
            want = torch.relu(want)
        assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last_3d)
        assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(want, nan=-7.0))
        assert bool(torch.isnan(got[0, 1, 0, 0, 0]))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 3, 24, 20, 12), (1, 2, 22, 18, 10), (5, 8, 8, 8)])
def test_rfft3d_plans_equal_torch_fft(shape):
    """sp3d_rfft3d / sp3d_irfft3d run the same rocFFT transforms as torch.fft.rfftn / irfftn(norm='forward'): equal
    spectra (fp32 round-off), the input is preserved, round trip = N * x"""
    from selfpose3d_amd import _lib
    g = torch.Generator().manual_seed(11)
    x = torch.randn(shape, generator=g).cuda()
    x0 = x.clone()
    dims = tuple(range(x.dim() - 3, x.dim()))
    ref = torch.fft.rfftn(x, dim=dims)
    got = _lib.rfft3d(x)
    assert torch.equal(x, x0)
    assert got.shape == ref.shape and got.dtype == torch.complex64
    scale = float(torch.view_as_real(ref).abs().max())
    assert float((torch.view_as_real(got) - torch.view_as_real(ref)).abs().max()) <= 2e-6 * scale
    back = _lib.irfft3d_(got.clone(), shape[-1])
    want = torch.fft.irfftn(ref, s=shape[-3:], dim=dims, norm="forward")
    assert float((back - want).abs().max()) <= 2e-6 * float(want.abs().max())
    n = shape[-1] * shape[-2] * shape[-3]
    assert float((back / n - x0).abs().max()) <= 1e-5
    with pytest.raises(_lib.Sp3dError):
        _lib.rfft3d(x.transpose(-1, -2))


@pytest.mark.gpu
@pytest.mark.parametrize("J", [1, 15, 9])
def test_upsample_scatter_with_output_head(J):
    """sp3d_upsample2x_scatter_head == 1x1x1 output conv applied to sp3d_upsample2x_scatter's result"""
    from selfpose3d_amd import _lib
    g = torch.Generator().manual_seed(3)
    B, Cin, X, Y, Z, O = 2, 64, 6, 5, 3, 32
    x = torch.randn(B, Cin, X, Y, Z, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    wg = (0.1 * torch.randn(Cin, 8 * O, generator=g)).cuda()
    shift = torch.randn(O, generator=g).cuda()
    skip = torch.randn(B, O, 2 * X, 2 * Y, 2 * Z, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    wo = torch.randn(J, O, 1, 1, 1, generator=g).cuda()
    bo = torch.randn(J, generator=g).cuda()
    full = _lib.upsample2x_(x, wg, shift, skip)
    want = torch.nn.functional.conv3d(full.double(), wo.double(), bo.double())
    got = _lib.upsample2x_head_(x, wg, shift, skip, wo, bo)
    assert got.shape == want.shape
    assert float((got.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [(2, 12, 20, 18, 26, 15), (1, 16, 7, 22, 8, 16)])
def test_direct_z_dft_passes_equal_torch_fft(dims):
    """sp3d_zdft_fwd_cl + sp3d_cfft2d == rfftn of the zero-padded planar volume (kz slowest), and
    sp3d_cfft2d(inverse) + sp3d_zdft_inv_cl == relu(shift + irfftn(norm='forward'))[:X,:Y,:Z] in channels-last"""
    from selfpose3d_amd import _lib
    B, X, Y, SX, SY, cout = dims
    Z, SZ, C = 20, 28, 16
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, C, X, Y, Z, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    pad = torch.zeros(B, cout, SX, SY, SZ, device="cuda")
    pad[:, :, :X, :Y, :Z] = x[:, :cout]
    ref = torch.fft.rfftn(pad, dim=(2, 3, 4)).permute(0, 1, 4, 2, 3).contiguous()
    spec = _lib.zdft_fwd_cl(x, cout, (SX, SY, SZ))
    zref = torch.fft.rfft(pad, dim=4).permute(0, 1, 4, 2, 3)
    scale = float(torch.view_as_real(zref).abs().max())
    assert float((torch.view_as_real(spec) - torch.view_as_real(zref)).abs().max()) <= 3e-6 * scale
    _lib.cfft2d_(spec, False)
    scale = float(torch.view_as_real(ref).abs().max())
    assert float((torch.view_as_real(spec) - torch.view_as_real(ref)).abs().max()) <= 3e-6 * scale
    # way back, 16 output channels: spectrum of a real (B,16,SX,SY,SZ) volume
    vol = torch.randn(B, 16, SX, SY, SZ, generator=g).cuda()
    sp = torch.fft.rfftn(vol, dim=(2, 3, 4))
    shift = torch.randn(16, generator=g).cuda()
    for relu in (True, False):
        want = torch.fft.irfftn(sp, s=(SX, SY, SZ), dim=(2, 3, 4), norm="forward")[:, :, :X, :Y, :Z] + shift.view(1, 16, 1, 1, 1)
        if relu:
            want = torch.relu(want)
        ks = sp.permute(0, 1, 4, 2, 3).contiguous()
        got = _lib.zdft_inv_cl(_lib.cfft2d_(ks, True), X, Y, Z, SZ, shift, relu)
        assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last_3d)
        assert float((got - want).abs().max()) <= 3e-6 * float(want.abs().max())
    with pytest.raises(_lib.Sp3dError):
        _lib.zdft_fwd_cl(x[..., :16].contiguous(memory_format=torch.channels_last_3d), cout, (SX, SY, 20))


@pytest.mark.gpu
def test_single_kernel_88x88_plane_transform_equals_torch_fft():
    """sp3d_cfft2d_ex on 88x88 planes (one kernel, plane in LDS) == torch.fft.fft2 / unnormalised ifft2, == the hipFFT
    plan; rows_in / rows_out only skip work on rows that are zero / unread"""
    from selfpose3d_amd import _lib
    g = torch.Generator().manual_seed(8)
    z = torch.view_as_complex(torch.randn(7, 3, 88, 88, 2, generator=g)).cuda()
    ref = torch.fft.fft2(z)
    scale = float(torch.view_as_real(ref).abs().max())
    got = _lib.cfft2d_(z.clone(), False)
    assert float(torch.view_as_real(got - ref).abs().max()) <= 3e-6 * scale
    lib_plan = _lib.cfft2d_(z.clone(), False, library=True)
    assert float(torch.view_as_real(got - lib_plan).abs().max()) <= 3e-6 * scale
    back = _lib.cfft2d_(ref.clone(), True)
    assert float(torch.view_as_real(back / (88 * 88) - z).abs().max()) <= 1e-5
    want = torch.fft.ifft2(ref, norm="forward")
    assert float(torch.view_as_real(back - want).abs().max()) <= 3e-6 * float(torch.view_as_real(want).abs().max())
    # zero-padded input: rows >= 80 are zero and declared so
    zp = z.clone()
    zp[..., 80:, :] = 0
    got = _lib.cfft2d_(zp.clone(), False, rows_in=80)
    ref = torch.fft.fft2(zp)
    assert float(torch.view_as_real(got - ref).abs().max()) <= 3e-6 * float(torch.view_as_real(ref).abs().max())
    # only the first 80 rows of the inverse are needed
    part = _lib.cfft2d_(ref.clone(), True, rows_out=80)
    want = torch.fft.ifft2(ref, norm="forward")
    assert float(torch.view_as_real(part[..., :80, :] - want[..., :80, :]).abs().max()) <= \
        3e-6 * float(torch.view_as_real(want).abs().max())
    # other plane sizes go through the library plan
    y = torch.view_as_complex(torch.randn(5, 30, 18, 2, generator=g)).cuda()
    got = _lib.cfft2d_(y.clone(), False, rows_in=20)
    assert float(torch.view_as_real(got - torch.fft.fft2(y)).abs().max()) <= 3e-6 * float(torch.view_as_real(torch.fft.fft2(y)).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 32, (16, 12, 8), 1), (1, 16, (9, 7, 5), 2), (1, 32, (8, 8, 4), 3), (2, 16, (24, 8, 4), 0)])
def test_winograd_fused_split_kernel_has_fp32_accuracy(shape):
    """sp3d_wino_fused_split (three exact bf16 pieces per operand on v_mfma_f32_32x32x16_bf16, fp32 accumulation) against
    a float64 conv: same tolerance as the fp32-MFMA kernel AND an error no larger than 1.5x that kernel's (+1e-7)"""
    import torch.nn.functional as F
    from selfpose3d_amd import _lib
    B, C, (X, Y, Z), mode = shape
    O = 32
    g = torch.Generator(device="cpu").manual_seed(19)
    x = (torch.randn((B, C, X, Y, Z), generator=g) * 3.0).cuda().contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn((O, C, 3, 3, 3), generator=g) * 0.05).cuda()
    shift = torch.randn((O,), generator=g).cuda()
    res = torch.randn((B, O, X, Y, Z), generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    U = _lib.wino_weights(w)
    U3 = _lib.wino_weights_split(U)
    # the three pieces reproduce the fp32 weights exactly
    back = U3.float().sum(4).permute(0, 1, 2, 4, 3).reshape(64, C, O)
    assert torch.equal(back, U)
    ref = F.conv3d(x.double(), w.double(), padding=1) + shift.double().view(1, O, 1, 1, 1)
    if mode == 2:
        ref = ref + res.double()
    if mode >= 1:
        ref = ref.clamp_min(0)
    if mode == 3:
        ref = ref + res.double()
    r = res if mode >= 2 else None
    y32 = _lib.wino_fused_conv3d_(x, U, shift, mode, r)
    y3 = _lib.wino_fused_conv3d_(x, U, shift, mode, r, U3)
    e32 = float((y32.double() - ref).abs().max())
    e3 = float((y3.double() - ref).abs().max())
    assert y3.shape == ref.shape and y3.is_contiguous(memory_format=torch.channels_last_3d)
    assert e3 <= 5e-5 * max(1.0, float(ref.abs().max()))
    assert e3 <= 1.5 * e32 + 1e-7, (e3, e32)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 64, (16, 16, 4), 1), (1, 32, (9, 7, 5), 2), (1, 64, (8, 8, 2), 3), (2, 32, (24, 8, 6), 0),
                                   (1, 64, (40, 40, 10), 2),
                                   # round 5: interior blocks take a branch-free epilogue, edge blocks load their residual from
                                   # clamped addresses - ragged grids with a residual in both residual modes, and a grid that has
                                   # interior AND edge blocks
                                   (1, 64, (10, 9, 3), 3), (2, 64, (12, 10, 5), 2), (1, 32, (20, 17, 4), 3)])
def test_winograd_fused_split64_kernel(shape):
    """sp3d_wino_fused_split64 (half-resolution layers, 16x16x32 bf16 MFMA, three exact pieces) == conv3d + epilogue vs a
    float64 referee; block-edge and odd sizes included"""
    import torch.nn.functional as F
    from selfpose3d_amd import _lib
    B, C, (X, Y, Z), mode = shape
    O = 64
    g = torch.Generator(device="cpu").manual_seed(23)
    x = (torch.randn((B, C, X, Y, Z), generator=g) * 2.0).cuda().contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn((O, C, 3, 3, 3), generator=g) * 0.05).cuda()
    shift = torch.randn((O,), generator=g).cuda()
    res = torch.randn((B, O, X, Y, Z), generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    U = _lib.wino_weights(w)
    U3 = _lib.wino_weights_split(U, 16)
    ref = F.conv3d(x.double(), w.double(), padding=1) + shift.double().view(1, O, 1, 1, 1)
    if mode == 2:
        ref = ref + res.double()
    if mode >= 1:
        ref = ref.clamp_min(0)
    if mode == 3:
        ref = ref + res.double()
    y = _lib.wino_fused_conv3d_(x, U, shift, mode, res if mode >= 2 else None, U3)
    three = _lib.wino_conv3d_(x, U, shift, mode, res if mode >= 2 else None)          # the three-launch fp32 form
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last_3d)
    e, e3 = float((y.double() - ref).abs().max()), float((three.double() - ref).abs().max())
    assert e <= 5e-5 * max(1.0, float(ref.abs().max()))
    assert e <= 1.5 * e3 + 1e-7, (e, e3)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 32, 32, (16, 12, 8), 1), (1, 16, 32, (9, 7, 5), 2), (1, 32, 32, (8, 8, 2), 3),
                                   (2, 16, 32, (24, 8, 6), 0), (3, 32, 32, (40, 40, 10), 2), (1, 32, 32, (80, 80, 20), 1)])
def test_direct_conv3_split_kernel(shape):
    """sp3d_conv3_split (implicit GEMM, three exact bf16 pieces per operand, no Winograd) == conv3d + epilogue against a
    float64 referee, at least as close as MIOpen's fp32 convolution; block-edge and odd sizes included"""
    import torch.nn.functional as F
    from selfpose3d_amd import _lib
    B, C, O, (X, Y, Z), mode = shape
    g = torch.Generator(device="cpu").manual_seed(29)
    x = (torch.randn((B, C, X, Y, Z), generator=g) * 2.0).cuda().contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn((O, C, 3, 3, 3), generator=g) * 0.05).cuda()
    shift = torch.randn((O,), generator=g).cuda()
    res = torch.randn((B, O, X, Y, Z), generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    W3 = _lib.conv_weights_split(w)
    back = ((W3[..., 0, :].float() + W3[..., 4, :].float()) + W3[..., 1, :].float()).permute(0, 1, 2, 4, 3).reshape(27, C, O)
    assert torch.equal(back, w.permute(4, 3, 2, 1, 0).reshape(27, C, O))       # hi + lo + mid reproduce the weights exactly

    def epi(c, dt):
        c = c + shift.to(dt).view(1, O, 1, 1, 1)
        if mode == 2:
            c = c + res.to(dt)
        if mode >= 1:
            c = c.clamp_min(0)
        if mode == 3:
            c = c + res.to(dt)
        return c
    ref = epi(F.conv3d(x.double(), w.double(), padding=1), torch.float64)
    lib32 = epi(F.conv3d(x, w, padding=1), torch.float32)
    y = _lib.conv3_split_(x, W3, shift, mode, res if mode >= 2 else None)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last_3d)
    e, e32 = float((y.double() - ref).abs().max()), float((lib32.double() - ref).abs().max())
    assert e <= 1e-5 * max(1.0, float(ref.abs().max()))
    assert e <= 1.5 * e32 + 1e-6, (e, e32)


def _nonfinite_case(C, dims, seed):
    """activations with +inf, -inf, NaN and 1e30 planted at isolated voxels (at least 5 apart: no window holds two)"""
    X, Y, Z = dims
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn((1, C, X, Y, Z), generator=g) * 2.0
    plants = [((2, 2, 2), 3, float("inf")), ((8, 3, 2), 0, float("-inf")), ((3, 9, 5), 5, float("nan")),
              ((9, 9, 2), 7, 1e30), ((13, 3, 6), 1, -1e30)]
    for (px, py, pz), c, v in plants:
        if px < X and py < Y and pz < Z:
            x[0, c % C, px, py, pz] = v
    return x, [p for p in plants if p[0][0] < X and p[0][1] < Y and p[0][2] < Z]


@pytest.mark.gpu
def test_direct_conv3_split_kernel_nonfinite_inputs():
    """round-2 review: the three-piece split must not turn +-inf into NaN (hi = inf, x - hi = NaN).  +inf / -inf / NaN / 1e30
    activations through sp3d_conv3_split make EXACTLY the outputs non-finite that an fp32 convolution does (float64 referee:
    no spreading, nothing lost); where the convolution gives +-inf the kernel gives that infinity or NaN (documented
    below); finite values - also the 1e30-sized ones - stay accurate"""
    import torch.nn.functional as F
    from selfpose3d_amd import _lib
    C, O, dims = 32, 32, (16, 12, 8)
    x, _ = _nonfinite_case(C, dims, 41)
    g = torch.Generator(device="cpu").manual_seed(43)
    w = torch.randn((O, C, 3, 3, 3), generator=g) * 0.05
    shift = torch.zeros(O)
    ref = F.conv3d(x.double(), w.double(), padding=1)
    xg = x.cuda().contiguous(memory_format=torch.channels_last_3d)
    y = _lib.conv3_split_(xg, _lib.conv_weights_split(w.cuda()), shift.cuda(), 0, None).cpu()
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(y), fin)               # exactly the outputs an fp32 convolution makes non-finite
    assert int((~fin).sum()) > 100
    # ... of the same kind or NaN: an infinity reaches the matrix pipe in the operand's hi piece and meets the weight's
    # three pieces, whose signs differ, so inf * w is formed as (inf * w_hi) + (inf * w_mid) + ... = NaN about half of the
    # time; giving non-finite operands a path that meets w_hi only costs 3 more VALU per loaded value in the loader waves
    # that already are this kernel's pole, and was not taken.  Never the other way round:
    same_or_nan = (torch.isposinf(y) == torch.isposinf(ref)) & (torch.isneginf(y) == torch.isneginf(ref)) | torch.isnan(y)
    assert bool(same_or_nan[~fin].all())
    assert bool(torch.isnan(y)[torch.isnan(ref)].all())      # a NaN of the reference is never lost
    big = ref.abs() > 1e20                                   # outputs that carry the 1e30 activations
    assert float(((y.double() - ref)[fin & ~big]).abs().max()) <= 1e-5 * 10.0
    assert float(((y.double() - ref)[fin & big] / ref[fin & big]).abs().max()) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["full", "half"])
def test_winograd_split_kernels_confine_nonfinite_inputs(kind):
    """Winograd F(2,3) transforms a 4x4x4 input patch per 2x2x2 output tile, so a non-finite input reaches every output of
    the tiles whose patch holds it (any Winograd form does this, in fp32 too): outputs whose window holds the value are
    non-finite as in the direct convolution, outputs of tiles whose patch is clean are finite and accurate"""
    import torch.nn.functional as F
    from selfpose3d_amd import _lib
    C, O, dims = (32, 32, (16, 12, 8)) if kind == "full" else (64, 64, (16, 16, 6))
    x, plants = _nonfinite_case(C, dims, 47)
    g = torch.Generator(device="cpu").manual_seed(49)
    w = torch.randn((O, C, 3, 3, 3), generator=g) * 0.05
    ref = F.conv3d(torch.nan_to_num(x, nan=0.0, posinf=0.0, neginf=0.0).clamp(-1e3, 1e3).double(), w.double(), padding=1)
    refnf = F.conv3d(x.double(), w.double(), padding=1)
    xg = x.cuda().contiguous(memory_format=torch.channels_last_3d)
    U = _lib.wino_weights(w.cuda())
    U3 = _lib.wino_weights_split(U) if kind == "full" else _lib.wino_weights_split(U, 16)
    y = _lib.wino_fused_conv3d_(xg, U, torch.zeros(O).cuda(), 0, None, U3).cpu()
    X, Y, Z = dims
    dirty = torch.zeros((X, Y, Z), dtype=torch.bool)        # outputs of tiles whose 4x4x4 patch holds a planted value
    for (px, py, pz), _, _ in plants:
        for tx in range((X + 1) // 2):
            for ty in range((Y + 1) // 2):
                for tz in range((Z + 1) // 2):
                    if 2 * tx - 1 <= px <= 2 * tx + 2 and 2 * ty - 1 <= py <= 2 * ty + 2 and 2 * tz - 1 <= pz <= 2 * tz + 2:
                        dirty[2 * tx:2 * tx + 2, 2 * ty:2 * ty + 2, 2 * tz:2 * tz + 2] = True
    clean = ~dirty
    assert bool(torch.isfinite(y[0][:, clean]).all())
    assert float((y[0][:, clean].double() - ref[0][:, clean]).abs().max()) <= 5e-5 * 10.0
    assert bool((~torch.isfinite(y))[~torch.isfinite(refnf)].all())          # superset of the direct convolution's pattern


@pytest.mark.gpu
def test_backbone_training_pass_batches_views_like_the_per_view_loop(monkeypatch):
    """PoseResNet.forward_views in TRAIN mode: one (B*V)-image pass with per-view BatchNorm statistics (ViewBatchNorm2d) ==
    the reference's loop over cameras (lib/models/multi_person_posenet.py:44-47).  In FLOAT64 on the GPU heat-maps, running
    statistics and every parameter's gradient agree to rounding (fp32 gradients through train-mode BatchNorm are
    ill-conditioned: the two passes use different convolution kernels and differ by percents there, as the reference's own
    fp32 run does from its float64 rerun); in fp32 the heat-maps agree to 2e-4; the same for the
    channels_last form of the pass (grouped BatchNorm kernels, round 5)"""
    import copy
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd import pose_resnet as pr
    monkeypatch.setattr(torch.backends.cudnn, "benchmark", False)
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)
    dev = torch.device("cuda:0")
    cfg = load_config(None)
    torch.manual_seed(3)
    a = pr.PoseResNet(cfg, 18).to(dev).train()
    for m in a.modules():                                   # the reference's N(0, 1e-3) init gives ~0 everywhere: use a live net
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            torch.nn.init.kaiming_normal_(m.weight)
    V, B = 3, 2
    views = [torch.randn(B, 3, 128, 192, device=dev) for _ in range(V)]
    for dt, out_tol, grad_tol in ((torch.float64, 1e-9, 1e-7), (torch.float32, 2e-4, None)):
        na = copy.deepcopy(a).to(dt)
        nb = copy.deepcopy(na)
        nb.batch_views_in_training = False
        vv = [v.to(dt) for v in views]
        ya, yb = na.forward_views(vv), nb.forward_views(vv)
        scale = max(float(y.detach().abs().max()) for y in yb)
        for u, w in zip(ya, yb):
            assert u.shape == w.shape and float((u - w).detach().abs().max()) <= out_tol * scale, dt
        for (n, u), w in zip(na.named_buffers(), nb.buffers()):
            assert torch.allclose(u.double(), w.double(), rtol=1e-4 if dt == torch.float32 else 1e-10,
                                  atol=(1e-5 if dt == torch.float32 else 1e-11) * max(1.0, float(w.double().abs().max()))), (n, dt)
        if grad_tol is None:
            continue
        sum((y * y).mean() for y in ya).backward()
        sum((y * y).mean() for y in yb).backward()
        for (n, p), q in zip(na.named_parameters(), nb.parameters()):
            g = float(q.grad.abs().max())
            assert float((p.grad - q.grad).abs().max()) <= grad_tol * max(g, 1e-30), (n, dt)
    # round 5: channels_last weights = ONE pass through the grouped channels-last BatchNorm kernels (sp3d_gbn_*, ReLU fused),
    # group of image n = n % V: float64 to rounding against the per-view loop (outputs, buffers, every gradient), fp32 to 2e-4
    for dt, out_tol, grad_tol in ((torch.float64, 1e-9, 1e-7), (torch.float32, 2e-4, None)):
        nb = copy.deepcopy(a).to(dt)
        nb.batch_views_in_training = False
        nc = copy.deepcopy(a).to(dt).to(memory_format=torch.channels_last)
        assert nc.batch_views_in_training and not nc.conv1.weight.is_contiguous()
        vv = [v.to(dt) for v in views]
        calls = []
        orig = nc.forward
        nc.forward = lambda x, *a_, **k: (calls.append(tuple(x.shape)), orig(x, *a_, **k))[1]
        yc, yb = nc.forward_views(vv), nb.forward_views(vv)
        assert calls == [(B * V, 3, 128, 192)]                          # one pass over all views
        assert all(m.groups is None for m in nc.modules() if isinstance(m, pr.ViewBatchNorm2d))      # spec detached again
        scale = max(float(y.detach().abs().max()) for y in yb)
        for u, w in zip(yc, yb):
            assert u.shape == w.shape and float((u - w).detach().abs().max()) <= out_tol * scale, dt
        for (n, u), w in zip(nc.named_buffers(), nb.buffers()):
            assert torch.allclose(u.double(), w.double(), rtol=1e-4 if dt == torch.float32 else 1e-10,
                                  atol=(1e-5 if dt == torch.float32 else 1e-11) * max(1.0, float(w.double().abs().max()))), (n, dt)
        if grad_tol is None:
            continue
        sum((y * y).mean() for y in yc).backward()
        sum((y * y).mean() for y in yb).backward()
        top = max(float(q.grad.abs().max()) for q in nb.parameters())
        for (n, p), q in zip(nc.named_parameters(), nb.parameters()):
            g = max(float(q.grad.abs().max()), 1e-6 * top)
            assert float((p.grad - q.grad).abs().max()) <= grad_tol * g, (n, dt)
    # the model-level switch: channels_last for a GPU backbone in train and eval mode alike
    assert not pr.set_backbone_memory_format(a, True).conv1.weight.is_contiguous()
    assert not pr.set_backbone_memory_format(a.eval(), True).conv1.weight.is_contiguous()
    assert pr.set_backbone_memory_format(a, False).conv1.weight.is_contiguous()
