"""Round 6: the root grid's unprojection fused with the z pass of the opening 7^3 conv (sp3d_unproject_fwd_zdft +
sp3d_cfft2d_88_tiled; round-5 review item 3a).  The contract is BIT-identity with the two-kernel path it replaces
(sp3d_unproject_fwd channels-last -> sp3d_zdft_fwd_cl -> sp3d_cfft2d_ex), which is itself pinned to the reference's
ProjectLayer by the golden cases of tests/test_gpu_parity.py / test_gpu_bwd_full_size.py - on those same inputs."""
import numpy as np
import pytest
import torch

from tests import golden_io as gio

pytestmark = pytest.mark.gpu
SZ, S = 28, (88, 88, 28)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def untile(spec, X, Y):
    """(B,J,K,X/4,Y/4,16) tiled -> (B,J,K,X,Y)"""
    B, J, K = spec.shape[:3]
    return spec.view(B, J, K, X // 4, Y // 4, 4, 4).permute(0, 1, 2, 3, 5, 4, 6).reshape(B, J, K, X, Y)


def both_paths(dev, case, valid=None):
    from selfpose3d_amd import _lib
    w, h = case.hm
    X, Y, Z = case.cube
    packed = _lib.pack_heatmaps([x.to(dev) for x in case.hms], jp=16)
    views = [packed[c] for c in range(case.V)]
    cam, cen = torch.from_numpy(case.cam).to(dev), torch.from_numpy(case.centers).to(dev)
    val = torch.from_numpy(case.valid if valid is None else valid).to(dev)
    cubes, _ = _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, cen, val, case.B, 16, h, w, case.cube, case.grid_size,
                                  case.img, False, channels_last=True)
    two = _lib.zdft_fwd_cl(cubes, case.J, S)                                       # (B,J,15,88,88)
    one = _lib.unproject_fwd_zdft(views, 16, cam, cen, val, case.B, case.J, h, w, case.cube, case.grid_size, case.img, SZ)
    return cubes, two, one


@pytest.mark.parametrize("name", ["unproj_coarse_b4", "unproj_grad_root_full", "unproj_coarse_full_240x128", "unproj_people_coarse"])
def test_fused_spectrum_is_the_two_kernel_spectrum_bit_for_bit(dev, name):
    """configs[1] exactly, the augmented + flipped B=4 case (values outside [0,1]: the clamp is active), B=1 and the people
    scene: every complex value equal as BITS, the padding the two-kernel form stores is zero, nothing else differs"""
    case = gio.Case(name)
    X, Y, Z = case.cube
    assert (X, Y, Z) == (80, 80, 20)
    cubes, two, one = both_paths(dev, case)
    assert one.shape == (case.B, case.J, SZ // 2 + 1, X // 4, Y // 4, 16) and one.dtype == torch.complex64
    got = torch.view_as_real(untile(one, X, Y))
    ref = torch.view_as_real(two[..., :X, :Y])
    assert torch.equal(got.view(torch.int32), ref.contiguous().view(torch.int32))
    assert torch.count_nonzero(torch.view_as_real(two[..., X:, :])) == 0 and torch.count_nonzero(torch.view_as_real(two[..., :, Y:])) == 0
    assert float(ref.abs().max()) > 1.0
    # the cubes behind it are the reference's (the same golden the unfused path is pinned to)
    g = case.g
    exp = g["cubes_sub"] if "cubes_sub" in g else g["cubes"].reshape(case.B, case.J, -1)
    idx = g["sub_idx"] if "sub_idx" in g else slice(None)
    o = cubes[:, :case.J].cpu().numpy().reshape(case.B, case.J, -1)[:, :, idx]
    assert np.abs(o - exp).max() <= 5e-7


def test_skipped_sample_gives_a_zero_spectrum(dev):
    case = gio.Case("unproj_coarse_b4")
    valid = np.array([1, 0, 1, 0], np.uint8)
    _, two, one = both_paths(dev, case, valid)
    got = untile(one, 80, 80)
    assert torch.equal(torch.view_as_real(got), torch.view_as_real(two[..., :80, :80]).contiguous())
    assert torch.count_nonzero(torch.view_as_real(one[1])) == 0 and torch.count_nonzero(torch.view_as_real(one[3])) == 0
    assert torch.count_nonzero(torch.view_as_real(one[0])) > 0


def test_tiled_plane_transform_equals_the_padded_one(dev):
    from selfpose3d_amd import _lib
    case = gio.Case("unproj_grad_root_full")
    _, two, one = both_paths(dev, case)
    ref = _lib.cfft2d_(two.clone(), False, rows_in=80)
    got = _lib.cfft2d_88_tiled(one, 80, 80)
    assert got.shape == ref.shape == (case.B, case.J, 15, 88, 88)
    assert torch.equal(torch.view_as_real(got).view(torch.int32), torch.view_as_real(ref).view(torch.int32))


def test_unsupported_shapes_are_refused(dev):
    from selfpose3d_amd import _lib
    case = gio.Case("unproj_coarse_small")                          # 8 x 8 x 4 grid
    packed = _lib.pack_heatmaps([x.to(dev) for x in case.hms], jp=16)
    cam, cen, val = (torch.from_numpy(a).to(dev) for a in (case.cam, case.centers, case.valid))
    with pytest.raises(_lib.Sp3dError):
        _lib.unproject_fwd_zdft([packed[c] for c in range(case.V)], 16, cam, cen, val, case.B, case.J, case.hm[1], case.hm[0],
                                case.cube, case.grid_size, case.img, SZ)


@pytest.mark.parametrize("graph", [False, True])
def test_root_net_fused_equals_unfused(dev, graph):
    """CuboidProposalNet on the rootnet_full golden's inputs: V2VNet.fuse_zdft on / off -> identical root cubes and proposals
    (eager and as a replayed HIP graph), and both equal the reference golden"""
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.cuboid_proposal_net import CuboidProposalNet
    from selfpose3d_amd.graphs import GraphedRootNet
    from tests.test_gpu_reference_pins_r2 import _check_root, _rootnet_full_inputs
    g = gio.load("rootnet_full")
    img, hm, V, J, hms, meta = _rootnet_full_inputs(g)
    net = CuboidProposalNet(load_config(None)).eval()
    syn.fill_parameters_deterministic(net, seed=int(g["param_seed"]), scale=float(g["param_scale"]))
    net.to(dev).use_channels_last(True)
    hms = [x.to(dev) for x in hms]
    outs, called = {}, {}
    inner = net.project_layer.get_voxel_zspectrum
    for fuse in (True, False):
        net.v2v_net.fuse_zdft = fuse
        called[fuse] = 0

        def spy(*a, _f=fuse, **k):
            called[_f] += 1
            return inner(*a, **k)
        net.project_layer.get_voxel_zspectrum = spy
        with torch.no_grad():
            if graph:
                for _ in range(2):
                    net(hms, meta)
                gr = GraphedRootNet(net, hms, meta)
                gr()
                rc, gc = gr()
                rc, gc = rc.clone(), gc.clone()
            else:
                rc, gc = net(hms, meta)
        torch.cuda.synchronize()
        outs[fuse] = (rc, gc)
    del net.project_layer.get_voxel_zspectrum
    assert called[True] > 0 and called[False] == 0                  # the switch really selects the path
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    _check_root(outs[True][0], outs[True][1], g)


def test_contraction_with_the_y_transform_rebuilt_per_bin(dev):
    """sp3d_freq_contract_ty (17.7 MB table of the taps transformed along z and x) against sp3d_freq_contract on the full
    223 MB weight spectrum (conj(rfftn(taps)) / N, what the plan used until round 6) and against a float64 evaluation"""
    from selfpose3d_amd import _lib
    from selfpose3d_amd.v2v_net import V2VNet, _FoldedV2V
    torch.manual_seed(5)
    net = V2VNet(15, 1).to(dev).eval()
    plan = _FoldedV2V(net)
    plan._build()
    plan.key = plan._key(net)
    w0, s0 = plan.t["front"]
    S = (88, 88, 28)
    Wz = plan._weights_z(w0, S)                                           # (16,15,15,88,88)
    T, tw = plan._weights_ty(w0, S)
    assert T.shape == (15 * 88, 16, 15, 14) and tw.shape == (88, 3, 2)
    for B in (4, 1, 6):
        X = torch.view_as_complex(torch.randn((B, 15, 15, 88, 88, 2), device=dev))
        ref = _lib.freq_contract(X, Wz)
        got = _lib.freq_contract_ty(X, T, tw)
        exact = torch.einsum("bckxy,ockxy->bokxy", X.to(torch.complex128), Wz.to(torch.complex128))
        scale = float(exact.detach().abs().max())
        e_ref, e_got = float((ref - exact).abs().max()) / scale, float((got - exact).abs().max()) / scale
        assert e_got <= max(2.0 * e_ref, 2e-6), (B, e_got, e_ref)
        assert float((got - ref).abs().max()) / scale <= 4e-6
