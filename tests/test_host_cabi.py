"""CPU-side tests: the C-ABI library loads and exports everything include/sp3d.h declares,
argument validation happens before any GPU work, and the host logic (camera table, grid
centres, config) behaves like the reference's.  No compute calls are made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from selfpose3d_amd import _lib, build as sbuild, synthetic as syn
from selfpose3d_amd.camera_pack import (CAM_A, CAM_AXY, CAM_C2, CAM_F2, CAM_FLIP, CAM_FLIP2, CAM_H0, CAM_K2, CAM_P2, CAM_RXY, CAM_RZ,
                                        CAM_STRIDE, CAM_TAME, CAM_TXY, CAM_TZ, CAM_W0, CAM_WH, finish, meta_cache_key, pack_cameras)
from selfpose3d_amd.config import default_config, load_config
from selfpose3d_amd.project_layer import ProjectLayer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    sbuild.build()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "sp3d.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(sp3d_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 8
    assert sorted(_lib.EXPORTS) == declared
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sp3d_abi_version() == 3               # 2: camera records of 64 floats (derived second half); 3: per-call `scatter` argument of the packed backward
    # header constants match the python binding
    for macro, val in (("SP3D_CAM_STRIDE", CAM_STRIDE), ("SP3D_CAM_A", CAM_A), ("SP3D_CAM_W0", CAM_W0),
                       ("SP3D_CAM_H0", CAM_H0), ("SP3D_CAM_FLIP", CAM_FLIP), ("SP3D_CAM_P2", CAM_P2), ("SP3D_CAM_RXY", CAM_RXY),
                       ("SP3D_CAM_TXY", CAM_TXY), ("SP3D_CAM_AXY", CAM_AXY), ("SP3D_CAM_TAME", CAM_TAME), ("SP3D_CAM_RZ", CAM_RZ),
                       ("SP3D_CAM_TZ", CAM_TZ), ("SP3D_CAM_K2", CAM_K2), ("SP3D_CAM_F2", CAM_F2), ("SP3D_CAM_C2", CAM_C2),
                       ("SP3D_CAM_WH", CAM_WH), ("SP3D_CAM_FLIP2", CAM_FLIP2), ("SP3D_MAX_VIEWS", _lib.MAX_VIEWS),
                       ("SP3D_MAX_TOPK", _lib.MAX_TOPK)):
        m = re.search(r"#define\s+%s\s+(\d+)" % macro, hdr)
        assert m and int(m.group(1)) == val, macro


def test_library_keeps_no_selector_state(lib):
    """include/sp3d.h: no state that selects behaviour (round 4 had a process-wide scatter selector; round 5: a per-call
    argument).  Every writable symbol of the library is either the toolchain's / HIP runtime's registration data or one of
    the three documented caches."""
    import subprocess
    out = subprocess.run(["nm", "-C", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    runtime = re.compile(r"_kernel(<.*>)?\(|__hip_|__do_|_DYNAMIC|_GLOBAL_OFFSET_TABLE_|__dso_handle|DW\.ref|__fini|__init|"
                         r"__TMC_END__|completed\.|__bss_start|_edata|_end$")
    documented = ("g_mu", "g_plans", "attr_set", "attr_dev", "cu_count")
    writable = [ln.split(None, 2)[2] for ln in out.splitlines() if re.match(r"^[0-9a-f]+ [bBdDsScC] ", ln)]
    assert len(writable) > 10                                   # (nm really listed the library)
    own = [w for w in writable if not runtime.search(w)]
    stray = [w for w in own if not w.endswith(documented)]
    assert not stray, stray
    assert not hasattr(lib, "sp3d_set_bwd_scatter")
    # the per-call selector is validated before any launch
    gs = (C.c_float * 3)(2000, 2000, 2000)
    d = C.c_void_p(0x1000)
    f = lib.sp3d_unproject_bwd_packed
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] * 7 + [C.c_int] * 10 + [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    assert f(d, None, d, d, d, d, d, 1, 1, 2, 15, 16, 8, 8, 4, 4, 4, gs, 96, 72, 1, None) == -1        # scatter = 1: EINVAL
    assert f(d, None, d, d, d, d, None, 1, 1, 2, 15, 16, 8, 8, 4, 4, 4, gs, 96, 72, _lib.SCATTER_MERGE, None) == -2


def test_argument_validation_before_any_launch(lib):
    gs = (C.c_float * 3)(8000, 8000, 2000)
    views = (C.c_void_p * 2)(0x1000, 0x1000)      # never dereferenced: validation fails first
    dummy = C.c_void_p(0x1000)
    f = lib.sp3d_unproject_fwd
    # dimension <= 0
    assert f(views, 0, 0, dummy, dummy, dummy, dummy, None, 0, 2, 15, 8, 8, 4, 4, 4, gs, 96, 72, None) == -1
    # too many views
    assert f(views, 0, 0, dummy, dummy, dummy, dummy, None, 1, 17, 15, 8, 8, 4, 4, 4, gs, 96, 72, None) == -1
    # voxel count overflow
    assert f(views, 0, 0, dummy, dummy, dummy, dummy, None, 1, 2, 15, 8, 8, 2048, 2048, 2048, gs, 96, 72, None) == -3
    # null pointers
    assert f(views, 0, 0, None, dummy, dummy, dummy, None, 1, 2, 15, 8, 8, 4, 4, 4, gs, 96, 72, None) == -2
    assert f(None, 0, 0, dummy, dummy, dummy, dummy, None, 1, 2, 15, 8, 8, 4, 4, 4, gs, 96, 72, None) == -2
    assert f(views, 0, 0, dummy, dummy, dummy, dummy, None, 1, 2, 15, 8, 8, 4, 4, 4, None, 96, 72, None) == -2
    # NHWC with a channel stride that is not a multiple of 4 / smaller than J / unknown layout
    assert f(views, 1, 15, dummy, dummy, dummy, dummy, None, 1, 2, 15, 8, 8, 4, 4, 4, gs, 96, 72, None) == -4
    assert f(views, 1, 8, dummy, dummy, dummy, dummy, None, 1, 2, 15, 8, 8, 4, 4, 4, gs, 96, 72, None) == -4
    assert f(views, 7, 16, dummy, dummy, dummy, dummy, None, 1, 2, 15, 8, 8, 4, 4, 4, gs, 96, 72, None) == -1
    assert lib.sp3d_pack_heatmaps(views, None, 1, 2, 15, 16, 8, 8, None) == -2
    assert lib.sp3d_pack_heatmaps(views, dummy, 1, 2, 15, 15, 8, 8, None) == -4
    assert lib.sp3d_nms_topk(dummy, 1, 4, 4, 4, 33, None, None, dummy, dummy, None, dummy, None) == -1
    assert lib.sp3d_nms_topk(None, 1, 4, 4, 4, 10, None, None, dummy, dummy, None, dummy, None) == -2
    # one k-entry candidate list per 4 x 8 x 32 voxel tile (round 3): 20 x 10 x 1 tiles per 80x80x20 sample
    assert lib.sp3d_nms_topk_workspace_bytes(4, 80, 80, 20, 10) == 4 * 200 * 10 * 8
    assert lib.sp3d_nms_topk_workspace_bytes(1, 64, 64, 64, 10) == 16 * 8 * 2 * 10 * 8
    assert lib.sp3d_soft_argmax(dummy, dummy, dummy, 0, 15, 64, C.c_float(100.0), None) == -1
    assert b"NULL" in lib.sp3d_error_string(-2)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.Sp3dError, match="no CPU fallback"):
        _lib.load()


def test_project_layer_refuses_cpu_tensors():
    cfg = load_config(None, NETWORK__IMAGE_SIZE=[96, 72], NETWORK__HEATMAP_SIZE=[24, 18])
    layer = ProjectLayer(cfg)
    meta = syn.make_meta(1, 2, (96, 72))
    hms = syn.random_heatmaps(1, 2, 3, 18, 24)
    with pytest.raises(_lib.Sp3dError, match="GPU"):
        layer(hms, meta, syn.SPACE_SIZE, [list(syn.SPACE_CENTER)], [4, 4, 4])


def test_camera_table_fields_and_cache_key():
    B, V, img = 3, 4, (192, 144)
    flip = torch.tensor([False, True, True])
    meta = syn.make_meta(B, V, img, rotations=[0.0, 30.0, -30.0], scale_mults=[1.0, 1.3, 0.8], ssv_style=True)
    tab = pack_cameras(meta, B, img, flip)
    assert tab.shape == (B, V, CAM_STRIDE) and tab.dtype == np.float32
    assert np.all(tab[:, :, CAM_W0] == 1920.0) and np.all(tab[:, :, CAM_H0] == 1080.0)
    assert np.array_equal(tab[:, 0, CAM_FLIP], np.array([0.0, 1.0, 1.0], np.float32))
    R = tab[:, :, 0:9].reshape(B, V, 3, 3)
    assert np.allclose(np.einsum("bvij,bvkj->bvik", R, R), np.eye(3), atol=1e-6)
    # rot = 0 closed form: a = W_in / (200*scale_w), t = [W_in/2, H_in/2] - a*center
    A = tab[0, 0, CAM_A:CAM_A + 6].reshape(2, 3)
    a = img[0] / (200.0 * float(meta[0]["scale"][0, 0]))
    assert np.allclose(A, [[a, 0, img[0] / 2 - a * 960.0], [0, a, img[1] / 2 - a * 540.0]], atol=1e-5)
    # derived fields: operand pairs as neighbours + the affine sanity flag; the C ABI's sp3d_camera_finish is the same function
    assert np.array_equal(tab[..., CAM_RXY:CAM_RXY + 6].reshape(B, V, 3, 2), R[:, :, :2, :].transpose(0, 1, 3, 2))
    assert np.array_equal(tab[..., CAM_TXY:CAM_TXY + 2], tab[..., 9:11]) and np.array_equal(tab[..., CAM_P2:CAM_P2 + 2], tab[..., 19:21])
    assert np.array_equal(tab[..., CAM_RZ:CAM_RZ + 3], tab[..., 6:9]) and np.array_equal(tab[..., CAM_TZ], tab[..., 11])
    assert np.array_equal(tab[..., CAM_K2:CAM_K2 + 3], tab[..., 16:19]) and np.array_equal(tab[..., CAM_F2:CAM_F2 + 4], tab[..., 12:16])
    assert np.array_equal(tab[..., CAM_WH:CAM_WH + 2], tab[..., 27:29]) and np.array_equal(tab[..., CAM_FLIP2], tab[..., CAM_FLIP])
    assert np.array_equal(tab[..., CAM_AXY:CAM_AXY + 6].reshape(B, V, 3, 2), tab[..., CAM_A:CAM_A + 6].reshape(B, V, 2, 3).transpose(0, 1, 3, 2))
    assert np.all(tab[..., CAM_TAME] == 1.0)
    edited = tab.copy()
    edited[0, 1, CAM_A + 4] = np.inf
    edited[1, 2, CAM_A] = np.float32(2e30)
    edited[2, 0, CAM_A + 2] = np.nan
    edited[2, 3, CAM_A + 1] = np.float32(-9e29)           # large but tame
    by_c = edited.copy()
    by_c[..., 30:] = -7.0                                 # garbage behind the camera fields: the C function overwrites all of it
    from selfpose3d_amd import _lib as L
    lib = L.load()
    assert lib.sp3d_camera_finish(by_c.ctypes.data_as(C.c_void_p), B * V) == 0
    finish(edited)
    assert np.array_equal(by_c.view(np.uint32), edited.view(np.uint32))
    tame = edited[..., CAM_TAME]
    assert tame[0, 1] == 0.0 and tame[1, 2] == 0.0 and tame[2, 0] == 0.0 and tame[2, 3] == 1.0 and tame.sum() == B * V - 3
    k1 = meta_cache_key(meta, flip, img)
    assert k1 == meta_cache_key(meta, flip, img)
    # the key is the CONTENT of what pack_cameras reads: a value-preserving edit keeps it, any value change - also of
    # a numpy entry edited in place, or of a new object at a recycled address - changes it
    meta[0]["scale"].mul_(1.0)
    assert k1 == meta_cache_key(meta, flip, img)
    meta[1]["camera"]["fx"] = meta[1]["camera"]["fx"].numpy().copy()      # numpy entry, same values
    assert k1 == meta_cache_key(meta, flip, img)
    meta[1]["camera"]["fx"][2] += 1.0                                     # in-place numpy edit
    k2 = meta_cache_key(meta, flip, img)
    assert k2 != k1
    meta[0]["scale"].mul_(1.01)
    assert meta_cache_key(meta, flip, img) not in (k1, k2)
    assert meta_cache_key(meta, ~flip, img) != meta_cache_key(meta, flip, img)


def test_centers_valid_follow_reference_rules():
    dev = torch.device("cpu")
    c, v = ProjectLayer.centers_valid([[0.0, -500.0, 800.0]], 3, dev)           # shared list centre
    assert c.shape == (3, 3) and torch.all(v == 1) and torch.all(c[:, 1] == -500.0)
    gc = torch.tensor([[1.0, 2.0, 3.0, 0.0, 0.9], [4.0, 5.0, 6.0, -1.0, 0.1]])
    c, v = ProjectLayer.centers_valid(gc, 2, dev)                                # per-sample (B,5)
    assert torch.equal(c, gc[:, :3]) and v.tolist() == [1, 0]
    c, v = ProjectLayer.centers_valid(gc[:1], 1, dev)
    assert v.tolist() == [1]


def test_config_reads_reference_yaml_shape(tmp_path):
    y = tmp_path / "c.yaml"
    y.write_text("NETWORK:\n  IMAGE_SIZE: [384, 288]\n  HEATMAP_SIZE: [96, 72]\n  ROOTNET_BUFFER_SIZE: 3\n"
                 "MULTI_PERSON:\n  THRESHOLD: 0.1\n")
    cfg = load_config(str(y))
    assert cfg.NETWORK.IMAGE_SIZE == [384, 288] and cfg.MULTI_PERSON.THRESHOLD == 0.1
    assert cfg.NETWORK.ROOTNET_BUFFER_SIZE == 3 and cfg.PICT_STRUCT.CUBE_SIZE == [64, 64, 64]
    # round 6: a key the reference's schema does not know is an error inside a section too (lib/core/config.py:253-257);
    # round 5 accepted it silently, which is how an unconsumed INIT_ROOTNET went unnoticed (tests/test_stage_handoff.py)
    y.write_text("NETWORK:\n  SOME_NEW_KEY: 3\n")
    with pytest.raises(ValueError, match="NETWORK.SOME_NEW_KEY"):
        load_config(str(y))
    bad = tmp_path / "bad.yaml"
    bad.write_text("NO_SUCH_SECTION:\n  A: 1\n")
    with pytest.raises(ValueError):
        load_config(str(bad))
    assert default_config().NETWORK.BETA == 100.0


def test_v2v_matches_reference_golden_on_cpu():
    from selfpose3d_amd.v2v_net import V2VNet
    from tests import golden_io as gio
    g = gio.load("v2v")
    m = V2VNet(int(g["in_shape"][1]), 1)
    assert sorted(m.state_dict().keys()) == list(g["keys"])
    syn.fill_parameters_deterministic(m, seed=int(g["param_seed"]), scale=float(g["param_scale"]))
    m.eval()
    x = torch.from_numpy(np.random.default_rng(int(g["in_seed"])).random(tuple(g["in_shape"]), dtype=np.float32))
    with torch.no_grad():
        y = m(x)
    assert float((y - torch.from_numpy(g["out"])).abs().max()) <= 1e-5


def _fma32(a, b, c):
    # products of two fp32 are exact in fp64; the sum is rounded once to fp64 then to fp32
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def test_const_division_is_exact():
    """div_const / fuse_rcp of csrc/sp3d_device.h: q=x*rc; r=fma(-q,c,x); q'=fma(r,rc,q) equals IEEE x/c
    for every constant the kernels divide by (image sizes, heat-map sizes - 1, view counts + 1e-6)."""
    rng = np.random.default_rng(0)
    consts = [960, 512, 239, 127, 384, 288, 95, 71, 192, 144, 47, 35, 96, 72, 23, 17, 1920, 1080, 320, 79, 63]
    consts += [float(np.float32(k) + np.float32(1e-6)) for k in range(1, 17)]
    for c in consts:
        cf = np.float32(c)
        rc = np.float32(1.0) / cf
        for scale in (1.0, 16.0, 2000.0):
            x = ((rng.random(400_000, dtype=np.float32) * 2 - 1) * np.float32(scale)).astype(np.float32)
            q = x * rc
            r = _fma32(-q, np.full_like(q, cf), x)
            q1 = _fma32(r, np.full_like(q, rc), q)
            assert np.array_equal(q1, (x / cf).astype(np.float32)), c


def test_rootnet_soft_synthetic_branch_matches_reference_golden():
    """vectorised target volume + rendered root heat-maps == the reference's loops (fixed roots, no noise)"""
    from selfpose3d_amd.cuboid_proposal_net_soft import CuboidProposalNetSoft
    from tests import golden_io as gio
    g = gio.load("rootnet_soft_synth")
    cfg = load_config(None, NETWORK__ROOTNET_ROOTHM=True, NETWORK__ROOTNET_TRAIN_SYNTH=True)
    net = CuboidProposalNetSoft(cfg)
    assert np.allclose(net.lo, g["lo"]) and np.allclose(net.hi, g["hi"])
    u, uz, zn = g["u"], float(g["uz"]), g["zn"]
    lo, hi = g["lo"], g["hi"]
    x = (hi[0] - lo[0]) * torch.from_numpy(u[..., 0:1]) + lo[0]
    y = (hi[1] - lo[1]) * torch.from_numpy(u[..., 1:2]) + lo[1]
    z = ((hi[2] - lo[2]) * torch.full((1, 1, 1), uz) + lo[2]).expand(1, int(g["R"]), 1) + torch.from_numpy(zn) * 50
    roots = torch.cat([x, y, z], -1).float()
    target = net.target_cubes(roots)
    assert float((target - torch.from_numpy(g["target"])).abs().max()) <= 1e-6
    meta = syn.make_meta(1, 5, (960, 512), ssv_style=True)
    meta[0]["trans"] = torch.from_numpy(g["trans"])
    net.noise_std = 0.0
    hms = net.render_root_heatmaps(roots, meta)
    for v in range(5):
        assert hms[v].shape == (1, 1, 128, 240)
        assert float((hms[v][0] - torch.from_numpy(g["hms"][v][0])).abs().max()) <= 2e-5
    # per-GPU batch > 1 works here (the reference is limited to 1)
    r2 = net.sample_roots(3, "cpu", torch.Generator().manual_seed(1))
    assert r2.shape[0] == 3 and net.target_cubes(r2).shape == (3, 80, 80, 20)


def test_torch_op_registry_face():
    """torch.ops.selfpose3d_mi.* exist with the §8(b) schema, infer shapes under FakeTensorMode and refuse CPU tensors
    loudly (the product has no CPU implementation)."""
    import selfpose3d_amd.torch_ops  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    fwd = torch.ops.selfpose3d_mi.unproject_fwd
    schema = str(fwd.default._schema)
    for arg in ("Tensor hm", "Tensor cam", "Tensor centers", "Tensor valid", "float[] grid_size", "cube_size",
                "img_size", "hm_size"):
        assert arg in schema, schema
    assert "-> (Tensor, Tensor)" in schema
    assert "Tensor grad_cubes" in str(torch.ops.selfpose3d_mi.unproject_bwd.default._schema)
    args = lambda hm: (hm, torch.zeros(1, 2, 32), torch.zeros(1, 3), torch.ones(1, dtype=torch.uint8),
                       [100.0, 100.0, 100.0], [4, 4, 8], [48, 32], [12, 8])
    with pytest.raises(NotImplementedError):
        fwd(*args(torch.zeros(2, 1, 8, 12, 16)), 15)
    with FakeTensorMode():
        a, b = fwd(*args(torch.zeros(2, 1, 8, 12, 16)), 15)
        assert tuple(a.shape) == (1, 15, 4, 4, 8) and tuple(b.shape) == (1, 128, 3)
        a, b = fwd(*args(torch.zeros(2, 1, 15, 8, 12)))
        assert tuple(a.shape) == (1, 15, 4, 4, 8)


def test_fft_lengths_for_the_opening_conv():
    """FFT sizes of the frequency-domain 7x7x7 conv: even, >= n + 6, prime factors <= 13, at most one distinct odd
    prime (70 = 2*5*7 and 140 lose to 72 and 144 in rocFFT); the z length is a multiple of 4 (16-byte aligned rows)"""
    from selfpose3d_amd.v2v_net import _FoldedV2V
    for n, want in ((86, 88), (26, 26), (70, 72), (22, 22), (14, 14), (46, 48), (134, 144)):
        m = _FoldedV2V._fft_len(n)
        assert m == want and m % 2 == 0 and m >= n
        r, odd = m, 0
        for q in (2, 3, 5, 7, 11, 13):
            odd += 1 if (q > 2 and r % q == 0) else 0
            while r % q == 0:
                r //= q
        assert r == 1 and odd <= 1
    assert _FoldedV2V._fft_shape(80, 80, 20, 7) == (88, 88, 28)
    assert _FoldedV2V._fft_shape(64, 64, 64, 7) == (72, 72, 72)


def test_project_joints_matches_reference_project_pose_batch():
    """vectorised reprojection of predicted poses == cameras.project_pose_batch of the reference (golden generated by
    tests/golden/make_goldens.py: random rigs, augmented crops, ragged people per sample)"""
    from selfpose3d_amd.camera_pack import pack_cameras
    from selfpose3d_amd.reprojection import project_joints
    from tests import golden_io as gio
    g = gio.load("project_pose_batch")
    people = [int(n) for n in g["people"]]
    B, V, P = len(people), 5, max(people)
    meta = syn.random_meta(B, V, (960, 512), seed=5, augment=True, ssv_style=True)
    cam = torch.from_numpy(pack_cameras(meta, B, (960, 512)))
    joints = torch.zeros(B, P, 15, 3)
    for b in range(B):
        joints[b, :people[b]] = torch.from_numpy(g[f"pose{b}"])
    px = project_joints(joints, cam, stride=1.0, trans=torch.from_numpy(g["trans"]))      # (V,B,P,J,2)
    for b in range(B):
        ref = g[f"px{b}"]                                                                  # (V, people, J, 2)
        got = px[:, b, :people[b]].numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= 2e-5 * scale, (b, np.abs(got - ref).max())


def test_v2v_inference_plan_key_sees_every_weight_change():
    """the folded plan must not survive memory-format changes, in-place writes, entering train() or load_state_dict"""
    from selfpose3d_amd.v2v_net import V2VNet, _FoldedV2V
    net = V2VNet(2, 1).eval()
    k0 = _FoldedV2V._key(net)
    assert k0 == _FoldedV2V._key(net)
    net.to(memory_format=torch.channels_last_3d)
    k1 = _FoldedV2V._key(net)
    assert k1 != k0
    with torch.no_grad():
        net.encoder_decoder.mid_res.res_branch[0].weight.mul_(2.0)
    assert _FoldedV2V._key(net) != k1
    for action in (lambda: net.train(), lambda: net.load_state_dict(net.state_dict()), lambda: net.invalidate_plan()):
        net.eval()
        net._plan = object()
        action()
        assert net._plan is None
    # eval() on a module that already is in eval mode (every validation pass calls it) keeps the plan and its
    # padded buffers; weight edits made in eval mode are caught by the (data_ptr, version) key checked above
    net.eval()
    marker = net._plan = object()
    net.eval()
    assert net._plan is marker
    net._plan = None


def test_padded_fft_input_view_is_recognised_by_address_not_by_python_attribute():
    """get_voxel(out=view) returns the buffer through an autograd Function; under no_grad that re-wraps the
    tensor into a new Python object without the old `_sp3d_fft_shape` tag, and the opening conv then copied the padded
    buffer's corner onto itself.  The plan now recognises its own buffer by data_ptr + strides."""
    from selfpose3d_amd.v2v_net import V2VNet, _FoldedV2V
    net = V2VNet(2, 1).eval()
    plan = _FoldedV2V(net)
    view = plan.fft_input_view(3, 8, 8, 4, torch.device("cpu"))
    assert view is not None and tuple(view.shape) == (3, 2, 8, 8, 4)
    S = plan._fft_shape(8, 8, 4, 7)

    class Fill(torch.autograd.Function):
        @staticmethod
        def forward(ctx, out):
            out.fill_(1.0)
            return out

    with torch.no_grad():
        back = Fill.apply(view)
    assert plan._is_padded_view(back, 2, S)                      # same memory, whatever Python object
    assert plan._is_padded_view(view[1:], 2, S)                  # whole samples further into the buffer
    assert not plan._is_padded_view(view.clone(), 2, S)          # a copy is not the buffer
    assert not plan._is_padded_view(view[:, :1], 1, S)           # wrong channel count
    assert not plan._is_padded_view(view[:, :, 1:], 2, S)        # not the corner


def test_three_piece_weight_splits_are_exact_and_laid_out_as_documented():
    """_lib.wino_weights_split / conv_weights_split (host side of the split-product kernels): hi + mid + lo reproduces the
    fp32 weights bit for bit (8+8+8 mantissa bits), pieces are bf16, records follow the layout include/sp3d.h documents"""
    import torch
    from selfpose3d_amd import _lib
    g = torch.Generator().manual_seed(5)
    w = torch.randn(32, 16, 3, 3, 3, generator=g) * 0.07
    w[0, 0, 0, 0, 0] = 0.0
    w[1, 2, 1, 1, 1] = 1e-30          # deep in the subnormal range of the low pieces: must still add up
    U = _lib.wino_weights(w)
    for chunk in (8, 16):
        U3 = _lib.wino_weights_split(U, chunk)
        assert U3.dtype == torch.bfloat16 and tuple(U3.shape) == (64, 16 // chunk, chunk // 4, 32, 3, 4)
        mid, hi, lo = U3[..., 0, :].float(), U3[..., 1, :].float(), U3[..., 2, :].float()
        back = ((hi + mid) + lo).permute(0, 1, 2, 4, 3).reshape(64, 16, 32)     # [p, chunk, group, q, o] -> [p, c, o]
        assert torch.equal(back, U)
    W3 = _lib.conv_weights_split(w)
    assert W3.dtype == torch.bfloat16 and tuple(W3.shape) == (27, 2, 2, 32, 6, 4)
    hi, lo, hi2, hi3, mid, mid2 = (W3[..., i, :].float() for i in range(6))
    assert torch.equal(hi, hi2) and torch.equal(hi, hi3) and torch.equal(mid, mid2)
    back = ((hi + mid) + lo).permute(0, 1, 2, 4, 3).reshape(27, 16, 32)
    want = w.permute(4, 3, 2, 1, 0).reshape(27, 16, 32)                         # tap = kz*9 + ky*3 + kx
    assert torch.equal(back, want)


def test_view_batchnorm_equals_the_per_view_loop():
    """ViewBatchNorm2d on a sample-major (B*V) batch == BatchNorm2d called once per view (the reference's loop over cameras,
    lib/models/multi_person_posenet.py:44-47): outputs, input / weight / bias gradients, and the running statistics after
    the V sequential momentum updates - float64 on the CPU, to rounding"""
    import copy
    import torch
    from selfpose3d_amd.pose_resnet import ViewBatchNorm2d
    torch.manual_seed(0)
    V, B, C, H, W = 5, 2, 7, 6, 5
    bn = ViewBatchNorm2d(C, momentum=0.1).double()
    bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.uniform_(-1, 1)
    bn.running_mean.uniform_(-1, 1); bn.running_var.uniform_(0.5, 2)
    ref = copy.deepcopy(bn)
    views = [torch.randn(B, C, H, W, dtype=torch.double, requires_grad=True) for _ in range(V)]
    wgt = [torch.randn(B, C, H, W, dtype=torch.double) for _ in range(V)]
    sum((ref(v) * w).sum() for v, w in zip(views, wgt)).backward()
    views2 = [v.detach().clone().requires_grad_(True) for v in views]
    bn.views = V
    y = bn(torch.stack(views2, 1).flatten(0, 1)).view(B, V, C, H, W)
    sum((y[:, i] * w).sum() for i, w in enumerate(wgt)).backward()
    outs_ref = [copy.deepcopy(ref).train()(v) for v in views]                    # statistics of each view alone
    for i in range(V):
        assert (y[:, i] - outs_ref[i]).abs().max() < 1e-12
        assert (views2[i].grad - views[i].grad).abs().max() < 1e-10
    assert (bn.weight.grad - ref.weight.grad).abs().max() < 1e-9 and (bn.bias.grad - ref.bias.grad).abs().max() < 1e-9
    assert (bn.running_mean - ref.running_mean).abs().max() < 1e-13 and (bn.running_var - ref.running_var).abs().max() < 1e-13
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == V
    # views = 1 / eval mode: plain BatchNorm2d
    bn.views = 1
    x = torch.randn(3, C, H, W, dtype=torch.double)
    assert torch.equal(bn.eval()(x), ref.eval()(x)) or (bn.eval()(x) - ref.eval()(x)).abs().max() < 1e-12


def test_forward_views_keeps_the_per_view_loop_where_the_batched_pass_cannot_run():
    """CPU tensors, a single view, partly frozen BatchNorm or batch_views_in_training = False: PoseResNet.forward_views in
    train mode is the reference's loop over cameras; set_backbone_memory_format leaves a training backbone in NCHW"""
    import torch
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd import pose_resnet as pr
    cfg = load_config(None)
    torch.manual_seed(0)
    net = pr.PoseResNet(cfg, 18).train()
    views = [torch.randn(2, 3, 32, 48) for _ in range(3)]
    calls = []
    orig = net.forward
    net.forward = lambda x, *a, **k: (calls.append(tuple(x.shape)), orig(x, *a, **k))[1]
    out = net.forward_views(views)
    assert [tuple(o.shape[:2]) for o in out] == [(2, int(cfg.NETWORK.NUM_JOINTS))] * 3
    assert calls == [(2, 3, 32, 48)] * 3                                  # CPU: one call per view
    assert all(m.views == 1 for m in net.modules() if isinstance(m, pr.ViewBatchNorm2d))
    assert pr.set_backbone_memory_format(net, True).conv1.weight.is_contiguous()            # training + batching: stays NCHW
    net.batch_views_in_training = False
    assert not pr.set_backbone_memory_format(net, True).conv1.weight.is_contiguous()        # loop mode: channels_last is fine
    assert pr.set_backbone_memory_format(net, False).conv1.weight.is_contiguous()
    # state_dict keys are BatchNorm2d's
    keys = [k for k in net.state_dict() if k.startswith("bn1.")]
    assert keys == ["bn1.weight", "bn1.bias", "bn1.running_mean", "bn1.running_var", "bn1.num_batches_tracked"]


def test_shared_gpu_flavour_selection(monkeypatch):
    """SP3D_SHARED_GPU picks the library flavour; unset, more local ranks than GPUs (torchrun's LOCAL_WORLD_SIZE) means sharing"""
    monkeypatch.delenv("SP3D_SHARED_GPU", raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    assert _lib.shared_gpu() is False
    monkeypatch.setenv("SP3D_SHARED_GPU", "1")
    assert _lib.shared_gpu() is True
    monkeypatch.setenv("SP3D_SHARED_GPU", "0")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert _lib.shared_gpu() is False                         # an explicit 0 wins
    monkeypatch.delenv("SP3D_SHARED_GPU")
    for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.warns(UserWarning, match="share a GPU"):
        assert _lib.shared_gpu() is True                      # 8 local ranks, 1 GPU
    # a launcher that narrows every rank's view to its own GPU: 8 local ranks, device_count() == 1, and NOBODY shares
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "3")
    assert _lib.shared_gpu() is False
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert _lib.shared_gpu() is False
    assert os.path.basename(_lib.NOPK_LIB_PATH) == "libsp3d_nopk.so" and os.path.exists(_lib.NOPK_LIB_PATH)


def test_nopk_flavour_exports_the_same_abi(lib):
    """libsp3d_nopk.so (shared GPUs) is the same C ABI: every symbol of include/sp3d.h, same ABI version"""
    nopk = C.CDLL(_lib.NOPK_LIB_PATH)
    for name in _lib.EXPORTS:
        assert hasattr(nopk, name), name
    nopk.sp3d_abi_version.restype = C.c_int
    assert nopk.sp3d_abi_version() == lib.sp3d_abi_version()


def test_packed_source1_rewrite_of_the_build():
    """selfpose3d_amd/pk_src1.py: source 0 and source 1 of an affected packed instruction trade places with their modifier
    bits; everything else passes through untouched; an instruction the exchange cannot help stops the build"""
    from selfpose3d_amd import pk_src1
    asm = "\n".join([
        "\tv_pk_fma_f32 v[6:7], s[6:7], v[2:3], v[6:7] op_sel:[0,1,0]",
        "\tv_pk_mul_f32 v[2:3], v[2:3], s[68:69] op_sel:[0,1] op_sel_hi:[0,0]",
        "\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] ; a comment",
        "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0,1]",
        "\tv_pk_mul_f32 v[12:13], v[12:13], 0.5 op_sel_hi:[1,0]",
        "\tv_fma_f32 v0, v1, v2, v3",
        "\tv_pk_add_f16 v0, v1, v2 op_sel:[0,1]"]) + "\n"
    assert pk_src1.count_risky(asm) == 3
    fixed, n = pk_src1.fix_asm(asm)
    assert n == 3 and pk_src1.count_risky(fixed) == 0
    lines = fixed.splitlines()
    assert lines[0].strip() == "v_pk_fma_f32 v[6:7], v[2:3], s[6:7], v[6:7] op_sel:[1,0,0]"
    assert lines[1].strip() == "v_pk_mul_f32 v[2:3], s[68:69], v[2:3] op_sel:[1,0] op_sel_hi:[0,0]"
    assert lines[2].strip() == "v_pk_add_f32 v[0:1], v[4:5], v[2:3] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0] ; a comment"
    assert lines[3:] == asm.splitlines()[3:]                     # unaffected forms, other instructions: byte for byte
    with pytest.raises(ValueError, match="does not help"):
        pk_src1.fix_asm("\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,1]\n")
    # llvm-objdump text (address prefix or encoding comment) counts the same way
    assert pk_src1.count_risky("\tv_pk_fma_f32 v[6:7], v[2:3], v[8:9], v[6:7] op_sel:[0,1,0] // 00000002B068: D3B04806 1C180D02\n") == 1


def test_no_packed_instruction_of_either_library_reads_the_high_half_of_source_1(tmp_path):
    """the finished libraries, disassembled: the default flavour keeps its ~14 000 packed-fp32 instructions, none of them in the
    form that goes wrong next to matrix instructions of another kernel (tools/mfma_pk_hazard5.hip); the nopk flavour has none"""
    import shutil
    import subprocess
    from selfpose3d_amd import pk_src1
    objdump = os.path.join(sbuild.LLVM_BIN, "llvm-objdump")
    for path, want_packed in ((_lib._DEFAULT_LIB_PATH, True), (_lib.NOPK_LIB_PATH, False)):
        d = tmp_path / os.path.basename(path)
        d.mkdir()
        shutil.copy(path, d / "lib.so")
        subprocess.run([objdump, "--offloading", "lib.so"], cwd=d, check=True, capture_output=True, timeout=300)
        objs = sorted(p for p in os.listdir(d) if p.endswith("gfx950"))
        assert len(objs) >= 8, objs
        text = "".join(subprocess.run([objdump, "-d", o], cwd=d, check=True, capture_output=True, text=True, timeout=300).stdout for o in objs)
        packed = len(re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b", text))
        assert (packed > 10000) if want_packed else (packed == 0), (path, packed)
        assert pk_src1.count_risky(text) == 0, path


def test_device_assembly_has_no_load_wait_store_chains_and_no_scratch(lib):
    """tools/isa_scan.py over the assembly the build keeps: no kernel may contain >= 8 `load -> s_waitcnt vmcnt(0) -> store`
    chains (the epilogue shape that cost the half-resolution Winograd kernel 9 of its 81 us until round 5,
    profiles/r05_epilogue_fix.md), none outside the scan's by-design list may wait for most of its loads one at a time, and
    none may spill registers to scratch memory."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_scan", os.path.join(ROOT, "tools", "isa_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    recs = mod.scan()
    if len(recs) < 150:                      # library present but its object directory is not (a copied tree): compile again
        sbuild.build(force=True)
        recs = mod.scan()
    assert len(recs) >= 150, "the build keeps its device assembly under selfpose3d_amd/build/obj"
    bad = [r for r in recs if r["flag"]]
    assert not bad, bad
    # the scan does see the pattern: a synthetic kernel body with ten chains is flagged
    body = ["_ZN4sp3d4fakeEv:"] + ["\tglobal_load_dword v1, v[2:3], off", "\ts_waitcnt vmcnt(0)", "\tv_add_f32 v1, v1, v4",
                                    "\tglobal_store_dword v[5:6], v1, off"] * 10 + ["; ScratchSize: 0"]
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "fake.s"), "w") as f:
            f.write("\n".join(body))
        old = mod.OBJ
        mod.OBJ = d
        try:
            fake = mod.scan()
        finally:
            mod.OBJ = old
    assert len(fake) == 1 and fake[0]["load_wait_store_chains"] == 10 and fake[0]["flag"]


def _pack_cameras_per_view(meta, batch, img_size, flip_xcoords=None):
    """the camera pack as it was written first - one view at a time, field by field - kept as the yardstick of the batched form"""
    from selfpose3d_amd import camera_pack as cp
    V, B = len(meta), int(batch)
    tab = np.zeros((B, V, cp.CAM_STRIDE), np.float32)
    flips = None if flip_xcoords is None else cp._np(flip_xcoords).astype(bool).reshape(B)
    for c in range(V):
        m = meta[c]
        cam = m["camera"]
        center = cp._np(m["center"], np.float64).reshape(B, 2)
        scale = cp._np(m["scale"]).reshape(B, -1)
        if scale.shape[1] == 1:
            scale = np.repeat(scale, 2, 1)
        rot = cp._np(m["rotation"], np.float64).reshape(B)
        A = cp.get_affine_transform_batch(center, scale.astype(np.float32), rot, img_size)
        tab[:, c, cp.CAM_R:cp.CAM_R + 9] = cp._np(cam["R"], np.float32).reshape(B, 9)
        tab[:, c, cp.CAM_T:cp.CAM_T + 3] = cp._np(cam["T"], np.float32).reshape(B, 3)
        tab[:, c, cp.CAM_F] = cp._np(cam["fx"], np.float32).reshape(B)
        tab[:, c, cp.CAM_F + 1] = cp._np(cam["fy"], np.float32).reshape(B)
        tab[:, c, cp.CAM_C] = cp._np(cam["cx"], np.float32).reshape(B)
        tab[:, c, cp.CAM_C + 1] = cp._np(cam["cy"], np.float32).reshape(B)
        tab[:, c, cp.CAM_K:cp.CAM_K + 3] = cp._np(cam["k"], np.float32).reshape(B, 3)
        tab[:, c, cp.CAM_P:cp.CAM_P + 2] = cp._np(cam["p"], np.float32).reshape(B, 2)
        tab[:, c, cp.CAM_A:cp.CAM_A + 6] = A.astype(np.float32).reshape(B, 6)
        tab[:, c, cp.CAM_W0] = (center[:, 0] * 2.0).astype(np.float32)
        tab[:, c, cp.CAM_H0] = (center[:, 1] * 2.0).astype(np.float32)
        if flips is not None:
            tab[:, c, cp.CAM_FLIP] = flips.astype(np.float32)
    return cp.finish(tab)


@pytest.mark.parametrize("B,V", [(4, 5), (1, 3), (2, 10), (3, 4)])
def test_batched_camera_pack_equals_the_per_view_form_bit_for_bit(B, V):
    """pack_cameras gathers every field of all views with one stack and solves all B x V crop affines in one call (the per-step
    host cost of a graphed step); the table must be the per-view form's, byte for byte: augmented crops, flips, scalar scales,
    numpy entries mixed with tensors (the fallback of the field gather)"""
    from selfpose3d_amd.camera_pack import pack_cameras
    cfg = load_config(None)
    img = [int(v) for v in cfg.NETWORK.IMAGE_SIZE]
    meta = syn.make_meta(B, V, img)
    rng = np.random.default_rng(B * 10 + V)
    for m in meta:
        m["rotation"] = torch.tensor(rng.uniform(-40, 40, B))
        m["scale"] = torch.tensor(rng.uniform(3, 9, (B, 2)).astype(np.float32))
        m["center"] = torch.tensor(rng.uniform(200, 800, (B, 2)))
    for flip in (None, torch.tensor(rng.integers(0, 2, B).astype(bool))):
        assert pack_cameras(meta, B, img, flip).tobytes() == _pack_cameras_per_view(meta, B, img, flip).tobytes()
    for m in meta:
        m["scale"] = torch.tensor(rng.uniform(3, 9, (B,)).astype(np.float32))          # scalar scale -> [s, s]
    assert pack_cameras(meta, B, img).tobytes() == _pack_cameras_per_view(meta, B, img).tobytes()
    for i, m in enumerate(meta):                                                        # numpy entries: the gather falls back
        m["center"] = m["center"].numpy()
        if i % 2:
            m["camera"]["R"] = m["camera"]["R"].numpy()
    assert pack_cameras(meta, B, img).tobytes() == _pack_cameras_per_view(meta, B, img).tobytes()


def test_build_survives_an_unrepairable_packed_form_and_has_a_plain_hipcc_path(tmp_path, monkeypatch):
    """round-5 advice: the assembly rewrite must not be able to fail the build.  (a) the rewrite refuses an instruction ->
    that source is recompiled without packed-fp32 instructions (a warning, an object, nothing risky inside); (b)
    SP3D_PLAIN_HIPCC=1 -> one plain `hipcc -c` per source, no packed-fp32 instructions, no hand-rolled bundling steps"""
    from selfpose3d_amd import build, pk_src1
    src = "sp3d_fftconv.hip"                                    # the smallest source: seconds to compile
    cflags = [f for f in build.FLAGS if f != "-shared"]
    real, calls = pk_src1.fix_asm, {"n": 0}

    def refuse_once(text):
        calls["n"] += 1
        if calls["n"] == 1:
            raise ValueError("both multiplicands take their low result from a high half (simulated)")
        assert "v_pk_fma_f32" not in text and "v_pk_mul_f32" not in text and "v_pk_add_f32" not in text
        return real(text)
    monkeypatch.setattr(pk_src1, "fix_asm", refuse_once)
    with pytest.warns(UserWarning, match="without packed-fp32"):
        obj = build._compile_one(src, str(tmp_path), cflags, False)
    assert calls["n"] == 2 and os.path.getsize(obj) > 1000
    monkeypatch.setattr(pk_src1, "fix_asm", real)
    monkeypatch.setenv("SP3D_PLAIN_HIPCC", "1")
    plain = tmp_path / "plain"
    plain.mkdir()
    obj2 = build._compile_one(src, str(plain), cflags, False)
    assert os.path.getsize(obj2) > 1000 and sorted(os.listdir(plain)) == [os.path.basename(obj2)]      # no intermediate files
