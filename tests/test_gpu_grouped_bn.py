"""Grouped training-mode BatchNorm (sp3d_gbn_forward / _backward, selfpose3d_amd/grouped_bn.py) against what it replaces:
the reference's per-call BatchNorm - one ``nn.BatchNorm`` call per candidate slot
(/root/reference/lib/models/multi_person_posenet.py:84-88 over /root/reference/lib/models/v2v_net.py:10-45) or per camera
(multi_person_posenet.py:44-47).  Proven the way ViewBatchNorm2d was: in FLOAT64 on the GPU, where the two are the same
function up to summation order - outputs <= 1e-9, gradients <= 1e-7, running statistics equal - then the fp32 kernels
against the float64 loop, then the whole V2VNet and the model's train step with slots batched against the per-slot loop."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _loop_reference(x, group_of, G, weight, bias, relu, dims, momentum=0.1, eps=1e-5, n_update=None):
    """per-group nn.BatchNorm calls in group order on float64 copies -> (y, dx, dw, db, running_mean, running_var, tracked)"""
    C = x.shape[1]
    bn = (nn.BatchNorm3d if dims == 3 else nn.BatchNorm2d)(C, eps=eps, momentum=momentum).to(x.device).double().train()
    with torch.no_grad():
        bn.weight.copy_(weight.double())
        bn.bias.copy_(bias.double())
    xd = x.detach().double().contiguous().requires_grad_(True)
    y = torch.zeros_like(xd)
    go = torch.as_tensor(group_of, device=x.device)
    n_update = G if n_update is None else n_update
    for g in range(G):
        idx = torch.nonzero(go == g).flatten()
        if idx.numel() == 0:
            continue
        if g >= n_update:                                   # a padding group: normalised, but no running update
            keep = (bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone())
        yg = bn(xd[idx])
        if g >= n_update:
            with torch.no_grad():
                bn.running_mean.copy_(keep[0]); bn.running_var.copy_(keep[1]); bn.num_batches_tracked.copy_(keep[2])
        y = y.index_put((idx,), torch.relu(yg) if relu else yg)
    return bn, xd, y


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-300))


def _worst_param_grad(net_ref, net_got, floor):
    """largest gradient error over the parameters, each relative to max(its own largest reference gradient, floor x the
    largest gradient of the net): a convolution bias in front of a BatchNorm has an analytically ZERO gradient (the layer
    removes the mean) - what both runs hold there is rounding noise, not a number to compare"""
    pairs = [(n, pl.grad, po.grad) for (n, pl), (_, po) in zip(net_ref.named_parameters(), net_got.named_parameters())
             if pl.grad is not None]
    top = max(float(gl.detach().double().abs().max()) for _, gl, _ in pairs)
    worst, where = 0.0, None
    for n, gl, go in pairs:
        assert go is not None, n
        gl, go = gl.detach().double(), go.detach().double()
        e = float((go - gl).abs().max() / max(float(gl.abs().max()), floor * top))
        if e > worst:
            worst, where = e, n
    return worst, where


CASES = [
    # dims, spatial, C, group_of, relu
    (3, (8, 8, 8), 32, [0, 0, 1, 1, 2], True),                     # the bench's slots: 2 + 2 + 1 cubes
    (3, (8, 6, 5), 16, [0, 1, 1, 2, 2, 2, 3], False),              # ragged, odd spatial size
    (3, (4, 4, 4), 128, [1, 0, 1, 0, 2, 1], True),                 # interleaved groups (two view sets per slot)
    (3, (6, 6, 6), 64, [0, 0, 0], False),                          # one group = plain BatchNorm
    (2, (12, 10), 64, [0, 1, 2, 0, 1, 2], True),                   # cameras: n % V
    (2, (6, 4), 256, [0, 1, 0, 1], False),
    (2, (3, 4), 2048, [0, 1, 1, 0], True),                         # two 16-byte column passes per thread (float32)
    (2, (5, 7), 48, [0, 0, 1], True),                              # 12 columns: 256 is not a multiple
]


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_grouped_batchnorm_equals_the_per_group_loop(dev, case, dtype):
    from selfpose3d_amd.grouped_bn import GroupedBatchNorm2d, GroupedBatchNorm3d, GroupSpec
    dims, spatial, C, group_of, relu = CASES[case]
    if dtype == torch.float64 and C > 1024:
        pytest.skip("float64 rows: C <= 1024")
    G = max(group_of) + 1
    sizes = [group_of.count(g) for g in range(G)]
    gen = torch.Generator().manual_seed(100 + case)
    N = len(group_of)
    fmt = torch.channels_last_3d if dims == 3 else torch.channels_last
    x = (torch.randn((N, C) + spatial, generator=gen) * torch.rand((1, C) + (1,) * dims, generator=gen) * 3 +
         torch.randn((1, C) + (1,) * dims, generator=gen) * 2)              # per-channel scale and offset (mean^2 ~ var)
    w, b = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen)
    gy = torch.randn((N, C) + spatial, generator=gen)
    x, w, b, gy = (t.to(dev) for t in (x, w, b, gy))
    ref_bn, xd, yref = _loop_reference(x, group_of, G, w, b, relu, dims)
    (yref * gy.double()).sum().backward()

    bn = (GroupedBatchNorm3d if dims == 3 else GroupedBatchNorm2d)(C).to(dev).to(dtype).train()
    with torch.no_grad():
        bn.weight.copy_(w)
        bn.bias.copy_(b)
    bn.groups = GroupSpec(sizes, dev, group_of=group_of)
    xin = x.to(dtype).contiguous(memory_format=fmt).requires_grad_(True)
    y = bn.grouped_forward(xin, relu=relu)
    assert y.is_contiguous(memory_format=fmt)
    (y * gy.to(dtype)).sum().backward()
    if dtype == torch.float64:
        tol_y, tol_g, tol_s = 1e-9, 1e-7, 1e-12
    else:                               # fp32 kernels against the float64 loop: fp32 rounding of x * scale + shift
        tol_y, tol_g, tol_s = 1e-5, 5e-5, 1e-5
    assert _rel(y, yref) <= tol_y
    assert _rel(xin.grad, xd.grad) <= tol_g
    assert _rel(bn.weight.grad, ref_bn.weight.grad) <= tol_g
    assert _rel(bn.bias.grad, ref_bn.bias.grad) <= tol_g
    assert _rel(bn.running_mean, ref_bn.running_mean) <= tol_s
    assert _rel(bn.running_var, ref_bn.running_var) <= tol_s
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == G


def test_padding_group_leaves_running_statistics_alone_and_eval_is_plain_batchnorm(dev):
    from selfpose3d_amd.grouped_bn import GroupedBatchNorm3d, GroupSpec
    gen = torch.Generator().manual_seed(7)
    C, group_of = 32, [0, 0, 1, 2, 2, 2]                       # group 2 = three zero padding cubes
    x = torch.randn(6, C, 4, 4, 4, generator=gen).to(dev)
    x[3:] = 0.0
    w, b = (torch.rand(C, generator=gen) + 0.5).to(dev), torch.randn(C, generator=gen).to(dev)
    ref_bn, xd, yref = _loop_reference(x, group_of, 3, w, b, False, 3, n_update=2)
    bn = GroupedBatchNorm3d(C).to(dev).double().train()
    with torch.no_grad():
        bn.weight.copy_(w); bn.bias.copy_(b)
    bn.groups = GroupSpec([2, 1, 3], dev, group_of=group_of, n_update=2)
    y = bn(x.double().contiguous(memory_format=torch.channels_last_3d))
    assert _rel(y[:3], yref[:3]) <= 1e-9
    assert torch.isfinite(y).all()                              # var = 0 in the padding group: invstd = 1/sqrt(eps), x - mean = 0
    assert _rel(bn.running_mean, ref_bn.running_mean) <= 1e-12 and _rel(bn.running_var, ref_bn.running_var) <= 1e-12
    assert int(bn.num_batches_tracked) == 2
    bn.eval()                                                   # eval mode ignores the spec: running statistics, as nn.BatchNorm3d
    ye = bn(x.double())
    ref_bn.eval()
    assert _rel(ye, ref_bn(x.double())) <= 1e-12
    bn.train()
    bn.groups = None                                            # no spec: plain train-mode BatchNorm3d over the batch
    plain = nn.BatchNorm3d(C).to(dev).double().train()
    with torch.no_grad():
        plain.weight.copy_(w); plain.bias.copy_(b)
    assert _rel(bn(x.double()), plain(x.double())) <= 1e-12


def test_bad_arguments_are_refused(dev):
    from selfpose3d_amd import _lib
    from selfpose3d_amd.grouped_bn import GroupedBatchNorm3d, GroupSpec
    with pytest.raises(ValueError):
        GroupSpec([2, 1], dev, group_of=[0, 0, 0])
    bn = GroupedBatchNorm3d(6).to(dev).train()                 # 6 channels: not a multiple of 4
    bn.groups = GroupSpec([2], dev)
    with pytest.raises(_lib.Sp3dError, match="not implemented|unsupported|EUNSUPPORTED|-4"):
        bn(torch.randn(2, 6, 4, 4, 4, device=dev))
    bn = GroupedBatchNorm3d(8).to(dev).train()
    bn.groups = GroupSpec([2, 1], dev)
    with pytest.raises(_lib.Sp3dError, match="describes 3"):
        bn(torch.randn(2, 8, 4, 4, 4, device=dev))
    with pytest.raises(_lib.Sp3dError, match="GPU"):
        bn(torch.randn(3, 8, 4, 4, 4))


def _v2v_pair(dev, dtype, seed=11):
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.v2v_net import V2VNet
    nets = []
    for _ in range(2):
        net = V2VNet(16, 15)
        syn.fill_parameters_deterministic(net, seed=seed, scale=0.08)
        nets.append(net.to(dev).to(dtype).train())
    return nets


def test_v2vnet_all_slots_in_one_call_equals_the_per_slot_loop_float64(dev):
    """the whole network: per-slot calls (the reference's loop) against ONE call with a GroupSpec, float64 on the GPU"""
    from selfpose3d_amd.grouped_bn import GroupSpec, bn_groups
    loop_net, one_net = _v2v_pair(dev, torch.float64)
    gen = torch.Generator().manual_seed(3)
    sizes = [2, 2, 1]
    x = torch.rand(5, 16, 16, 16, 16, generator=gen, dtype=torch.float64).to(dev)
    gy = torch.randn(5, 15, 16, 16, 16, generator=gen, dtype=torch.float64).to(dev)
    outs, s0 = [], 0
    for n in sizes:                                            # multi_person_posenet.py:84-88
        outs.append(loop_net(x[s0:s0 + n]))
        s0 += n
    yl = torch.cat(outs, 0)
    (yl * gy).sum().backward()
    with bn_groups(one_net, GroupSpec(sizes, dev)):
        yo = one_net(x.contiguous(memory_format=torch.channels_last_3d))
    (yo * gy).sum().backward()
    assert _rel(yo, yl) <= 1e-9
    worst, where = _worst_param_grad(loop_net, one_net, 1e-6)
    assert worst <= 1e-7, (worst, where)
    for (name, bl), (_, bo) in zip(loop_net.named_buffers(), one_net.named_buffers()):
        if name.endswith("num_batches_tracked"):
            assert int(bl) == int(bo) == 3, name
        else:
            assert _rel(bo, bl) <= 1e-10, name
    from selfpose3d_amd.grouped_bn import GroupedBatchNorm3d
    assert all(m.groups is None for m in one_net.modules() if isinstance(m, GroupedBatchNorm3d))     # the spec is detached again


@pytest.mark.miopen_sensitive            # fp32 through the library's convolutions at two batch compositions: collected last
@pytest.mark.parametrize("ssv_sets", [1, 2])
def test_pose_net_forward_slots_equals_the_loop_fp32(dev, ssv_sets):
    """PoseRegressionNet.forward_slots (one indexed unprojection per view set, one V2V pass, grouped BatchNorm) against the
    loop of PoseRegressionNet.forward calls in the reference's order: poses, heat-map gradients, running statistics"""
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.pose_regression_net import PoseRegressionNet
    B, V, J, img, hm = 2, 3, 15, (128, 96), (32, 24)
    cfg = load_config(None, NETWORK__IMAGE_SIZE=list(img), NETWORK__HEATMAP_SIZE=list(hm), NETWORK__NUM_JOINTS=J,
                      PICT_STRUCT__CUBE_SIZE=[16, 16, 16])
    def make():             # (no deepcopy: a ProjectLayer may hold device events)
        net = PoseRegressionNet(cfg)
        syn.fill_parameters_deterministic(net, seed=5, scale=0.08)
        return net.to(dev).train().use_channels_last(True)
    net_l, net_o = make(), make()
    sets_l, sets_o = [], []
    for s in range(ssv_sets):
        meta = syn.make_meta(B, V, img, rotations=[0.0, 20.0 * s], scale_mults=[1.0, 1.0 + 0.2 * s], ssv_style=True)
        base, _ = syn.people_heatmaps(B, V, J, hm[1], hm[0], img, seed=21 + s)
        flip = torch.tensor([False, bool(s)])
        sets_l.append(([h.to(dev).requires_grad_(True) for h in base], meta, flip))
        sets_o.append(([h.to(dev).requires_grad_(True) for h in base], meta, flip))
    gc = torch.zeros(B, 4, 5, device=dev)
    gc[:, :, 3] = -1.0
    gc[0, :3, :3] = torch.tensor([[0.0, -500.0, 900.0], [600.0, -900.0, 800.0], [-700.0, 100.0, 1000.0]], device=dev)
    gc[1, :2, :3] = torch.tensor([[300.0, -300.0, 850.0], [-400.0, -800.0, 950.0]], device=dev)
    gc[0, :3, 3], gc[1, :2, 3] = torch.arange(3.0, device=dev), torch.arange(2.0, device=dev)
    wgt = torch.randn(ssv_sets, B, 4, J, 3, generator=torch.Generator().manual_seed(9)).to(dev)
    # the reference's loop: slot by slot, view set by view set
    preds_l = [torch.zeros(B, 4, J, 3, device=dev) for _ in range(ssv_sets)]
    for n in range(4):
        if bool((gc[:, n, 3] >= 0).any()):
            for s, (hms, meta, flip) in enumerate(sets_l):
                preds_l[s] = preds_l[s].index_put((torch.arange(B, device=dev), torch.full((B,), n, device=dev)),
                                                  net_l(hms, meta, gc[:, n], flip_xcoords=flip))
    sum((p * wgt[s]).sum() for s, p in enumerate(preds_l)).backward()
    assert net_o.can_batch_slots()
    preds_o = net_o.forward_slots(sets_o, gc)
    sum((p * wgt[s]).sum() for s, p in enumerate(preds_o)).backward()
    for s in range(ssv_sets):
        d = float((preds_o[s] - preds_l[s]).detach().abs().max())
        # mm, on 2000 mm cubes: the library's fp32 convolution kernels differ by batch size, and the soft-argmax of a
        # random-weight net amplifies their rounding (the float64 test above is the exact proof; a pooled or misassigned
        # statistic moves joints by centimetres)
        assert d <= 2.0, d
        for hl, ho in zip(sets_l[s][0], sets_o[s][0]):
            assert _rel(ho.grad, hl.grad) <= 5e-2
    for (name, bl), (_, bo) in zip(net_l.named_buffers(), net_o.named_buffers()):
        if name.endswith("num_batches_tracked"):
            assert int(bl) == int(bo) == 3 * ssv_sets, name
        else:
            assert _rel(bo, bl) <= 1e-3, name
    worst, where = _worst_param_grad(net_l, net_o, 1e-3)
    assert worst <= 5e-2, (worst, where)
    # padding to a listed cube count: zero cubes in their own group change nothing
    net_p, net_q = make(), make()
    assert net_p.slot_pad_sizes == "auto"                    # default: multiples of ceil(B K / 4) -> four batch sizes at most
    net_p.slot_pad_sizes = (8, 16)
    net_q.slot_pad_sizes = None
    with torch.no_grad():
        pp = net_p.forward_slots([(sets_o[0][0], sets_o[0][1], sets_o[0][2])], gc)[0]
        pq = net_q.forward_slots([(sets_o[0][0], sets_o[0][1], sets_o[0][2])], gc)[0]
    assert float((pp - pq).abs().max()) <= 2.0
    for (name, bq), (_, bp) in zip(net_q.named_buffers(), net_p.named_buffers()):
        if name.endswith("num_batches_tracked"):
            assert int(bq) == int(bp), name
        else:
            assert _rel(bp, bq) <= 1e-3, name


def test_soft_argmax_training_pair_matches_the_torch_graph(dev):
    """sp3d_soft_argmax_grid_train / _bwd (one workgroup per row forward keeping max and sum; one elementwise pass backward)
    against SoftArgmaxLayer's torch graph (lib/models/pose_regression_net.py:19-28: softmax(beta x) . grid) in float64, on
    planar and channels-last inputs; the voxel centres come from the unprojection kernel's own grids"""
    from selfpose3d_amd import _lib, synthetic as syn
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.project_layer import ProjectLayer
    P, J, cube, gs, beta = 3, 15, (16, 12, 10), (2000.0, 1800.0, 1500.0), 100.0
    cfg = load_config(None, NETWORK__IMAGE_SIZE=[128, 96], NETWORK__HEATMAP_SIZE=[32, 24])
    centers = torch.tensor([[100.0, -300.0, 900.0], [-700.0, 250.0, 1100.0], [0.0, 0.0, 800.0]], device=dev)
    meta = syn.make_meta(1, 2, (128, 96))
    hms = [h.to(dev) for h in syn.random_heatmaps(1, 2, J, 24, 32, seed=4)]
    _, grids = ProjectLayer(cfg).get_voxel(hms, meta, list(gs), centers, list(cube), sample_of=torch.zeros(P, dtype=torch.int32))
    gen = torch.Generator().manual_seed(12)
    x0 = (torch.randn(P, J, *cube, generator=gen) * 0.03).to(dev)            # beta x of a few units: a soft maximum
    wgt = torch.randn(P, J, 3, generator=gen).to(dev)
    xd = x0.double().requires_grad_(True)
    p = torch.softmax(beta * xd.reshape(P, J, -1, 1), dim=2)
    ref = (p * grids.double().unsqueeze(1)).sum(dim=2)
    (ref * wgt.double()).sum().backward()
    for fmt in (torch.contiguous_format, torch.channels_last_3d):
        x = x0.clone().contiguous(memory_format=fmt).requires_grad_(True)
        out = _lib.soft_argmax_grid_autograd(x, centers, gs, cube, beta)
        (out * wgt).sum().backward()
        assert float((out.double() - ref).abs().max()) <= 2e-3                # mm, on a 2 m cube
        assert _rel(x.grad, xd.grad) <= 2e-4
        assert x.grad.shape == x.shape


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("dims,spatial,C,group_of", [(3, (8, 6, 5), 32, [0, 0, 1, 1, 2]), (2, (10, 12), 256, [0, 1, 2, 0, 1, 2])])
def test_residual_block_tail_in_one_pass(dev, dims, spatial, C, group_of, dtype):
    """mode SP3D_GBN_ADD_RELU: y = relu(BatchNorm(x) + residual) - the tail of Res3DBlock (lib/models/v2v_net.py:42-45) and
    of the ResNet blocks - against per-group nn.BatchNorm + add + relu in float64: output, gradients of x, of the residual
    (the masked upstream gradient), of weight and bias, running statistics"""
    from selfpose3d_amd.grouped_bn import GroupedBatchNorm2d, GroupedBatchNorm3d, GroupSpec
    G = max(group_of) + 1
    sizes = [group_of.count(g) for g in range(G)]
    gen = torch.Generator().manual_seed(77)
    N = len(group_of)
    fmt = torch.channels_last_3d if dims == 3 else torch.channels_last
    x = (torch.randn((N, C) + spatial, generator=gen) * 2 + 1).to(dev)
    r = torch.randn((N, C) + spatial, generator=gen).to(dev)
    w, b = (torch.rand(C, generator=gen) + 0.5).to(dev), torch.randn(C, generator=gen).to(dev)
    gy = torch.randn((N, C) + spatial, generator=gen).to(dev)
    ref_bn = (nn.BatchNorm3d if dims == 3 else nn.BatchNorm2d)(C).to(dev).double().train()
    with torch.no_grad():
        ref_bn.weight.copy_(w); ref_bn.bias.copy_(b)
    xd, rd = x.double().requires_grad_(True), r.double().requires_grad_(True)
    go = torch.as_tensor(group_of, device=dev)
    yref = torch.zeros_like(xd)
    for g in range(G):
        idx = torch.nonzero(go == g).flatten()
        yref = yref.index_put((idx,), torch.relu(ref_bn(xd[idx]) + rd[idx]))
    (yref * gy.double()).sum().backward()
    bn = (GroupedBatchNorm3d if dims == 3 else GroupedBatchNorm2d)(C).to(dev).to(dtype).train()
    with torch.no_grad():
        bn.weight.copy_(w); bn.bias.copy_(b)
    bn.groups = GroupSpec(sizes, dev, group_of=group_of)
    xi = x.to(dtype).contiguous(memory_format=fmt).requires_grad_(True)
    ri = r.to(dtype).contiguous(memory_format=fmt).requires_grad_(True)
    y = bn.grouped_forward(xi, relu=True, residual=ri)
    (y * gy.to(dtype)).sum().backward()
    ty, tg, ts = (1e-9, 1e-7, 1e-12) if dtype == torch.float64 else (1e-5, 5e-5, 1e-5)
    assert _rel(y, yref) <= ty
    assert _rel(xi.grad, xd.grad) <= tg and _rel(ri.grad, rd.grad) <= tg
    assert _rel(bn.weight.grad, ref_bn.weight.grad) <= tg and _rel(bn.bias.grad, ref_bn.bias.grad) <= tg
    assert _rel(bn.running_mean, ref_bn.running_mean) <= ts and _rel(bn.running_var, ref_bn.running_var) <= ts
    # without a spec / in eval mode the same call is the plain modules
    bn.groups = None
    plain = bn.grouped_forward(xi.detach(), relu=True, residual=ri.detach())
    want = torch.relu(nn.functional.batch_norm(xi.detach(), None, None, bn.weight, bn.bias, True, 0.0, bn.eps) + ri.detach())
    assert _rel(plain, want) <= 1e-5
