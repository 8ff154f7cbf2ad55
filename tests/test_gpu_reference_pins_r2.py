"""Round-2 GPU pins against outputs of the reference's own Python (tests/golden/make_goldens_r2.py):
  * the BENCHMARKED root-net pipeline (FFT opening conv + fused/low-res Winograd + GEMM up-convs + HIP NMS, channels-last,
    as one HIP graph - exactly bench.py:build_workload) at its own size, 80x80x20 / J=15 / 240x128, against the reference
    CuboidProposalNet -> V2VNet -> nms (cuboid_proposal_net.py:102-122, v2v_net.py:128-133, proposal.py:35-48);
  * the batched pose path (PoseRegressionNet.forward_batched, MultiPersonPoseNet.forward in eval) against the
    reference's per-candidate outputs (multi_person_posenet.py:84-88);
  * MultiPersonPoseNetSSV.do_inference against the reference's (multi_person_posenet_ssv.py:105-153).
"""
import numpy as np
import pytest
import torch

from tests import golden_io as gio

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _rootnet_full_inputs(g):
    from selfpose3d_amd import synthetic as syn
    img, hm, V, J = [int(v) for v in g["img"]], [int(v) for v in g["hm"]], int(g["V"]), int(g["J"])
    seed = int(g["hm_seed"])
    rnd = syn.random_heatmaps(2, V, J, hm[1], hm[0], seed=seed)
    ppl, _ = syn.people_heatmaps(2, V, J, hm[1], hm[0], img, seed=seed + 1)
    hms = [torch.stack([0.35 * rnd[v][0], ppl[v][1]]) for v in range(V)]
    sums = np.array([float(h.double().sum()) for h in hms])
    assert np.allclose(sums, g["hm_sum"], rtol=0, atol=1e-6 * float(np.abs(sums).max()))
    return img, hm, V, J, hms, syn.make_meta(2, V, img)


def _check_root(root_cubes, grid_centers, g, tol=5e-5, cube=None, min_checked=10):
    # 5e-5 x the output range (+-5.13) = 2.6e-4 absolute: 20x the measured error (1.3e-5); a conv kernel that lost a
    # decimal digit would fail here (round-2 review: the old 2e-4 would have let a 75x regression through)
    from selfpose3d_amd import synthetic as syn
    rc = root_cubes.float().cpu().numpy()
    N = rc[0].size
    ref = g["root_sub"]
    scale = float(np.abs(ref).max())                       # outputs reach +-5: tolerance relative to that
    d = np.abs(rc.reshape(2, N)[:, g["sub_idx"]] - ref).max()
    assert d <= tol * max(1.0, scale), float(d)
    assert np.abs(rc.astype(np.float64).sum(axis=(1, 2, 3)) - g["root_sum"]).max() <= 1e-6 * g["root_abs_sum"].max()
    # proposals: indices bit-exact wherever the reference's score is separated from its neighbours by more than the
    # conv rounding (MIOpen/rocFFT/Winograd vs oneDNN), scores within tol
    vals, idx = g["nms_vals"], g["nms_idx"]
    gc = grid_centers.float().cpu()
    cs = torch.tensor(syn.INITIAL_CUBE_SIZE if cube is None else cube, dtype=torch.float32)
    gs, cen = torch.tensor(syn.SPACE_SIZE), torch.tensor(syn.SPACE_CENTER)
    checked = 0
    for b in range(2):
        for k in range(vals.shape[1]):
            gap = min(abs(float(vals[b, k] - vals[b, k - 1])) if k else 9.0,
                      abs(float(vals[b, k] - vals[b, k + 1])) if k + 1 < vals.shape[1] else 9.0)
            if gap > 4 * tol * scale:
                loc = torch.from_numpy(idx[b, k]).float() / (cs - 1) * gs + cen - gs / 2.0
                assert torch.equal(gc[b, k, :3], loc), (b, k)
                assert abs(float(gc[b, k, 4]) - float(vals[b, k])) <= tol * scale
                checked += 1
    assert checked >= min_checked, checked
    ref_gc = torch.from_numpy(g["grid_centers"])
    assert torch.equal((gc[:, :, 3] >= 0), (ref_gc[:, :, 3] >= 0))


@pytest.mark.parametrize("mode", ["bench", "eager_cl", "eager_plain"])
def test_rootnet_full_size_vs_reference(dev, mode):
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.cuboid_proposal_net import CuboidProposalNet
    g = gio.load("rootnet_full")
    img, hm, V, J, hms, meta = _rootnet_full_inputs(g)
    cfg = load_config(None)
    assert list(cfg.NETWORK.IMAGE_SIZE) == img and list(cfg.NETWORK.HEATMAP_SIZE) == hm
    net = CuboidProposalNet(cfg)
    syn.fill_parameters_deterministic(net, seed=int(g["param_seed"]), scale=float(g["param_scale"]))
    net.eval().to(dev)
    hms = [h.to(dev) for h in hms]
    if mode == "eager_plain":                               # plain module path: MIOpen convs, NCDHW
        net.v2v_net.fft_front = False
        net.v2v_net.winograd = False
    else:                                                   # bench.py:build_workload
        net.use_channels_last(True)
        net.v2v_net.fft_front = True
        net.v2v_net.winograd = True
    with torch.no_grad():
        root_cubes, grid_centers = net(hms, meta)
        if mode == "bench":
            from selfpose3d_amd.graphs import GraphedRootNet
            torch.cuda.synchronize()
            graphed = GraphedRootNet(net, hms, meta)
            root_cubes, grid_centers = graphed()
            torch.cuda.synchronize()
    _check_root(root_cubes, grid_centers, g)


def _small_models(g, dev):
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.config import load_config
    B, V, J = int(g["B"]), int(g["V"]), int(g["J"])
    img, hm = [int(v) for v in g["img"]], [int(v) for v in g["hm"]]
    cfg = load_config(None, NETWORK__IMAGE_SIZE=img, NETWORK__HEATMAP_SIZE=hm, NETWORK__NUM_JOINTS=J,
                      MULTI_PERSON__INITIAL_CUBE_SIZE=[int(v) for v in g["cube"]],
                      PICT_STRUCT__CUBE_SIZE=[int(v) for v in g["fine_cube"]],
                      MULTI_PERSON__THRESHOLD=float(g["threshold"]), BACKBONE_MODEL="")
    hms, _ = syn.people_heatmaps(B, V, J, hm[1], hm[0], img, seed=int(g["hm_seed"]))
    return cfg, [h.to(dev) for h in hms], B, V, J, img


@pytest.mark.parametrize("cl", [False, True])
def test_batched_posenet_vs_reference_per_candidate_outputs(dev, cl):
    """f1: forward_batched (one indexed launch + chunked V2V + fused soft-argmax) == the reference's loop outputs"""
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.pose_regression_net import PoseRegressionNet
    g = gio.load("rootnet_posenet")
    cfg, hms, B, V, J, img = _small_models(g, dev)
    meta = syn.make_meta(B, V, img)
    net = PoseRegressionNet(cfg)
    syn.fill_parameters_deterministic(net, seed=int(g["pose_seed"]), scale=float(g["param_scale"]))
    net.eval().to(dev)
    if cl:
        net.use_channels_last(True)
    gc_ref = torch.from_numpy(g["grid_centers"]).to(dev)
    n = g["preds"].shape[0]
    gc = gc_ref.clone()
    gc[:, n:, 3] = -1.0                                   # the golden holds the first n candidates
    pred = net.forward_batched(hms, meta, gc, max_cubes_per_call=4)
    ref = torch.from_numpy(g["preds"]).permute(1, 0, 2, 3)            # (B, n, J, 3)
    valid = (gc_ref[:, :n, 3] >= 0).cpu()
    assert bool(valid.any())
    d = (pred[:, :n].cpu() - ref).abs().amax(dim=(2, 3))
    assert float(d[valid].max()) <= 0.5, float(d[valid].max())      # mm on +-2000 mm coordinates
    assert torch.count_nonzero(pred[:, n:]) == 0
    assert torch.count_nonzero(pred[:, :n].cpu()[~valid]) == 0


def test_multi_person_posenet_eval_vs_reference(dev):
    """the model the CLI runs (MultiPersonPoseNet.forward, eval: root net -> forward_batched) against the reference's
    root net + per-candidate pose net outputs"""
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.multi_person_posenet import MultiPersonPoseNet
    g = gio.load("rootnet_posenet")
    cfg, hms, B, V, J, img = _small_models(g, dev)
    meta = syn.make_meta(B, V, img)
    model = MultiPersonPoseNet(None, cfg)
    syn.fill_parameters_deterministic(model.root_net, seed=int(g["root_seed"]), scale=float(g["param_scale"]))
    syn.fill_parameters_deterministic(model.pose_net, seed=int(g["pose_seed"]), scale=float(g["param_scale"]))
    model.eval().to(dev)
    with torch.no_grad():
        pred, _, grid_centers, _, _, _ = model(views=None, meta=meta, input_heatmaps=hms)
    gc_ref = torch.from_numpy(g["grid_centers"])
    ref = torch.from_numpy(g["preds"]).permute(1, 0, 2, 3)
    n = ref.shape[1]
    sc = gc_ref[:, :, 4]
    checked = 0
    for b in range(B):
        for k in range(n):
            gap = min(abs(float(sc[b, k] - sc[b, k - 1])) if k else 1.0, abs(float(sc[b, k] - sc[b, k + 1])))
            if gap > 1e-3 and float(gc_ref[b, k, 3]) >= 0:       # same proposal in this slot as in the reference
                assert torch.equal(grid_centers[b, k, :4].cpu(), gc_ref[b, k, :4])
                assert float((pred[b, k, :, :3].cpu() - ref[b, k]).abs().max()) <= 0.5
                assert torch.equal(pred[b, k, :, 3].cpu(), gc_ref[b, k, 3].expand(J))
                checked += 1
    assert checked >= 3


def test_ssv_do_inference_vs_reference(dev):
    """MultiPersonPoseNetSSV.forward(inference=True) == the reference's do_inference (soft root net on the root channel
    + pose net on every valid proposal)"""
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.models import get_multi_person_pose_net
    g = gio.load("ssv_inference")
    B, V, J = int(g["B"]), int(g["V"]), int(g["J"])
    img, hm = [int(v) for v in g["img"]], [int(v) for v in g["hm"]]
    cfg = load_config(None, MODEL="multi_person_posenet_ssv", BACKBONE_MODEL="", NETWORK__IMAGE_SIZE=img,
                      NETWORK__HEATMAP_SIZE=hm, NETWORK__NUM_JOINTS=J, NETWORK__ROOTNET_ROOTHM=True,
                      NETWORK__ROOTNET_TRAIN_SYNTH=True, MULTI_PERSON__INITIAL_CUBE_SIZE=[int(v) for v in g["cube"]],
                      PICT_STRUCT__CUBE_SIZE=[int(v) for v in g["fine_cube"]], MULTI_PERSON__THRESHOLD=float(g["threshold"]))
    model = get_multi_person_pose_net(cfg, is_train=False)
    assert sorted(model.state_dict().keys()) == [str(k) for k in g["keys"]]
    syn.fill_parameters_deterministic(model, seed=int(g["param_seed"]), scale=float(g["param_scale"]))
    model.eval().to(dev)
    meta = syn.make_meta(B, V, img, ssv_style=True)
    hms, _ = syn.people_heatmaps(B, V, J, hm[1], hm[0], img, seed=int(g["hm_seed"]))
    pred, hm_out, grid_centers = model(views1=None, meta1=meta, input_heatmaps1=[h.to(dev) for h in hms], inference=True)
    ref_p, ref_gc = torch.from_numpy(g["pred"]), torch.from_numpy(g["grid_centers"])
    sc = ref_gc[:, :, 4]
    checked = 0
    for b in range(B):
        for k in range(sc.shape[1]):
            gap = min(abs(float(sc[b, k] - sc[b, k - 1])) if k else 1.0,
                      abs(float(sc[b, k] - sc[b, k + 1])) if k + 1 < sc.shape[1] else 1.0)
            if gap > 1e-3:
                assert torch.equal(grid_centers[b, k, :4].cpu(), ref_gc[b, k, :4]), (b, k)
                assert abs(float(grid_centers[b, k, 4].cpu() - ref_gc[b, k, 4])) <= 2e-4
                assert float((pred[b, k, :, :3].cpu() - ref_p[b, k, :, :3]).abs().max()) <= 0.5
                assert torch.equal(pred[b, k, :, 3:].cpu()[:, 0], ref_p[b, k, :, 3])
                checked += 1
    assert checked >= 4, checked


def test_soft_argmax_kernel_vs_reference_layer(dev):
    """sp3d_soft_argmax against the reference SoftArgmaxLayer's own output (pose_regression_net.py:19-28), not only
    against the oracle: sharp peak, flat channel, noisy channels; coordinates up to ~2 m"""
    from selfpose3d_amd import _lib
    g = gio.load("softargmax")
    x = torch.from_numpy(g["x"]).to(dev)
    grids = torch.from_numpy(g["grids"]).to(dev)
    out = _lib.soft_argmax(x, grids, float(g["beta"])).cpu().numpy()
    err = float(np.abs(out - g["out"]).max())
    assert err <= 2e-3, err                     # mm (measured 8.5e-4)
    print("soft-argmax vs reference layer: max |d| = %.2e mm" % err)
