"""Round-2 pins against outputs of the reference's own Python (tests/golden/make_goldens_r2.py), CPU side:
the soft-argmax oracle, the 2D backbones, and the checkpoint format (state_dict keys and shapes)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from selfpose3d_amd.config import load_config
from tests import golden_io as gio

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_soft_argmax_vs_reference_layer():
    """oracle.soft_argmax == reference SoftArgmaxLayer (pose_regression_net.py:19-28) on a sharp, a flat and noisy channels"""
    g = gio.load("softargmax")
    got = oracle.soft_argmax(g["x"], g["grids"], float(g["beta"]))
    ref = g["out"]
    # mm; fp32 softmax over 512 cells of coordinates up to 2 m: 1e-3 mm is ~1 ulp of the sum
    assert np.abs(got - ref).max() <= 2e-3, float(np.abs(got - ref).max())


def _he_fill(net, seed):
    """the deterministic weights make_goldens_r2.g_pose_resnet gave the reference nets"""
    rng = np.random.default_rng(seed)
    sd = net.state_dict()
    with torch.no_grad():
        for k in sorted(sd):
            t = sd[k]
            if not torch.is_floating_point(t):
                continue
            if t.dim() == 4:
                fan = t.shape[1] * t.shape[2] * t.shape[3]
                a = rng.standard_normal(tuple(t.shape)).astype(np.float32) * np.sqrt(2.0 / fan)
            elif k.endswith("running_var"):
                a = 1.0 + 0.1 * np.abs(rng.standard_normal(tuple(t.shape))).astype(np.float32)
            elif k.endswith("weight"):
                a = 1.0 + 0.05 * rng.standard_normal(tuple(t.shape)).astype(np.float32)
            else:
                a = 0.05 * rng.standard_normal(tuple(t.shape)).astype(np.float32)
            t.copy_(torch.from_numpy(a))
    return net


@pytest.mark.parametrize("which", ["r50", "attn18"])
def test_backbone_vs_reference_pose_resnet(which):
    """same keys, same output as reference get_pose_net / get_pose_attn_net (pose_resnet.py:274-333) on a 64x64 input"""
    from selfpose3d_amd import pose_resnet
    g = gio.load("pose_resnet")
    cfg = load_config(None)
    net = pose_resnet.get_pose_net(cfg, is_train=False) if which == "r50" else \
        pose_resnet.get_pose_attn_net(cfg, is_train=False)
    assert sorted(net.state_dict().keys()) == [str(k) for k in g[which + "_keys"]]
    _he_fill(net, int(g["w_seed"])).eval()
    x = torch.from_numpy(np.random.default_rng(int(g["x_seed"])).standard_normal((1, 3, 64, 64)).astype(np.float32))
    with torch.no_grad():
        y = net(x).numpy()
    ref = g[which + "_out"]
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max()))
    if which == "r50":              # forward_views (all views as one batch) == per-view calls
        with torch.no_grad():
            ys = net.forward_views([x, 0.5 * x])
        assert np.abs(ys[0].numpy() - ref).max() <= 2e-5 * float(np.abs(ref).max())


def _ref_keys():
    with open(os.path.join(HERE, "golden", "state_dict_keys.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["multi_person_posenet", "multi_person_posenet_ssv", "multi_person_posenet_ssv_attn"])
def test_state_dict_is_the_reference_checkpoint_format(name):
    """keys AND shapes of the top-level models == the reference's, and a reference-shaped checkpoint loads strict"""
    from selfpose3d_amd.models import get_multi_person_pose_net
    ref = _ref_keys()[name]
    over = {}
    if name != "multi_person_posenet":
        over = dict(MODEL="multi_person_posenet_ssv", NETWORK__ROOTNET_ROOTHM=True, NETWORK__ROOTNET_TRAIN_SYNTH=True,
                    WITH_ATTN=name.endswith("attn"))
    cfg = load_config(None, **over)
    model = get_multi_person_pose_net(cfg, is_train=False)
    own = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert sorted(own) == sorted(ref)
    assert own == ref
    assert len(ref) == (790 if name.endswith("attn") else 650)
    # a checkpoint as the reference writes it (utils.py:109-115: plain state_dict) loads with strict=True
    ck = {k: torch.full(shape, 0.25) if shape else torch.tensor(3) for k, shape in ref.items()}
    for k in ck:
        if k.endswith("num_batches_tracked"):
            ck[k] = torch.tensor(3, dtype=torch.long)
    missing = model.load_state_dict(ck, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert float(model.state_dict()["pose_net.v2v_net.output_layer.weight"].flatten()[0]) == 0.25


def test_unknown_model_refuses_loudly_and_ssv_training_has_no_cpu_path():
    from selfpose3d_amd import _lib
    from selfpose3d_amd.models import get_multi_person_pose_net
    with pytest.raises(ValueError):
        get_multi_person_pose_net(load_config(None, MODEL="multi_person_posenet_xyz"), is_train=False)
    # round 3: the SSV training forward is built (tests/test_gpu_reference_pins_r3.py pins it to the reference); like
    # every path through ProjectLayer it has no CPU fallback and says so
    from tests import golden_io as gio
    cfg = gio.train_cfg(ssv=True)
    cfg.BACKBONE_MODEL, cfg.WITH_ATTN = "", False
    model = get_multi_person_pose_net(cfg, is_train=True)
    b = gio.train_batch(cfg, B=1, seed=3, ssv=True)
    hm = [torch.zeros(1, 15, 24, 32) for _ in range(3)]
    with pytest.raises(_lib.Sp3dError):
        model(views1=None, meta1=b[4], input_heatmaps1=hm, views2=None, meta2=b[10], input_heatmaps2=hm, views3=None,
              meta3=b[16], input_heatmaps3=hm)


@pytest.mark.parametrize("yaml_name", ["cam5_rootnet.yaml", "cam5_posenet.yaml", "cam5_posenet_finetune.yaml"])
def test_ssv_yamls_dispatch_to_the_ssv_model(yaml_name):
    """the reference's SSL YAMLs (MODEL: multi_person_posenet_ssv) build the SSV model, never the supervised one"""
    path = os.path.join("/root/reference/configs/panoptic_ssl/resnet50", yaml_name)
    if not os.path.exists(path):
        pytest.skip("reference YAMLs are only in the build container")
    from selfpose3d_amd.models import get_multi_person_pose_net, is_ssv
    from selfpose3d_amd.multi_person_posenet_ssv import MultiPersonPoseNetSSV
    cfg = load_config(path)
    assert is_ssv(cfg)
    cfg.BACKBONE_MODEL = ""            # skip building two ResNets here (covered above)
    cfg.WITH_ATTN = False
    model = get_multi_person_pose_net(cfg, is_train=False)
    assert isinstance(model, MultiPersonPoseNetSSV)
    assert hasattr(model, "root_net") and model.root_net.rootnet_roothm == bool(cfg.NETWORK.ROOTNET_ROOTHM)
