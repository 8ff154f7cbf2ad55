#!/usr/bin/env python3
"""Round-4 golden vectors, by RUNNING THE REFERENCE'S OWN PYTHON on CPU (build container only; import shims of
make_goldens.py / make_goldens_r2.py):

  posenet_full.npz     reference PoseRegressionNet.forward (lib/models/pose_regression_net.py:41-53: ProjectLayer on the
                       fine grid -> V2VNet lib/models/v2v_net.py:113-144 -> SoftArgmaxLayer :19-28) at the size the pose stage
                       is benchmarked and trained at: 5 views, 240x128 heat-maps, J = 15, 64^3 cubes of 2000 mm, B = 2, two
                       candidate slots = 3 valid proposals + 1 invalid one (flag < 0, sample 1 of slot 1).  Stored: the
                       predictions, and of every valid cube the V2V output as a strided sub-sample + float64 sums + range
                       (the soft-argmax input), and of the unprojected cube a strided sub-sample + sums.

Inputs are this repo's synthetic people scene (selfpose3d_amd/synthetic.py), so the tests rebuild them exactly; the
proposal centres sit near (not on) the people's roots, as the root net's proposals do.

    python tests/golden/make_goldens_r4.py [name ...]
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")

import make_goldens as mg        # noqa: E402
import make_goldens_r2 as mg2    # noqa: E402
from selfpose3d_amd import synthetic as syn   # noqa: E402

import golden_io as gio          # noqa: E402
POSENET_FULL, posenet_full_inputs = gio.POSENET_FULL, gio.posenet_full_inputs


def g_posenet_full():
    from models.pose_regression_net import PoseRegressionNet
    c = POSENET_FULL
    cfg = mg.make_cfg(c["img"], c["hm"], syn.SPACE_SIZE, syn.SPACE_CENTER, syn.INITIAL_CUBE_SIZE, syn.FINE_GRID_SIZE,
                      c["fine_cube"], c["J"])
    net = PoseRegressionNet(cfg)
    syn.fill_parameters_deterministic(net, seed=c["pose_seed"], scale=c["param_scale"])
    net.eval()
    hms, meta, gc = posenet_full_inputs()
    grabbed = {}
    net.v2v_net.register_forward_hook(lambda m, i, o: grabbed.update(x=i[0].detach().clone(), y=o.detach().clone()))
    N = int(np.prod(c["fine_cube"]))
    sub = np.arange(0, N, c["stride"])
    rec = dict(img=np.array(c["img"]), hm=np.array(c["hm"]), V=c["V"], J=c["J"], B=c["B"], fine_cube=np.array(c["fine_cube"]),
               hm_seed=c["hm_seed"], pose_seed=c["pose_seed"], param_scale=c["param_scale"], grid_centers=gc.numpy(),
               sub_idx=sub, hm_sum=np.array([float(h.double().sum()) for h in hms]),
               pose_keys=np.array(sorted(net.state_dict().keys())))
    preds = []
    with torch.no_grad():
        for k in range(gc.shape[1]):
            pred = net(hms, meta, gc[:, k])
            preds.append(pred.numpy())
            x, y = grabbed["x"].numpy(), grabbed["y"].numpy()     # rows = the valid samples of this slot, in order
            nv = y.shape[0]
            rec[f"cube_sub_{k}"] = x.reshape(nv, c["J"], N)[:, :, sub]
            rec[f"cube_sum_{k}"] = x.astype(np.float64).sum(axis=(2, 3, 4))
            rec[f"v2v_sub_{k}"] = y.reshape(nv, c["J"], N)[:, :, sub]
            rec[f"v2v_sum_{k}"] = y.astype(np.float64).sum(axis=(2, 3, 4))
            rec[f"v2v_abs_sum_{k}"] = np.abs(y.astype(np.float64)).sum(axis=(2, 3, 4))
            rec[f"v2v_min_{k}"] = y.reshape(nv, c["J"], N).min(axis=2)
            rec[f"v2v_max_{k}"] = y.reshape(nv, c["J"], N).max(axis=2)
            print(f"posenet_full slot {k}: valid {nv}, cube range {float(x.min()):.4f}..{float(x.max()):.4f}, "
                  f"v2v range {float(y.min()):.4f}..{float(y.max()):.4f}, pred[0,:2] {pred[0, :2].numpy()}")
    rec["preds"] = np.stack(preds)                                # (K, B, J, 3)
    np.savez_compressed(os.path.join(HERE, "posenet_full.npz"), **rec)
    print("posenet_full: bytes", os.path.getsize(os.path.join(HERE, "posenet_full.npz")))


ALL = {"posenet_full": g_posenet_full}

if __name__ == "__main__":
    mg2.install_shims()
    torch.nn.Module.cuda = lambda self, device=None: self
    torch.set_num_threads(8)
    for n in (sys.argv[1:] or list(ALL)):
        ALL[n]()
