import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tests import golden_io as gio
from selfpose3d_amd.multi_person_posenet import get_multi_person_pose_net
dev = torch.device('cuda:0')
g = gio.load("train_step")
def rel(a,b): return float(np.abs(a-b).max()/max(1e-30,np.abs(b).max()))
for kw in ({}, {"freq": False}, {"cl": True}):
    cfg = gio.train_cfg(USE_GT=False)
    model = get_multi_person_pose_net(cfg, is_train=True)
    gio.he_fill(model, seed=int(g["param_seed"]))
    model.to(dev).train()
    if kw.get("cl"): model.use_channels_last(True)
    inputs, t2d, w2d, t3d, meta, _ = gio.train_batch(cfg, B=2, seed=int(g["data_seed"]))
    inputs = [x.to(dev) for x in inputs]
    pred, hms, gc, l2d, l3d, lcord = model(views=inputs, meta=meta, targets_2d=t2d, weights_2d=w2d, targets_3d=t3d[0])
    fl = model.backbone.final_layer.weight
    ol = model.root_net.v2v_net.output_layer.weight
    fc = model.root_net.v2v_net.front_layers[0].block[0].weight
    g3, ga, gb = torch.autograd.grad(l3d.mean(), (fl, ol, fc), retain_graph=True)
    print(kw, "loss3d", float(l3d), float(g["net_loss_3d"]), "final_3d", rel(g3.cpu().numpy(), g["net_grad_final_3d"]),
          "root_out", rel(ga.cpu().numpy(), g["net_grad_root_out"]), "root_front", rel(gb.cpu().numpy(), g["net_grad_root_front"]))
    # gradient wrt heat-maps directly
    gh = torch.autograd.grad(l3d.mean(), hms, retain_graph=True)
    print("   |grad hm| max", [float(x.abs().max()) for x in gh])
