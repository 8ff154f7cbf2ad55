"""HIP-graphed per-view backbone passes in training (selfpose3d_amd/graphs.py graph_backbone_views): the same kernels as the
eager passes, so the losses, the gradients, the BatchNorm statistics and the weights after optimizer steps must follow the
eager model's - over several iterations (a graph replays with the CURRENT weights) and when evaluation runs in between."""
import numpy as np
import pytest
import torch

from tests import golden_io as gio

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_graphed_backbone_views_follow_the_eager_model(dev, monkeypatch):
    from selfpose3d_amd.graphs import graph_backbone_views
    from selfpose3d_amd.multi_person_posenet import get_multi_person_pose_net
    monkeypatch.setattr(torch.backends.cudnn, "benchmark", False)         # the same MIOpen kernels in both models
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)
    cfg = gio.train_cfg(USE_GT=True)
    inputs, t2d, w2d, t3d, meta, _ = gio.train_batch(cfg, B=2, seed=5)
    inputs = [x.to(dev) for x in inputs]
    models, opts = [], []
    for _ in range(2):
        m = get_multi_person_pose_net(cfg, is_train=True)
        gio.he_fill(m, seed=77)
        m.to(dev).train()
        models.append(m)
        opts.append(torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=1e-3))
    eager, graphed = models
    graph_backbone_views(graphed.backbone, inputs)
    assert sorted(eager.state_dict().keys()) == sorted(graphed.state_dict().keys())       # nothing registered on the module
    for it in range(3):
        outs = []
        for m, opt in zip(models, opts):
            opt.zero_grad(set_to_none=True)
            _, hms, _, l2d, l3d, lcord = m(views=inputs, meta=meta, targets_2d=t2d, weights_2d=w2d, targets_3d=t3d[0])
            loss = l2d.mean() + l3d.mean() + lcord.mean()
            loss.backward()
            outs.append((float(loss), [h.detach().clone() for h in hms],
                         {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
            opt.step()
        (le, he, ge), (lg, hg, gg) = outs
        assert abs(le - lg) <= 1e-6 * max(1.0, abs(le)), (it, le, lg)
        for a, b in zip(hg, he):
            assert _rel(a, b) <= 1e-6
        assert set(ge) == set(gg)
        worst = max(_rel(gg[k], ge[k]) for k in ge if float(ge[k].abs().max()) > 0)
        assert worst <= 1e-5, (it, worst)
        if it == 1:                                # an evaluation pass in between must not disturb the captured graphs
            for m in models:
                m.eval()
                with torch.no_grad():
                    m(views=inputs, meta=meta)
                m.train()
    for (k, a), (_, b) in zip(graphed.backbone.state_dict().items(), eager.backbone.state_dict().items()):
        if a.is_floating_point():
            assert _rel(a.float(), b.float()) <= 1e-5, k      # weights and BatchNorm running statistics after 3 steps
        else:
            assert torch.equal(a, b), k                       # num_batches_tracked
