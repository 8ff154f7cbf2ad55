"""world_size-2 CPU (gloo) tests of the N>1 path: frame sharding, max-over-ranks timing, prediction
gather, and DDP gradient averaging == the reference's DataParallel semantics (mean of replica losses)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from selfpose3d_amd import distributed as D
    from selfpose3d_amd.v2v_net import V2VNet
    r, w = D.init("gloo")
    assert (r, w) == (rank, world)
    # 1. sharding covers every frame exactly once
    frames = D.shard_frames(7, rank, world)
    allf = [None] * world
    dist.all_gather_object(allf, frames)
    assert sorted(sum(allf, [])) == list(range(7))
    # 2. slowest rank defines the job time
    assert D.max_over_ranks(1.0 + rank) == float(world)
    # 3. predictions come back in frame order, also from UNEVEN shards (7 frames on 2 ranks: 4 + 3, no padding)
    pred = torch.tensor([[float(f)] for f in D.shard_frames(6, rank, world)])
    assert D.gather_predictions(pred).flatten().tolist() == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]
    pred7 = torch.tensor([[float(f), 10.0 * f] for f in D.shard_frames(7, rank, world)])
    got7 = D.gather_predictions(pred7)
    assert got7.shape == (7, 2) and got7[:, 0].tolist() == [float(f) for f in range(7)]
    assert got7[:, 1].tolist() == [10.0 * f for f in range(7)]
    # 3b. evaluation counters are summed over ranks; the bench timing rule runs exactly `steps` timed steps per rank,
    #     reports the slowest rank, and the whole-job rate counts every rank's units
    assert D.sum_over_ranks(1 + rank, 10) == (3.0, 20.0)
    calls = []

    def stub_step():
        calls.append(1)
        import time as _t
        _t.sleep(0.01 * (1 + rank))
        return len(calls)
    secs, last = D.timed_steps(stub_step, steps=3, warmup=2)
    assert len(calls) == 5 and last == 5
    assert 0.06 <= secs < 0.5                          # rank 1 sleeps 20 ms per step: both ranks report >= 60 ms
    assert D.job_throughput(4, 3, secs, world) == world * 4 * 3 / secs
    # 3c. per-rank random streams: the synthetic-root sampler draws different roots on every rank
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.cuboid_proposal_net_soft import CuboidProposalNetSoft
    cfg = load_config(None, NETWORK__ROOTNET_ROOTHM=True, NETWORK__ROOTNET_TRAIN_SYNTH=True,
                      MULTI_PERSON__INITIAL_CUBE_SIZE=[8, 8, 4])
    soft = CuboidProposalNetSoft(cfg).seed_sampler(123)
    roots = soft.sample_roots(2, torch.device("cpu"), soft.generator)
    allr = [None] * world
    dist.all_gather_object(allr, roots.flatten()[:6].tolist())
    assert allr[0] != allr[1]
    again = CuboidProposalNetSoft(cfg).seed_sampler(123)
    assert torch.equal(again.sample_roots(2, torch.device("cpu"), again.generator), roots)      # reproducible per rank
    # 4. DDP grads == grads of the mean loss over the global batch (identical replicas)
    torch.manual_seed(0)
    net = V2VNet(2, 1)
    for m in net.modules():
        if isinstance(m, torch.nn.Conv3d):
            torch.nn.init.normal_(m.weight, 0, 0.05)
    net.eval()                      # BN in eval: replicas are exactly comparable to one global pass
    x = torch.randn(4, 2, 8, 8, 4, generator=torch.Generator().manual_seed(1))
    ddp = D.wrap_ddp(net, find_unused=False)
    loss = ddp(D.shard_batch([x], rank, world)[0]).pow(2).mean()
    loss.backward()
    g_ddp = net.output_layer.weight.grad.clone()
    ref = V2VNet(2, 1)
    ref.load_state_dict(net.state_dict())
    ref.eval()
    ref(x).pow(2).mean().backward()
    err = float((g_ddp - ref.output_layer.weight.grad).abs().max())
    q.put((rank, err))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_path():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    errs = dict(q.get(timeout=10) for _ in range(world))
    assert set(errs) == {0, 1} and max(errs.values()) < 1e-6


def test_no_stage_needs_find_unused_parameters():
    """every trainable parameter gets a gradient on every rank in every iteration by construction (frozen sub-nets do
    not require grad; skipped sub-nets are zero-anchored inside forward), so DDP's static fast path applies"""
    from selfpose3d_amd import distributed as D
    from selfpose3d_amd.config import load_config
    for kw in ({}, {"NETWORK__TRAIN_ONLY_2D": True}, {"NETWORK__TRAIN_ONLY_ROOTNET": True}, {"NETWORK__USE_GT": True}):
        assert D.needs_find_unused(load_config(None, **kw)) is False
    assert D.rank_seed(5, 0) != D.rank_seed(5, 1) and D.rank_seed(5, 1) == D.rank_seed(5, 1)


def test_use_gt_freezes_the_unreached_root_net():
    """with proposals from ground truth the root net is never called; its parameters must not require grad
    (DDP with find_unused_parameters=False would otherwise raise on the second iteration)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import train_3d as tool
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.multi_person_posenet import MultiPersonPoseNet
    cfg = load_config(None, NETWORK__USE_GT=True, MULTI_PERSON__INITIAL_CUBE_SIZE=[8, 8, 4], PICT_STRUCT__CUBE_SIZE=[8, 8, 8])
    model = MultiPersonPoseNet(None, cfg)
    params = tool.select_trainable(model, cfg)
    assert not any(p.requires_grad for p in model.root_net.parameters())
    assert all(p.requires_grad for p in model.pose_net.parameters()) and len(params) > 0


class _StubNet(torch.nn.Module):
    """stands in for MultiPersonPoseNet in engine.train_3d: on rank `dead_rank` the losses reach no parameter at all
    (what a batch without a valid proposal does to a frozen-backbone model)"""

    def __init__(self, dead_rank):
        super().__init__()
        self.backbone = None
        self.lin = torch.nn.Linear(3, 1)
        self.dead_rank = dead_rank

    def forward(self, views=None, meta=None, targets_2d=None, weights_2d=None, targets_3d=None):
        import torch.distributed as dist
        x = views[0]
        z = torch.zeros((), device=x.device)
        if dist.get_rank() == self.dead_rank:
            return None, None, None, z, z.clone(), z.clone()
        return None, None, None, z, z.clone(), self.lin(x).pow(2).mean()


def _train_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from selfpose3d_amd import distributed as D
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.engine import train_3d
    D.init("gloo")
    cfg = load_config(None, PRINT_FREQ=100)
    torch.manual_seed(0)
    net = _StubNet(dead_rank=1)
    w0 = net.lin.weight.detach().clone()
    ddp = D.wrap_ddp(net, find_unused=D.needs_find_unused(cfg))
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(10 + rank)
    batches = [([torch.randn(2, 3, generator=g)], None, None, [None], [{}], None) for _ in range(2)]
    stats = train_3d(cfg, ddp, opt, batches, 0, device=torch.device("cpu"))        # would hang if rank 1 skipped backward()
    w = [None] * world
    dist.all_gather_object(w, net.lin.weight.detach().flatten().tolist())
    q.put((rank, w[0] == w[1], bool((net.lin.weight.detach() - w0).abs().max() > 0), stats["loss"]))
    dist.barrier()
    dist.destroy_process_group()


def test_train_loop_never_skips_backward_on_one_rank():
    """round-2 review: engine.train_3d skipped backward() when its loss had no grad - under DDP one rank doing so
    deadlocks the gradient all-reduce.  Two iterations on two ranks, rank 1's loss reaches no parameter: the loop must
    finish, the replicas must stay identical, and rank 0's gradient (halved by the mean) must have moved the weights."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, "train_3d hung or failed under DDP"
    res = [q.get(timeout=10) for _ in range(world)]
    assert all(r[1] for r in res) and all(r[2] for r in res)
    assert {r[0]: r[3] for r in res}[1] == 0.0


def test_single_process_helpers_are_noops():
    from selfpose3d_amd import distributed as D
    assert D.sum_over_ranks(2, 3) == (2.0, 3.0)
    secs, last = D.timed_steps(lambda: 7, steps=2, warmup=1)
    assert last == 7 and secs >= 0.0
    assert D.shard_frames(5, 0, 1) == [0, 1, 2, 3, 4]
    assert D.max_over_ranks(2.5) == 2.5
    t = torch.arange(4.0)
    assert torch.equal(D.gather_predictions(t), t)
    m = torch.nn.Linear(2, 2)
    assert D.wrap_ddp(m) is m


def _ssv_worker(rank, world, port, q, flags):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from selfpose3d_amd import distributed as D
    from selfpose3d_amd.engine import train_3d_ssv
    from selfpose3d_amd.multi_person_posenet_ssv import get_multi_person_pose_net
    from tests import golden_io as gio
    D.init("gloo")
    torch.manual_seed(0)
    cfg = gio.train_cfg(ssv=True, **flags)
    cfg.PRINT_FREQ = 100
    model = get_multi_person_pose_net(cfg, is_train=True)
    names = {n for n, p in model.named_parameters() if p.requires_grad}
    assert any(n.startswith("attn.") for n in names)
    ddp = D.wrap_ddp(model, find_unused=D.needs_find_unused(cfg))
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    batch = gio.train_batch(cfg, B=1, seed=11 + rank, ssv=True)
    # THREE iterations: DDP(find_unused_parameters=False) raises in the second one if a parameter got no gradient in the first
    train_3d_ssv(cfg, ddp, opt, [batch, batch, batch], 0, device=torch.device("cpu"))
    got = {n for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}
    q.put((rank, sorted(names - got)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("flags", [dict(TRAIN_ONLY_2D=True), dict(USE_GT=True, INIT_TRAIN_EPOCHS_ROOTNET=5)])
def test_ssv_attention_net_is_anchored_on_early_return_paths(flags):
    """with WITH_ATTN the attention net runs but reaches no loss on the TRAIN_ONLY_2D / INIT_TRAIN_EPOCHS_ROOTNET
    (/ TRAIN_ONLY_ROOTNET / SINGLE_AUG) return paths; every trainable parameter must still get a (zero) gradient on every
    rank in every iteration, or DDP(find_unused_parameters=False) raises on step 2.  CPU-runnable variants of those paths
    (no unprojection before the return), two ranks, three iterations of the real loop."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ssv_worker, args=(r, world, port, q, flags)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, "SSV loop failed or hung under DDP(find_unused_parameters=False)"
    res = [q.get(timeout=10) for _ in range(world)]
    assert all(missing == [] for _, missing in res), res


def _split_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from selfpose3d_amd import distributed as D
    r, w = D.init_split("gloo")                    # bench.py's groups; on the GPU box the data group is RCCL
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    assert D.data_group() is not None and D.data_group() is not dist.group.WORLD
    # control plane: the timing rule's numbers are host tensors whatever `device` says
    assert D.max_over_ranks(1.0 + rank, torch.device("cpu")) == float(world)
    assert D.sum_over_ranks(1 + rank, 10) == (3.0, 20.0)
    secs, last = D.timed_steps(lambda: rank, steps=2, warmup=1)
    assert secs >= 0.0 and last == rank
    # data plane: DDP reduces its gradient buckets on the data group -> the mean of the two ranks' gradients
    torch.manual_seed(0)
    net = torch.nn.Linear(3, 2)
    ddp = D.wrap_ddp(net, find_unused=False)
    assert ddp.process_group is D.data_group()
    x = torch.full((1, 3), float(rank + 1))
    ddp(x).sum().backward()
    q.put((rank, net.weight.grad.tolist()))        # plain lists: a tensor in the queue needs the sender alive
    dist.barrier()
    D.shutdown()
    assert D.data_group() is None and not dist.is_initialized()
    D.shutdown()                                   # a second call is a no-op


def test_control_plane_on_gloo_data_plane_on_its_own_group():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_split_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # d(sum(Wx + b))/dW = x on every row; mean over the ranks' x = 1 and 2 -> 1.5
    for g in res.values():
        assert torch.allclose(torch.tensor(g), torch.full((2, 3), 1.5))


def test_deadline_fires_once_and_cancels_cleanly():
    import time
    from selfpose3d_amd import distributed as D
    fired, exits = [], []
    with D.Deadline(0.05, lambda: fired.append(1), exit_code=0, _exit=exits.append) as d:
        time.sleep(0.3)                            # the region overstays: handler, then the exit hook with the status
    assert fired == [1] and exits == [0] and d.expired
    fired2, exits2 = [], []
    with D.Deadline(5.0, lambda: fired2.append(1), _exit=exits2.append) as d2:
        pass                                       # left in time: the timer is cancelled
    time.sleep(0.1)
    assert fired2 == [] and exits2 == [] and not d2.expired
    with D.Deadline(0.0, lambda: fired2.append(1), _exit=exits2.append):      # disabled
        time.sleep(0.05)
    assert fired2 == [] and exits2 == []
    # the handler raising must not keep the process alive
    exits3 = []
    with D.Deadline(0.05, lambda: 1 / 0, exit_code=3, _exit=exits3.append):
        time.sleep(0.3)
    assert exits3 == [3]


_ONE_LINE_CHILD = r"""
import os, socket, sys
import torch, torch.multiprocessing as mp
sys.path.insert(0, os.environ["SP3D_ROOT"])

def w(rank, port):
    os.environ.update(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from selfpose3d_amd import distributed as D
    import torch.distributed as dist
    D.init_split("gloo")
    assert D.max_over_ranks(1.0 + rank) == 2.0
    ddp = D.wrap_ddp(torch.nn.Linear(2, 2), find_unused=False)
    ddp(torch.ones(1, 2)).sum().backward()
    dist.barrier()
    if rank == 0:
        print('{"line": 1}', flush=True)
    D.shutdown()

if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(w, args=(port,), nprocs=2)
"""


def test_gloo_control_plane_keeps_stdout_to_the_one_line(tmp_path):
    """gloo announces "[Gloo] Rank r is connected to n peer ranks" on STDOUT when a group connects; bench.py's contract is ONE line
    on stdout (rank 0's JSON).  init_split connects its groups with stdout pointed at stderr: two ranks, both groups used, and the
    only thing on the job's stdout is the line rank 0 printed."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = tmp_path / "one_line_child.py"         # a file: mp.spawn pickles the worker by module attribute
    child.write_text(_ONE_LINE_CHILD)
    r = subprocess.run([sys.executable, str(child)], env=dict(os.environ, SP3D_ROOT=root), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.strip().splitlines() == ['{"line": 1}'], r.stdout
    assert "[Gloo]" in r.stderr                    # the announcement went to stderr, it was not swallowed
