"""RCCL on the test box (one GPU): a world-size-1 process group with backend "nccl" - communicator creation with
``device_id`` as selfpose3d_amd.distributed.init does it, all-reduce / all-gather / broadcast / barrier on device tensors, and a
DistributedDataParallel train step of the V2V net (32 MB buckets, gradient_as_bucket_view, grouped BatchNorm inside) whose
gradients must equal the unwrapped net's.  The 2-rank path is covered on CPU with gloo (tests/test_distributed_gloo.py) and on one
GPU with gloo (tests/test_gpu_multirank_one_gpu.py); what neither touches is the RCCL library itself on this image and GPU -
this does, before the driver's multi-GPU run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["SP3D_ROOT"])
from selfpose3d_amd import distributed as D, synthetic as syn
from selfpose3d_amd.v2v_net import V2VNet

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)          # eager communicator creation (what D.init does when world > 1)
rec = {"backend": dist.get_backend(), "world": dist.get_world_size(), "nccl_version": list(torch.cuda.nccl.version())}
t = torch.arange(8, dtype=torch.float32, device=dev)
dist.all_reduce(t); rec["all_reduce_sum_ok"] = bool(torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32)))
m = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(m, op=dist.ReduceOp.MAX); rec["all_reduce_max_ok"] = float(m.item()) == 3.5
parts = [torch.empty(5, 3, device=dev)]
src = torch.rand(5, 3, device=dev)
dist.all_gather(parts, src); rec["all_gather_ok"] = bool(torch.equal(parts[0], src))
b = torch.full((4,), 7.0, device=dev); dist.broadcast(b, 0); rec["broadcast_ok"] = bool((b == 7).all())
big = torch.ones(40 * 1024 * 1024 // 4, device=dev); dist.all_reduce(big); rec["all_reduce_40MB_ok"] = bool((big == 1).all())
dist.barrier(); torch.cuda.synchronize(dev)

def net():
    n = V2VNet(4, 4)
    syn.fill_parameters_deterministic(n, seed=5, scale=0.05)
    return n.to(dev).train()
x = torch.rand(2, 4, 16, 16, 16, generator=torch.Generator().manual_seed(3)).to(dev)
plain, wrapped = net(), net()
ddp = torch.nn.parallel.DistributedDataParallel(wrapped, device_ids=[0], output_device=0, find_unused_parameters=True, bucket_cap_mb=32,
                                                gradient_as_bucket_view=True)
for mod, out_key in ((plain, "plain"), (ddp, "ddp")):
    mod(x).square().mean().backward()
torch.cuda.synchronize(dev)
worst = 0.0
for (n1, p1), (n2, p2) in zip(plain.named_parameters(), wrapped.named_parameters()):
    worst = max(worst, float((p1.grad - p2.grad).abs().max()))
rec["ddp_grad_max_abs_diff_vs_plain"] = worst
rec["ddp_buffers_equal"] = all(bool(torch.equal(b1, b2)) for b1, b2 in zip(plain.buffers(), wrapped.buffers()))
sec, _ = D.timed_steps(lambda: ddp(x).sum().item(), 3, 1, dev)
rec["timed_steps_s"] = sec
dist.destroy_process_group()
print("REC " + json.dumps(rec))
"""


def test_rccl_process_group_and_ddp_step_on_one_rank(tmp_path):
    env = dict(os.environ, SP3D_ROOT=ROOT, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29631", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SP3D_SHARED_GPU", None)
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("REC ")][-1][4:])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_single_rank.json"), "w") as f:
        json.dump(rec, f, indent=1)
    assert rec["backend"] == "nccl" and rec["world"] == 1
    for k in ("all_reduce_sum_ok", "all_reduce_max_ok", "all_gather_ok", "broadcast_ok", "all_reduce_40MB_ok", "ddp_buffers_equal"):
        assert rec[k] is True, (k, rec)
    assert rec["ddp_grad_max_abs_diff_vs_plain"] == 0.0, rec      # one rank: the bucket view IS the gradient


SPLIT_CHILD = r"""
import json, os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["SP3D_ROOT"])
from selfpose3d_amd import distributed as D, synthetic as syn
from selfpose3d_amd.v2v_net import V2VNet

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
D.init_split("nccl", single_rank=True)                 # bench.py's layout at world > 1: control plane gloo, data plane RCCL
rec = {"control_backend": dist.get_backend(), "data_backend": dist.get_backend(D.data_group()), "world": dist.get_world_size()}
rec["max_over_ranks"] = D.max_over_ranks(2.5, dev)      # a host tensor on the gloo group although `dev` is the GPU
sec, last = D.timed_steps(lambda: torch.ones(4, device=dev).sum().item(), 3, 1, dev)
rec["timed_steps_ok"] = bool(sec >= 0.0 and last == 4.0)
t = torch.arange(8, dtype=torch.float32, device=dev)
dist.all_reduce(t, group=D.data_group())                # first collective of the lazily built RCCL communicator
rec["data_all_reduce_ok"] = bool(torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32)))

def net():
    n = V2VNet(4, 4)
    syn.fill_parameters_deterministic(n, seed=5, scale=0.05)
    return n.to(dev).train()
x = torch.rand(2, 4, 16, 16, 16, generator=torch.Generator().manual_seed(3)).to(dev)
plain, wrapped = net(), net()
ddp = torch.nn.parallel.DistributedDataParallel(wrapped, device_ids=[0], output_device=0, find_unused_parameters=False, bucket_cap_mb=32,
                                                gradient_as_bucket_view=True, process_group=D.data_group())
for mod in (plain, ddp):
    mod(x).square().mean().backward()
torch.cuda.synchronize(dev)
rec["ddp_on_data_group"] = ddp.process_group is D.data_group()
rec["ddp_grad_max_abs_diff_vs_plain"] = max(float((p1.grad - p2.grad).abs().max())
                                            for p1, p2 in zip(plain.parameters(), wrapped.parameters()))
dist.barrier()
dist.destroy_process_group()
print("REC " + json.dumps(rec))
"""


def test_bench_group_layout_control_gloo_data_rccl_on_one_rank():
    env = dict(os.environ, SP3D_ROOT=ROOT, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29633", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SP3D_SHARED_GPU", None)
    r = subprocess.run([sys.executable, "-c", SPLIT_CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("REC ")][-1][4:])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_split_groups.json"), "w") as f:
        json.dump(rec, f, indent=1)
    assert rec["control_backend"] == "gloo" and rec["data_backend"] == "nccl" and rec["world"] == 1
    assert rec["max_over_ranks"] == 2.5 and rec["timed_steps_ok"] and rec["data_all_reduce_ok"] and rec["ddp_on_data_group"]
    assert rec["ddp_grad_max_abs_diff_vs_plain"] == 0.0, rec
