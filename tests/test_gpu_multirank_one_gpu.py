"""The whole multi-rank code path of bench.py on ONE GPU (round-3 review, item 3c): `python bench.py --gpus 2 --share-gpu`
launches itself under torch.distributed.run (2 ranks on 127.0.0.1), both ranks on cuda:0, process group gloo with CUDA tensors:
frames sharded by rank, max-over-ranks timing, the DDP train_step leg with its gradient all-reduce - what the CPU gloo tests
cannot see (private MIOpen db directory per rank, TunableOp files, HIP-graph capture in two processes, port choice).  RCCL
itself needs >= 2 GPUs and is the driver's run."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_share_one_gpu():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "3", "--warmup", "1",
           "--roofline-iters", "20", "--no-cpu-baseline", "--no-cold", "--no-fp32-leg", "--legs", "train_step",
           "--train-steps", "1", "--train-warmup", "1", "--train-find", "immediate"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    if r.returncode != 0:                                         # keep the ranks' own tracebacks (torchrun's summary hides them)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "multirank_bench_failure.txt"), "w") as f:
            f.write(r.stdout + "\n==== stderr ====\n" + r.stderr)
    assert r.returncode == 0, r.stderr[-6000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints the ONE line
    # ... and nothing else reaches the job's stdout (gloo announces its connections there: distributed.init_split points
    # stdout at stderr while the groups connect)
    assert [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("{")] == [], r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 8 and rec["value"] > 0
    # round 5: ranks that share a GPU load libsp3d_nopk.so (no packed-fp32 instructions), which is immune to the interaction of
    # profiles/r04_gpu_sharing_finding.md - the golden check of the step is enforced again (a failed check ends bench.py != 0)
    assert "SMOKE" in rec["config"]["parallelism"] and "share_gpu_note" in rec
    assert rec["config"]["library"] == "libsp3d_nopk.so" and rec["output_check"]["ok"], rec["output_check"]
    ts = rec["legs"]["train_step"]
    assert "error" not in ts, ts
    assert ts["n_gpus"] == 2 and ts["allreduce_bytes_per_step"] == ts["gradient_bytes"] > 100e6
    assert ts["find_unused_parameters"] is False and ts["value"] > 0 and ts["loss_last"] == ts["loss_last"]
    # round 6: the line proves how many ranks the data-plane communicator spanned and says at top level that it is whole
    assert rec["config"]["rccl_ranks_seen"] == 2 and rec["config"]["data_plane"]["ranks_seen"] == 2
    assert rec["legs_complete"] is True and rec["scaling_valid"] is True and "legs_failed" not in rec
    ov = ts["allreduce_overlap"]
    assert "error" not in ov, ov
    assert ts["allreduce_ms"] == ov["allreduce_ms"] > 0 and ov["ms_per_step_without_allreduce"] > 0 and 0.0 <= ov["overlap"] <= 1.0


def test_leg_deadline_prints_the_headline_and_every_rank_exits_cleanly():
    """A leg that does not come back (here: the train leg against a 2-second deadline - its MIOpen warm-up alone takes longer)
    must not take the headline along AND must not pass for a finished run: the deadline thread of rank 0 prints the ONE line with
    what it has, marked legs_complete / scaling_valid = false at top level, every rank exits with status 3 and the launcher
    returns non-zero (bench.py --leg-deadline, distributed.Deadline; round-5 review: rc 0 turned a hang into a success)."""
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "3", "--warmup", "1",
           "--roofline-iters", "20", "--no-cpu-baseline", "--no-cold", "--no-fp32-leg", "--legs", "train_step",
           "--train-steps", "1", "--train-warmup", "1", "--train-find", "immediate", "--leg-deadline", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0, "a run whose legs hung must not exit 0"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["ms_per_step"] > 0
    assert "deadline" in rec["legs"] and "train_step" not in rec["legs"], rec["legs"]
    assert rec["legs_complete"] is False and rec["scaling_valid"] is False
    assert rec["config"]["rccl_ranks_seen"] == 2          # the communicator itself was fine: the leg is what hung
