"""bench.py pieces that can be checked without a GPU: the reference-golden output check really discriminates, the static
reference-CPU record is well formed, and the argument defaults follow the driver's contract."""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_output_check_accepts_the_reference_and_rejects_a_perturbed_output():
    from selfpose3d_amd import synthetic as syn
    bench = _bench()
    g = np.load(os.path.join(ROOT, "tests", "golden", "rootnet_full.npz"))
    N = int(np.prod(syn.INITIAL_CUBE_SIZE))
    # rebuild a full volume that agrees with the golden on its sub-sample and on its checksums
    root = np.zeros((2, N), np.float32)
    root[:, g["sub_idx"]] = g["root_sub"]
    rest = np.setdiff1d(np.arange(N), g["sub_idx"])
    for b in range(2):
        root[b, rest] = (g["root_sum"][b] - root[b].astype(np.float64).sum()) / len(rest)
    cs = torch.tensor(syn.INITIAL_CUBE_SIZE, dtype=torch.float32)
    gs, cen = torch.tensor(syn.SPACE_SIZE), torch.tensor(syn.SPACE_CENTER)
    gc = torch.zeros(2, 10, 5)
    gc[:, :, :3] = torch.from_numpy(g["nms_idx"]).float() / (cs - 1) * gs + cen - gs / 2.0
    gc[:, :, 4] = torch.from_numpy(g["nms_vals"])
    out = (torch.from_numpy(root).view(2, *syn.INITIAL_CUBE_SIZE), gc)
    ok = bench.check_output(out, g)
    assert ok["ok"] and ok["proposal_indices_checked"] >= 10 and ok["proposal_indices_wrong"] == 0
    bad = (out[0].clone(), gc.clone())
    bad[0].view(2, -1)[0, int(g["sub_idx"][5])] += 0.01                 # one voxel off by 1e-2
    assert not bench.check_output(bad, g)["ok"]
    bad = (out[0], gc.clone())
    bad[1][1, 0, 0] += 101.2658                                          # best proposal one voxel to the side
    r = bench.check_output(bad, g)
    assert not r["ok"] and r["proposal_indices_wrong"] == 1


def test_cpu_reference_record_and_defaults():
    bench = _bench()
    rec = bench.cpu_reference_record()
    assert rec is not None and rec["kind"] == "reference" and rec["unit"] == "samples/s"
    assert rec["value"] > rec["value_1_thread"] > 0 and rec["cores"] == rec["host"]["logical_cpus"]
    raw = json.load(open(os.path.join(ROOT, "profiles", "cpu_reference.json")))
    assert "configs[1] B=4 960x512->240x128" in raw["configs"]
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = argv
    assert (a.gpus, a.batch) == (1, 4) and a.steps >= 20 and a.warmup >= 1 and not a.planar_input
    assert bench.HBM_PEAK_GBS == 8000.0


def test_gpus_flag_never_runs_fewer_ranks_silently(monkeypatch):
    """round-3 review: `python bench.py --gpus 8` ran ONE rank and printed n_gpus = 1.  Now: without a launcher it re-execs
    itself under torch.distributed.run with --nproc-per-node N on 127.0.0.1 (or refuses when the box has fewer GPUs), and
    with a launcher the world size must equal --gpus."""
    import pytest
    bench = _bench()
    # (a) no launcher, not enough GPUs: loud refusal (this container has none)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "--gpus 8" in str(e.value) and "GPU(s) visible" in str(e.value)
    # (b) no launcher, --share-gpu (or enough GPUs): the exec'd command is the task statement's launch line
    seen = {}

    def fake_execv(exe, cmd):
        seen["cmd"] = cmd
        raise SystemExit("exec")
    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--share-gpu", "--steps", "3"])
    with pytest.raises(SystemExit):
        bench.main()
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-5:] == ["--gpus", "2", "--share-gpu", "--steps", "3"]
    # (c) launcher with another world size: refused, never a line with the wrong n_gpus
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=1" in str(e.value)


def test_host_share_pins_restores_and_cleans_up():
    """legs.host_contention (bench.py): the process is confined to 1/8 of the cores and released again; the burner
    processes it started are gone afterwards; the record carries step time and host CPU time per condition"""
    import os
    from selfpose3d_amd import distributed as D
    before = os.sched_getaffinity(0)
    with D.HostShare(burners=True) as h:
        inside = os.sched_getaffinity(0)
        pids = [p.pid for p in h.procs]
        assert len(inside) == max(1, len(before) // 8) and inside <= before
        assert len(pids) == 7 * 2                      # 7 neighbour ranks x 2 busy host threads each
    assert os.sched_getaffinity(0) == before
    assert not any(os.path.exists(f"/proc/{p}") for p in pids)
    rec = D.host_contention(lambda: sum(range(2000)), 5)
    assert set(rec) == {"unconstrained", "one_eighth_of_the_cores_1_thread", "one_eighth_of_the_cores_7_busy_neighbour_ranks",
                        "container_cpu_quota_cores"}
    for r in (v for v in rec.values() if isinstance(v, dict)):
        assert r["ms_per_step"] > 0 and r["host_ms_per_step"] > 0 and "vs_unconstrained" in r
    assert os.sched_getaffinity(0) == before


def test_burners_die_with_their_parent():
    """round-5 advice: a bench.py killed inside legs.host_contention left 14 endless spinners behind.  A child process that
    enters HostShare(burners=True) and is then SIGKILLed: its spinners are gone within a moment (PR_SET_PDEATHSIG), and a
    spinner left to itself ends after max_seconds"""
    import os
    import signal
    import subprocess
    import sys
    import time
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from selfpose3d_amd import distributed as D\n"
            "h = D.HostShare(burners=True, max_seconds=120).__enter__()\n"
            "print(' '.join(str(p.pid) for p in h.procs), flush=True)\n"
            "time.sleep(600)\n" % ROOT)
    child = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True)
    try:
        pids = [int(x) for x in child.stdout.readline().split()]
        assert len(pids) == 14 and all(os.path.exists(f"/proc/{p}") for p in pids)
        os.kill(child.pid, signal.SIGKILL)                       # exactly the process started above
        child.wait(timeout=10)
        deadline = time.time() + 10
        while time.time() < deadline and any(os.path.exists(f"/proc/{p}") for p in pids):
            time.sleep(0.1)
        assert not any(os.path.exists(f"/proc/{p}") for p in pids), "spinners outlived their parent"
    finally:
        if child.poll() is None:
            child.kill()
    from selfpose3d_amd import distributed as D
    t0 = time.time()
    with D.HostShare(burners=True, max_seconds=1) as h:
        while time.time() - t0 < 8 and any(p.poll() is None for p in h.procs):
            time.sleep(0.1)
        assert all(p.poll() is not None for p in h.procs), "a spinner ignored its own time bound"
