"""Scan the device assembly the build keeps (selfpose3d_amd/build/obj/*.s) for two things no counter shows directly:

* LOAD -> WAIT -> STORE CHAINS: a global / buffer load that is followed within a few instructions by `s_waitcnt vmcnt(0)` and a
  store, many times in one kernel - the shape the compiler gives `if (in_bounds) { v += res[i]; y[i] = v; }` in an unrolled
  epilogue: every element becomes a memory round trip of its own (round 5: 64 per lane in the half-resolution fused Winograd
  kernel, 9 us of an 81 us launch; profiles/r05_epilogue_fix.md);
* SERIAL LOADS: kernels most of whose loads are each followed by a full `vmcnt(0)` wait (a load inside a lane-varying branch
  whose value is used right away: NMS merge, heat-map re-tiling, root-grid scatter before round 5's fix);
* SCRATCH: any kernel with a non-zero ScratchSize (register spills).

    python tools/isa_scan.py [--chains 8] [--json]
Exit status 1 if a kernel outside the allow-list has findings (tests/test_host_cabi.py runs it on the built library)."""
import argparse
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "selfpose3d_amd", "build", "obj")

# kernels whose serial loads are by design (one dependent load per loop trip in a streaming loop that relies on occupancy)
ALLOW_SERIAL = ("gbn_bwd_stats_kernel", "upsample2x_scatter_head_kernel", "nms_merge_kernelILi16E",
                # boundary (non-interior) epilogue paths of the Winograd kernels: not taken on the grids of the plan
                "wino_fused_kernel", "wino_fused3_kernel", "wino_fused16_kernel",
                # per-view load of the camera record behind the visibility test: one per view by construction
                "unproject_bwd2_kernel")


def kernels(path):
    lines = open(path).read().split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_ZN4sp3d\w+):", l)] if m]
    for si, (a, name) in enumerate(starts):
        b = starts[si + 1][0] if si + 1 < len(starts) else len(lines)
        yield name, lines[a:b]


def scan(chain_limit=8):
    out = []
    for path in sorted(glob.glob(os.path.join(OBJ, "*.s"))):
        for name, body in kernels(path):
            loads = chains = waits0 = 0
            scratch = 0
            for i, l in enumerate(body):
                if "; ScratchSize:" in l:
                    scratch = int(l.split(":")[1])
                if "s_waitcnt vmcnt(0)" in l:
                    waits0 += 1
                if re.search(r"\b(global|buffer)_load", l):
                    loads += 1
                    w = [j for j in range(i + 1, min(i + 9, len(body))) if "s_waitcnt vmcnt(0)" in body[j]]
                    if w and any("global_store" in body[j] or "buffer_store" in body[j] for j in range(w[0], min(w[0] + 10, len(body)))):
                        chains += 1
            rec = {"file": os.path.basename(path), "kernel": name, "loads": loads, "vmcnt0_waits": waits0,
                   "load_wait_store_chains": chains, "scratch_bytes": scratch}
            rec["serial_loads"] = loads >= 8 and waits0 >= 0.5 * loads
            rec["flag"] = chains >= chain_limit or scratch > 0 or \
                (rec["serial_loads"] and not any(a in name for a in ALLOW_SERIAL))
            out.append(rec)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=8)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    recs = scan(a.chains)
    if not recs:
        print("no assembly under", OBJ, "- run `python __graft_entry__.py` first", file=sys.stderr)
        return 2
    bad = [r for r in recs if r["flag"]]
    if a.json:
        print(json.dumps({"kernels": len(recs), "flagged": bad,
                          "serial_by_design": [r["kernel"] for r in recs if r["serial_loads"] and not r["flag"]]}, indent=1))
    else:
        print(f"{len(recs)} kernels scanned, {len(bad)} flagged")
        for r in bad:
            print(f"  {r['file']}: {r['kernel'][:100]}: chains {r['load_wait_store_chains']}, loads {r['loads']}, "
                  f"vmcnt(0) {r['vmcnt0_waits']}, scratch {r['scratch_bytes']}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
