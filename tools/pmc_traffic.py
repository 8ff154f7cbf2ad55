#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a tools/collect_pmc.sh summary: HBM-side bytes per launch of the
unprojection kernel, corrected as MI355X_MICROARCH.md §HBM prescribes for gfx950 (FETCH_SIZE counts
128-B fabric reads at 64 B: double it; cross-checked against TCC_EA0_RDREQ_128B*128 + _64B*64;
WRITE_SIZE is exact for this kernel: it equals the cubes tensor size).

    python tools/pmc_traffic.py gpurun_out/pmc_xxx/summary.json B4_V5_J15_240x128_80x80x20 > profiles/pmc_traffic.json
"""
import json
import sys


def main():
    summ = json.load(open(sys.argv[1]))
    workload = sys.argv[2]
    k = [n for n in summ if "unproject" in n][0]
    d = {c: v["mean"] for c, v in summ[k].items()}
    fetch = 2.0 * d["FETCH_SIZE"] * 1024.0
    fetch_xcheck = d.get("TCC_EA0_RDREQ_128B_sum", 0) * 128.0 + d.get("TCC_EA0_RDREQ_64B_sum", 0) * 64.0 + \
        d.get("TCC_EA0_RDREQ_32B_sum", 0) * 32.0
    write = d["WRITE_SIZE"] * 1024.0
    out = {
        "workload": workload, "kernel": k,
        "hbm_bytes_per_launch": int(fetch + write),
        "read_bytes": int(fetch), "read_bytes_from_rdreq_counters": int(fetch_xcheck), "write_bytes": int(write),
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 gfx950 correction",
        "l2": {"TCC_HIT": d.get("TCC_HIT_sum"), "TCC_MISS": d.get("TCC_MISS_sum"), "TCC_REQ": d.get("TCC_REQ_sum")},
        "l1": {"TCP_TOTAL_CACHE_ACCESSES": d.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
               "TCP_TCC_READ_REQ": d.get("TCP_TCC_READ_REQ_sum"),
               "avg_l2_read_latency_cycles": (d.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) /
                                              max(1.0, d.get("TCP_TCC_READ_REQ_sum", 1)))},
        "sq": {c: d.get(c) for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SALU",
                                     "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                                     "SQ_BUSY_CYCLES")},
        "ta": {"TA_TA_BUSY_sum": d.get("TA_TA_BUSY_sum"), "TA_BUSY_avr": d.get("TA_BUSY_avr"),
               "TA_FLAT_READ_WAVEFRONTS": d.get("TA_FLAT_READ_WAVEFRONTS_sum")},
        "grbm_gui_active": d.get("GRBM_GUI_ACTIVE"),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
