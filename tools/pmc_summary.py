#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output (one directory per pass) into per-kernel means per dispatch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if "sp3d" not in name:
                    continue
                short = name.split("(")[0].replace("void ", "")
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for k, d in acc.items():
        res[k] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in sorted(d.items())}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
