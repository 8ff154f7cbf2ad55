#!/bin/bash
# per-kernel average durations of a command (rocprofv3 --kernel-trace --stats), filtered by a name pattern
#   bash tools/kstats.sh <pattern> <script.py relative to the repo root> [args...]
PAT=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kstats_out
SCRIPT=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats_out -o k -- python "$R/$SCRIPT" "$@" > /tmp/kstats.log 2>&1 || tail -5 /tmp/kstats.log
f=$(find /tmp/kstats_out -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$PAT" <<'PY'
import csv, sys, re
f, pat = sys.argv[1], sys.argv[2]
for r in csv.DictReader(open(f)):
    if re.search(pat, r["Name"]):
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.2f} min_us {float(r["MinNs"])/1e3:9.2f}')
PY
