#!/bin/bash
# developer tool: per-kernel summary (rocprofv3 --kernel-trace --stats) of one python tool.
#   tools/kstats.sh <tag> tools/kbench_topk.py [args...]   -> gpurun_out/kstats_<tag>.txt   (knobs: export them before the call)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out/_kstats_$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- python3 "$R/$1" "${@:2}" < /dev/null > "$O.log" 2>&1
f=$(find "$O" -type f -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY' | tee "$R/gpurun_out/kstats_$TAG.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    n = r["Name"].replace("void ", "").replace("svk::(anonymous namespace)::", "")[:84]
    print(f"{n:84s} {int(r['Calls']):7d}  avg {float(r['AverageNs']) / 1e3:9.2f} us  min {float(r['MinNs']) / 1e3:8.2f}  max {float(r['MaxNs']) / 1e3:8.2f}  {r['Percentage']:>6}%")
PY
rm -rf "$O"
