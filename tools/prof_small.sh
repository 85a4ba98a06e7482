#!/bin/bash
# developer tool: kernel summary of the H2O decode step at small batches -> gpurun_out/prof_small/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_small
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for b in "$@"; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/_p$b" -- python3 "$R/bench.py" --batch $b --steps 128 --warmup 8 --no-cpu-baseline --no-paths --no-kernel-events --no-e2e < /dev/null > "$O/b$b.log" 2>&1
  f=$(find "$O/_p$b" -type f -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$O/b${b}_kernel_stats.csv"
  rm -rf "$O/_p$b"
  grep '^{' "$O/b$b.log" | tail -1 | cut -c1-200
  python3 - "$O/b${b}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n = r['Name'].replace('void ', '').replace('svk::(anonymous namespace)::', '')[:80]
    print(f"  {n:80s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us {r['Percentage']:>6}%")
PY
done
