#!/bin/bash
# developer tool: rocprofv3 kernel summary of one pathbench configuration, per decode step -> gpurun_out/prof_path/<cfg>*
#   tools/prof_path.sh deltakv [steps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CFG=$1; STEPS=${2:-48}
O=$R/gpurun_out/prof_path
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/_p_$CFG"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/_p_$CFG" -- python3 "$R/tools/pathbench.py" --graph --configs $CFG --steps $STEPS --warmup 4 < /dev/null > "$O/$CFG.log" 2>&1
f=$(find "$O/_p_$CFG" -type f -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$O/${CFG}_kernel_stats.csv"
rm -rf "$O/_p_$CFG"
grep '^{' "$O/$CFG.log" | tail -1 | cut -c1-160
python3 - "$O/${CFG}_kernel_stats.csv" "$O/$CFG.log" <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
line = [l for l in open(sys.argv[2]) if l.startswith("{")][-1]
d = json.loads(line)
gs = d.get("graph_steps", {})
steps = sum(gs.values()) if gs else d["steps"]
tot = 0.0
print(f"  per step over {steps} steps (graph replay + warm-up), sorted by time per step")
out = []
for r in rows:
    per = float(r["TotalDurationNs"]) / steps / 1e3
    out.append((per, r))
out.sort(key=lambda x: -x[0])
for per, r in out[:22]:
    n = r["Name"].replace("void ", "").replace("svk::(anonymous namespace)::", "")[:74]
    print(f"  {n:74s} {int(r['Calls'])/steps:6.1f}/step {float(r['AverageNs'])/1e3:8.2f} us avg {per:8.1f} us/step")
    tot += per
print(f"  sum of the listed: {tot:.1f} us/step; all kernels: {sum(p for p, _ in out):.1f} us/step")
PY
