#!/bin/bash
# DeltaKV path A/B (developer tool, run through gpurun): parity tests of the path, the two pathbench configurations and a
# kernel summary of the 256k one -> gpurun_out/ab_deltakv/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ab_deltakv
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1200 python3 -m pytest tests/test_gpu_deltakv.py tests/test_gpu_deltakv_e2e.py tests/test_gpu_deltakv_prefill.py tests/test_gpu_reference_pins.py tests/test_gpu_kivi.py -x -q -m gpu < /dev/null > "$O/tests.log" 2>&1
tail -5 "$O/tests.log"
timeout 600 python3 tools/pathbench.py --graph --configs deltakv,deltakv_raw --steps 32 < /dev/null 2>/dev/null | grep '^{' | tee "$O/pathbench.jsonl"
SVK_TOPK_PLAN=chunks timeout 600 python3 tools/pathbench.py --graph --configs deltakv --steps 32 < /dev/null 2>/dev/null | grep '^{' | tee "$O/pathbench_chunks.jsonl"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/_prof" -- python3 "$R/tools/pathbench.py" --graph --configs deltakv --steps 20 < /dev/null > "$O/prof.log" 2>&1
f=$(find "$O/_prof" -type f -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$O/deltakv_kernel_stats.csv"
rm -rf "$O/_prof"
python3 - "$O/deltakv_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    n = r['Name'].replace('void ', '').replace('svk::(anonymous namespace)::', '')[:70]
    print(f"{n:70s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/24e3:8.1f} us/step")
PY
