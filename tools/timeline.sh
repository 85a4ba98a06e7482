#!/bin/bash
# developer tool: per-node timeline (duration and gap before each kernel of the replayed graph) of pathbench configurations
#   tools/timeline.sh h2o_b1 quest streamingllm   -> gpurun_out/timeline/<config>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/timeline
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$O/_p$c" -- python3 "$R/tools/pathbench.py" --configs $c --steps 64 --graph < /dev/null > "$O/$c.log" 2>&1
  { echo "== $c: the last replayed steps (tools/trace_gaps.py: mean duration and mean idle gap before each node) =="; grep '^{' "$O/$c.log" | tail -1 | cut -c1-160; python3 "$R/tools/trace_gaps.py" "$O/_p$c"; } > "$O/$c.txt" 2>&1
  rm -rf "$O/_p$c"
  cat "$O/$c.txt"
done
