#!/bin/bash
# developer tool: DeltaKV decode step time against the look-ahead reconstruction schedule (SVK_DELTAKV_RECON_BATCH)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for b in "$@"; do
  echo -n "RECON_BATCH=$b  "
  SVK_DELTAKV_RECON_BATCH=$b python3 "$R/tools/pathbench.py" --graph --configs deltakv --steps 48 2>/dev/null | grep '^{' | cut -c1-60
done
