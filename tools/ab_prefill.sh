#!/bin/bash
# Development helper: A/B two builds of libsvk.so (tools/bin/ab/libsvk_{A,B}.so) on the SAME GPU box, alternating,
# with tools/kbench_prefill.py (boxes of the pool differ by a few percent, so runs from two gpurun calls do not compare).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CMD=${1:-"python3 $R/tools/kbench_prefill.py"}
for round in 1 2; do
  for v in A B; do
    cp "$R/tools/bin/ab/libsvk_$v.so" "$R/sparse_vllm_amd/libsvk.so"
    echo "== $v (round $round)"
    $CMD 2>/dev/null | grep -v "^$"
  done
done
cp "$R/tools/bin/ab/libsvk_B.so" "$R/sparse_vllm_amd/libsvk.so"
