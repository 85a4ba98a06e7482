#!/bin/bash
# developer tool: the whole `-m gpu` suite under every opt-out knob of INTEGRATION.md section 4 (one gpurun call):
#   gpurun --timeout 2400 -- 'bash tools/variant_matrix.sh > gpurun_out/variants.txt 2>&1'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
run() {  # run <label> [VAR=value ...]
  local label=$1; shift
  local res
  res=$(env "$@" timeout 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -1)
  printf '  %-36s %s\n' "$label" "$res"
}
echo "The whole \`-m gpu\` suite under each opt-out knob of INTEGRATION.md section 4, one box:"
run "(default)" SVK_NONE=1
for kv in SVK_STAGE2_SPLIT=0 SVK_DQL_ROWS=0 SVK_DQL_GELU_TABLE=0 SVK_FUSE_DECODE_STORE=0 SVK_DECODE_DIRECT_OUT=0 SVK_H2O_DEVICE_STATE=0 SVK_H2O_SCORE_PREFILL=1 \
          SVK_PREFILL_SCORE_FUSE=0 SVK_DELTAKV_RECON_AHEAD=0 SVK_DELTAKV_FUSE_RAW_STORE=0 SVK_DELTAKV_FUSE_FULL_STORE=0 \
          SVK_DELTAKV_FUSED_UP=0 SVK_DELTAKV_FUSED_CLUSTER=0 SVK_DELTAKV_ROTATED_STORE=0 SVK_DELTAKV_ROTATED_POS=slot SVK_DELTAKV_RECON_LOAD_FIRST=0 SVK_TOPK_PLAN=hist SVK_TOPK_FINAL=select SVLLM_DEBUG_DECODE_BOUNDS=device; do
  run "$kv" "$kv"
done
res=$(SVK_FUZZ_SCALE=8 timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -q -p no:cacheprovider 2>&1 | tail -1)
echo "SVK_FUZZ_SCALE=8 tests/test_gpu_fuzz.py: $res"
