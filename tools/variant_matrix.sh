#!/bin/bash
# The -m gpu suite under every opt-in kernel variant / knob (developer tool, run through gpurun) -> gpurun_out/variants.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/variants.txt; : > $O
run() {
  echo "== $*" | tee -a $O
  env "$@" timeout 900 python3 -m pytest tests -q -m gpu < /dev/null 2>&1 | grep -E "^FAILED|passed|failed" | tail -6 | tee -a $O
}
run SVK_DUMMY=1
run SVK_KIVI_VARIANT=2
run SVK_KIVI_VARIANT=3
run SVK_KIVI_VARIANT=4 SVK_KIVI_BLOCK_SEQ=auto
run SVK_H2O_DEVICE_STATE=0
run SVK_TOPK_PLAN=chunks SVK_DQL_WIDE=0 SVK_DELTAKV_BIAS_COLUMN=0 SVK_DELTAKV_SCORE_REFILL=1 SVK_DELTAKV_RECON_BATCH=3
run SVK_H2O_DEFER_SCORE=0
run SVK_H2O_DEFER_SCORE=1
run SVK_STAGE1_VARIANT=4
run SVK_STAGE1_VARIANT=2
run SVK_PREFILL_SCORE_FUSE=0 SVK_PREFILL_ATTN_VARIANT=1 SVK_PREFILL_SCORE_VARIANT=1
run SVK_DELTAKV_RECON_AHEAD=0 SVK_DELTAKV_FUSED_UP=0 SVK_DELTAKV_FUSE_RAW_STORE=0 SVK_QUEST_VIEW_VARIANT=1
run SVK_FUSE_DECODE_STORE=0 SVK_DECODE_DIRECT_OUT=0
run SVK_PREFILL_ATTN_HELPER=0 SVK_DELTAKV_FUSE_FULL_STORE=0
