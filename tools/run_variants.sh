set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
python -m pytest tests/ -x -q -m gpu 2>&1 | tail -2
run() { echo "== $1"; env $1 python -m pytest tests/ -x -q -m gpu -k "$2" 2>&1 | tail -1; }
run SVK_STAGE1_VARIANT=4 "h2o or decode or stage1 or large or fuzz"
run SVK_H2O_DEFER_SCORE=0 "h2o"
run SVK_H2O_DEFER_SCORE=1 "h2o"
run SVK_PREFILL_ATTN_VARIANT=1 "prefill or context or large"
run SVK_PREFILL_ATTN_HELPER=0 "prefill or context or large"
run SVK_PREFILL_SCORE_VARIANT=1 "prefill_score or snapkv or h2o"
run SVK_KIVI_VARIANT=4 "kivi or deltakv"
run SVK_KIVI_VARIANT=2 "kivi or deltakv"
run SVK_FUSE_DECODE_STORE=0 "h2o or decode or streamingllm or quest or vanilla"
run SVK_DELTAKV_RECON_AHEAD=0 "deltakv"
run SVK_DELTAKV_FUSE_RAW_STORE=0 "deltakv"
run SVK_QUEST_VIEW_VARIANT=1 "quest"
run SVK_DECODE_DIRECT_OUT=0 "h2o or decode or streamingllm or fuzz or full_size"
