#!/bin/bash
# Regenerates the artifacts kept under profiles/rNN (run on the GPU box through gpurun; writes gpurun_out/refresh/).
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh'
# Every rocprofv3 call puts the program itself after `--`, keeps counters in their own passes and reads no stdin.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/refresh
rm -rf "$O"; mkdir -p "$O/paths"
cd /tmp && export TMPDIR=/tmp
stats() {  # stats <name> <cmd...>: kernel-trace stats csv of one command
  local name=$1; shift
  local tag=${name//\//_}; tag=${tag%.csv}
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/_prof_$tag" -- "$@" < /dev/null > "$O/_$tag.log" 2>&1
  local f; f=$(find "$O/_prof_$tag" -type f -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$O/$name" || echo "no stats for $name" >&2
}
# headline bench: plain run (the judged line) and the same command under the profiler
timeout 600 python3 "$R/bench.py" < /dev/null > "$O/bench_stdout.log" 2> "$O/bench_stderr.log"; tail -1 "$O/bench_stdout.log" > "$O/bench_line.json"
stats bench_kernel_stats.csv python3 "$R/bench.py" --steps 128 --warmup 8 --no-cpu-baseline --no-paths --no-e2e     # (the other configurations launch the same kernels at other shapes: they would pollute the per-kernel averages)
grep '^{' "$O/_bench_kernel_stats.log" | tail -1 > "$O/bench_line_profiled.json"
json_line() {  # json_line <file> <cmd...>: the command's last stdout line appended to <file>, one retry if it printed none
  local f=$1; shift
  local out
  for try in 1 2; do
    out=$(timeout 400 "$@" < /dev/null 2>> "$O/_json_line_stderr.log" | tail -1)
    [ -n "$out" ] && break
    sleep 5
  done
  echo "$out" >> "$f"
}
for b in 1 8 32; do json_line "$O/bench_small_batches.jsonl" python3 "$R/bench.py" --batch $b --no-cpu-baseline --no-kernel-events --no-e2e; done
# the roofline leg at the neighbouring batch sizes (the default is 256)
for b in 64 128 512; do json_line "$O/bench_other_batches.jsonl" python3 "$R/bench.py" --batch $b --no-cpu-baseline --no-e2e --no-paths; done
# other configurations
timeout 600 python3 "$R/tools/pathbench.py" --graph < /dev/null 2>/dev/null | grep '^{' > "$O/paths/pathbench_graph.jsonl"
timeout 600 python3 "$R/tools/pathbench.py" < /dev/null 2>/dev/null | grep '^{' > "$O/paths/pathbench_eager.jsonl"
timeout 900 python3 "$R/tools/pathbench.py" --graph --configs h2o_b64,h2o_b8,h2o_b1,streamingllm_b1,quest_b1,quest_b8,deltakv_b4 < /dev/null 2>/dev/null | grep '^{' > "$O/paths/pathbench_points.jsonl"
for c in streamingllm quest quest_b8 deltakv_raw deltakv deltakv_b4 vanilla h2o_b1; do
  stats paths/${c}_kernel_stats.csv python3 "$R/tools/pathbench.py" --graph --configs $c --steps 20
done
timeout 300 python3 "$R/tools/kbench_kivi.py" --batches 1 --block-seqs 512,1024 < /dev/null 2>/dev/null | grep kivi > "$O/paths/kbench_kivi.txt"
timeout 300 python3 "$R/tools/kbench_kivi.py" --batches 4 --block-seqs 1024,2048,2304 < /dev/null 2>/dev/null | grep kivi >> "$O/paths/kbench_kivi.txt"
timeout 300 python3 "$R/tools/kbench_kivi.py" --batches 8 --block-seqs 4480 < /dev/null 2>/dev/null | grep kivi >> "$O/paths/kbench_kivi.txt"
timeout 300 python3 "$R/tools/kbench_kivi.py" --batches 1,4 --block-seqs 1024,2304 --score < /dev/null 2>/dev/null | grep kivi >> "$O/paths/kbench_kivi.txt"
timeout 300 python3 "$R/tools/kbench_kivi.py" --batches 1,4 --block-seqs 1024,2304 --no-extra < /dev/null 2>/dev/null | grep kivi > "$O/paths/kbench_kivi_no_extra.txt"
timeout 300 python3 "$R/tools/kbench_kivi.py" --batches 1,4 --block-seqs 1024,2304 --sink 0 --tail 0 < /dev/null 2>/dev/null | grep kivi > "$O/paths/kbench_kivi_no_raw.txt"
timeout 300 python3 "$R/tools/kbench_quest.py" < /dev/null 2>/dev/null | grep quest > "$O/paths/kbench_quest.txt"
timeout 300 python3 "$R/tools/kbench_quest.py" --batch 1 < /dev/null 2>/dev/null | grep quest >> "$O/paths/kbench_quest.txt"
for v in "pages paged" "bf16 paged" "bf16" "normal paged"; do timeout 120 python3 "$R/tools/qv_bench.py" 131072 4 4672 $v < /dev/null 2>/dev/null | grep "per launch" >> "$O/paths/kbench_quest_view.txt"; done
timeout 120 python3 "$R/tools/qv_bench.py" 131072 8 4672 bf16 paged < /dev/null 2>/dev/null | grep "per launch" >> "$O/paths/kbench_quest_view.txt"
timeout 300 python3 "$R/tools/kbench.py" --graph-pair --batches 4,8 --len 4672 --block-seqs 64,96,128,160,256 --iters 40 < /dev/null 2>/dev/null | grep graph > "$O/paths/kbench_stage1_stage2_short_rows.txt"
timeout 300 python3 "$R/tools/kbench_prefill.py" < /dev/null 2>/dev/null | grep prefill > "$O/paths/kbench_prefill.txt"
timeout 300 python3 "$R/tools/kbench_prefill_score.py" < /dev/null 2>/dev/null | grep prefill_score > "$O/paths/kbench_prefill_score.txt"
stats paths/prefill_h2o_kernel_stats.csv python3 "$R/tools/prefillbench.py"
timeout 300 python3 "$R/tools/kbench.py" --batches 128,256 --block-seqs 2112,4224 --modes 2 --layers 6 < /dev/null 2>/dev/null | tail -6 > "$O/kbench_stage1.txt"
# HBM traffic of stage 1: separate counter passes, nothing else traced
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $ctr --output-format csv -d "$O/_pmc_$ctr" -- python3 "$R/tools/kbench.py" --batches 256 --block-seqs 4224 --modes 2 --iters 3 < /dev/null > "$O/_pmc_$ctr.log" 2>&1
  f=$(find "$O/_pmc_$ctr" -type f -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$O/kbench_pmc_$ctr.csv"
done
# the end-to-end leg on its own (tp = 1: B = 1 / 64 / 256, hipGraph replay and eager)
timeout 400 python3 "$R/tools/e2e_decoder.py" --batches 1,64,256 --steps 32 --modes graph,eager < /dev/null 2>/dev/null | grep '^{' > "$O/e2e_tp1.jsonl"
timeout 300 python3 "$R/tools/dql_ab.py" < /dev/null 2>/dev/null > "$O/paths/kbench_dequant_linear.txt"
timeout 300 python3 "$R/tools/kbench_up_recon.py" --rows 2048,4096,8192,16384 --hidden 2048 < /dev/null 2>/dev/null | grep layers > "$O/paths/kbench_up_recon.txt"
timeout 300 python3 "$R/tools/kbench_cluster.py" < /dev/null 2>/dev/null | grep rows > "$O/paths/kbench_cluster.txt"
timeout 300 python3 "$R/tools/kbench_topk.py" < /dev/null 2>/dev/null | grep rows > "$O/paths/kbench_topk.txt"
# per-node timelines of the replayed steps (kernel trace only)
bash "$R/tools/timeline.sh" h2o_b1 h2o_b8 quest streamingllm deltakv deltakv_b4 > /dev/null 2>&1
for f in "$R"/gpurun_out/timeline/*.txt; do cp "$f" "$O/paths/timeline_$(basename "$f")"; done
# PMC passes over the KIVI stage-1 kernel, one launch shape per run (counters only)
bash "$R/tools/pmc_kivi.sh" 1 1024 > /dev/null 2>&1; cp "$R/gpurun_out/pmc_kivi_b1_bs1024/summary.txt" "$O/pmc_kivi_b1.txt"
bash "$R/tools/pmc_kivi.sh" 4 2304 > /dev/null 2>&1; cp "$R/gpurun_out/pmc_kivi_b4_bs2304/summary.txt" "$O/pmc_kivi_b4.txt"
# bare access-pattern and instruction probes (built here by hipcc, see the header of each file)
for p in probe_gather probe_kdma probe_dma_offset mfma_valu_mix probe_fp8cvt probe_tr4; do
  [ -x "$R/tools/bin/$p" ] && timeout 120 "$R/tools/bin/$p" < /dev/null > "$O/$p.txt" 2>&1
done
rm -rf "$O"/_prof_* "$O"/_pmc_*/
ls -la "$O" "$O/paths"
