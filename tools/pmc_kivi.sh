#!/bin/bash
# PMC passes over the KIVI stage-1 kernel, ONE launch shape per run (developer tool):
#   tools/pmc_kivi.sh [batch] [block_seq] [tag]
# Every launch of a run has the same shape (kbench_kivi warms up with the shape it times), so the per-kernel means are
# attributable to it.  Derived figures follow profiles/r03/mfma_busy.md: SQ_* counters are summed over all SIMDs
# (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles), GRBM_GUI_ACTIVE is summed over the 8 XCDs.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=${1:-4}; BS=${2:-2304}; TAG=${3:-b${B}_bs${BS}}
O=$R/gpurun_out/pmc_kivi_$TAG
rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_LDS_ADDR_CONFLICT" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$O/p$i" -- python3 "$R/tools/kbench_kivi.py" --batches $B --block-seqs $BS --iters 4 --warmup-s 0.05 < /dev/null > "$O/p$i.log" 2>&1
done
python3 - "$O" "$B" "$BS" <<'PY' | tee "$O/summary.txt"
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kivi" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print(f"KIVI stage 1, B={sys.argv[2]} x 262152 tokens, block_seq {sys.argv[3]} (every launch of the run has this shape)")
for k in sorted(m):
    print(f"  {k:28s} launches={len(acc[k]):3d} mean={m[k]:16.1f}")
g = m.get("GRBM_GUI_ACTIVE", 0) / 8.0                      # cycles of the launch
if g > 0:
    simd_cycles = g * 1024
    q = 4.0
    def pct(x): return 100.0 * x / simd_cycles
    print(f"  -- derived (launch = {g:.0f} cycles = {g / 2.4e3:.1f} us at 2.4 GHz; 1024 SIMDs)")
    if "SQ_WAVE_CYCLES" in m: print(f"  resident waves per SIMD (avg)      {m['SQ_WAVE_CYCLES'] * q / simd_cycles:6.2f}")
    if "SQ_ACTIVE_INST_VALU" in m: print(f"  VALU active, % of SIMD cycles      {pct(m['SQ_ACTIVE_INST_VALU'] * q):6.1f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m: print(f"  MFMA busy, % of SIMD cycles        {pct(m['SQ_VALU_MFMA_BUSY_CYCLES']):6.1f}")
    if "SQ_ACTIVE_INST_LDS" in m: print(f"  LDS instruction active, %          {pct(m['SQ_ACTIVE_INST_LDS'] * q):6.1f}")
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        print(f"  LDS bank-conflict cycles / LDS active cycles  {100.0 * m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:6.1f} %")
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in m: print(f"  {k:24s} / SQ_WAVE_CYCLES  {100.0 * m[k] / w:6.1f} %")
    if "SQ_INSTS_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
        print(f"  cycles per VALU instruction        {m['SQ_ACTIVE_INST_VALU'] * q / m['SQ_INSTS_VALU']:6.2f}")
PY
