#!/bin/bash
# developer tool: SQ / GRBM counters of the kernels whose name contains <pattern> while <command...> runs (counters only,
# one group per pass):   tools/pmc_kernel.sh <pattern> <out-tag> <command...>
# <command...> must START with the program itself (python3 <script> ... or an ELF binary): with --pmc the profiler's
# preloaded library initialises the GPU before the program starts, so a launcher hop (env VAR=..., bash -c, sh -c, taskset,
# numactl, a script run through its #! line) is an exec from a GPU-initialised process - on this pool that takes the box
# down.  Export knobs BEFORE calling this script:   SVK_UP_RECON_TM=256 tools/pmc_kernel.sh up_recon ur256 python3 tools/ur_one.py
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PAT=$1; TAG=$2; shift 2
case "$(basename "$1")" in
  python3|python) ;;
  env|bash|sh|taskset|numactl|timeout|nice|*.py|*.sh)
    echo "pmc_kernel.sh: refusing launcher '$1' under rocprofv3 --pmc (see the header); start with python3 <script> or a binary" >&2; exit 2 ;;
  *) if ! head -c 4 "$(command -v "$1")" 2>/dev/null | grep -q ELF; then
       echo "pmc_kernel.sh: '$1' is not an ELF binary (a script would re-exec through its interpreter)" >&2; exit 2; fi ;;
esac
O=$R/gpurun_out/pmc_$TAG
rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$O/p$i" -- "$@" < /dev/null > "$O/p$i.log" 2>&1
done
python3 - "$O" "$PAT" <<'PY' | tee "$O/summary.txt"
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print(f"kernels matching {sys.argv[2]!r}")
for k in sorted(m):
    print(f"  {k:28s} launches={len(acc[k]):3d} mean={m[k]:16.1f}")
g = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
if g > 0:
    simd = g * 1024
    print(f"  -- derived (launch = {g:.0f} cycles = {g / 2.4e3:.1f} us at 2.4 GHz)")
    if "SQ_WAVE_CYCLES" in m: print(f"  resident waves per SIMD (avg)      {m['SQ_WAVE_CYCLES'] * 4 / simd:6.2f}")
    if "SQ_ACTIVE_INST_VALU" in m: print(f"  VALU active, % of SIMD cycles      {100 * m['SQ_ACTIVE_INST_VALU'] * 4 / simd:6.1f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m: print(f"  MFMA busy, % of SIMD cycles        {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd:6.1f}")
    if "SQ_LDS_IDX_ACTIVE" in m: print(f"  LDS active cycles, % of CU cycles  {100 * m['SQ_LDS_IDX_ACTIVE'] / (g * 256):6.1f}   bank-conflict share {100 * m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:5.1f} %")
    if "SQ_WAVE_CYCLES" in m:
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in m: print(f"  {k:24s} / SQ_WAVE_CYCLES  {100.0 * m[k] / m['SQ_WAVE_CYCLES']:6.1f} %")
    if "FETCH_SIZE" in m: print(f"  HBM read  {m['FETCH_SIZE'] * 2 * 1024 / 1e6:8.1f} MB (FETCH_SIZE x 2 KB)   write {m.get('WRITE_SIZE', 0) * 1024 / 1e6:8.1f} MB")
PY
