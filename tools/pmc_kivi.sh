#!/bin/bash
# PMC passes over the KIVI stage-1 kernel (developer tool): tools/pmc_kivi.sh [batch] [block_seq]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=${1:-4}; BS=${2:-2048}
O=$R/gpurun_out/pmc_kivi
mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$O/p$i" -- python3 "$R/tools/kbench_kivi.py" --batches $B --block-seqs $BS --iters 2 < /dev/null > "$O/p$i.log" 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kivi" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} launches={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
