#!/bin/bash
# MFMA-busy of the two MFMA-bound kernels by PMC (counters only, nothing else traced) -> gpurun_out/mfma/{pmc_prefill_attention.csv,
# pmc_prefill_score.csv, mfma_busy.md}.  Run through gpurun: bash tools/pmc_mfma.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/mfma
rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$O/_a" -- python3 "$R/tools/kbench_prefill.py" --iters 2 < /dev/null > "$O/a.log" 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$O/_s" -- python3 "$R/tools/kbench_prefill_score.py" --iters 2 --keys 16384 < /dev/null > "$O/s.log" 2>&1
fa=$(find "$O/_a" -name '*counter_collection.csv' | head -1); fs=$(find "$O/_s" -name '*counter_collection.csv' | head -1)
python3 - "$fa" "$fs" "$O" <<'PY'
import csv, sys, collections, statistics
fa, fs, O = sys.argv[1:4]
def load(path, match):
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        if match in r["Kernel_Name"]:
            rows[(r["Kernel_Name"].replace("void ", "").replace("svk::(anonymous namespace)::", "").split("(")[0], r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    return rows
out = ["# MFMA-busy of the two MFMA-bound kernels (PMC, re-measured on the current kernels)", "",
       "`tools/pmc_mfma.sh`: `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` (counters only) over",
       "`tools/kbench_prefill.py --iters 2` and `tools/kbench_prefill_score.py --iters 2 --keys 16384`.  MFMA-busy =",
       "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) (the busy counter is summed over the SIMDs, GUI_ACTIVE over",
       "the 8 XCDs); median over the launches of a shape, launches ordered by their MFMA work.", "",
       "| kernel | MFMA busy cycles | GRBM_GUI_ACTIVE / 8 | MFMA-busy |", "|---|---:|---:|---:|"]
for path, match, keep in ((fa, "context_attention_kernel", "pmc_prefill_attention.csv"), (fs, "prefill_score_kernel", "pmc_prefill_score.csv")):
    rows = load(path, match)
    with open(O + "/" + keep, "w") as f:
        f.write("kernel,dispatch,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,GRBM_GUI_ACTIVE\n")
        for (k, d), c in sorted(rows.items(), key=lambda kv: int(kv[0][1])):
            f.write(f"{k},{d},{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.0f},{c.get('SQ_BUSY_CYCLES', 0):.0f},{c.get('GRBM_GUI_ACTIVE', 0):.0f}\n")
    groups = collections.defaultdict(list)
    for (k, d), c in rows.items():
        if c.get("GRBM_GUI_ACTIVE", 0) > 0:
            groups[(k, round(c["SQ_VALU_MFMA_BUSY_CYCLES"]))].append(c)
    for (k, busy), cs in sorted(groups.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        gui = statistics.median(c["GRBM_GUI_ACTIVE"] for c in cs) / 8.0
        out.append(f"| `{k}` | {busy:,} | {gui:,.0f} | **{busy / (1024.0 * gui) * 100:.1f} %** |")
open(O + "/mfma_busy.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf "$O/_a" "$O/_s"
