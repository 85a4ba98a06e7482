#!/bin/bash
# A/B of the device-resident H2O bookkeeping (SVK_H2O_DEVICE_STATE) over batch sizes: ms per step of bench.py
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for B in ${@:-1 8 64 256}; do
  for m in 1 0; do
    SVK_H2O_DEVICE_STATE=$m python bench.py --batch $B --no-cpu-baseline --no-paths --no-kernel-events 2>/dev/null > /tmp/ab_$B_$m.json
    python - "$B" "$m" /tmp/ab_$B_$m.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
b = d.get("burst") or {}
print(f"B={sys.argv[1]:>3} device_state={sys.argv[2]}: {d['ms_per_step']:.4f} ms/step  {d['value']:.0f} tok/s  burst {b.get('ms', float('nan')):.3f} ms")
PY
  done
done
