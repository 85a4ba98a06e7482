#!/usr/bin/env python3
"""How the CPU leg of bench.py scales with worker processes on this host (no GPU used)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "usable", bench._usable_cores())
for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try:
        print(path, open(path).read().strip())
    except OSError as e:
        print(path, "-", e)
for w in [int(x) for x in (sys.argv[1:] or ["1", "8", "32", "64", "128", "256"])]:
    r = bench.cpu_baseline(steps=1, workers=w)
    print(json.dumps({"workers": w, "tokens_per_s": round(r["value"], 2), "sample": r["sample"][-90:]}), flush=True)
