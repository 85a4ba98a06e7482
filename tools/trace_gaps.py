#!/usr/bin/env python3
"""Per-node timeline of a replayed decode step from a rocprofv3 --kernel-trace output directory (development tool).

    python tools/trace_gaps.py <dir> [marker-substring]

The steps are the stretches between consecutive launches of the step's first kernel (the `*_device_begin_kernel` of the
device-resident managers by default) that have the most common launch count; per kernel name: launches per step, mean
duration, mean idle gap in front of it; then the step's span, the sum of its kernel durations and of its gaps."""
import collections
import csv
import glob
import sys


def main():
    f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    names = [r['Kernel_Name'].replace('void ', '').replace('svk::(anonymous namespace)::', '') for r in rows]
    marker = sys.argv[2] if len(sys.argv) > 2 else 'device_begin_kernel'
    marks = [i for i, n in enumerate(names) if marker in n]
    if len(marks) < 3:
        print(f"no '{marker}' launches: pass the step's first kernel as the second argument")
        return
    spans = [(a, b) for a, b in zip(marks, marks[1:])]
    mode = collections.Counter(b - a for a, b in spans).most_common(1)[0][0]
    steps = [(a, b) for a, b in spans if b - a == mode][-24:]
    dur, gap, cnt = collections.defaultdict(list), collections.defaultdict(list), collections.Counter()
    span_us, busy_us, idle_us = [], [], []
    for a, b in steps:
        prev_end, t_busy, t_idle = None, 0.0, 0.0
        for i in range(a, b):
            s, e = int(rows[i]['Start_Timestamp']), int(rows[i]['End_Timestamp'])
            n = names[i][:44]
            dur[n].append((e - s) / 1e3)
            cnt[n] += 1
            t_busy += (e - s) / 1e3
            if prev_end is not None:
                gap[n].append((s - prev_end) / 1e3)
                t_idle += max(0.0, (s - prev_end) / 1e3)
            prev_end = max(e, prev_end or e)
        span_us.append((prev_end - int(rows[a]['Start_Timestamp'])) / 1e3)
        busy_us.append(t_busy)
        idle_us.append(t_idle)
    print(f"{len(steps)} steps of {mode} launches each")
    order = []
    for i in range(steps[-1][0], steps[-1][1]):
        if names[i][:44] not in order:
            order.append(names[i][:44])
    for n in order:
        g = gap[n]
        print(f"  {n:46s} x{cnt[n] / len(steps):5.1f}  dur {sum(dur[n]) / len(dur[n]):7.2f} us  gap before {sum(g) / max(1, len(g)):6.2f} us")
    k = len(steps)
    print(f"  step: first start -> last end {sum(span_us) / k:8.1f} us, kernel durations {sum(busy_us) / k:8.1f} us, idle gaps {sum(idle_us) / k:7.1f} us"
          " (overlapping streams count twice in the durations)")


if __name__ == '__main__':
    main()
