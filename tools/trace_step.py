#!/usr/bin/env python3
"""One replayed decode step in launch order from a rocprofv3 --kernel-trace directory (development tool):

    python tools/trace_step.py <dir> [marker-substring]

start (us after the step's first kernel), duration, queue, kernel - to read which stream waits for which."""
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
    spans = list(zip(marks, marks[1:]))
    mode = collections.Counter(b - a for a, b in spans).most_common(1)[0][0]
    a, b = [(x, y) for x, y in spans if y - x == mode][-3]
    t0 = int(rows[a]['Start_Timestamp'])
    queues = {}
    for i in range(a, b):
        q = queues.setdefault(rows[i].get('Queue_Id', '?'), len(queues))
        s, e = int(rows[i]['Start_Timestamp']), int(rows[i]['End_Timestamp'])
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q}  {'    ' * q}{names[i][:40]}")
    print(f"step {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")


if __name__ == '__main__':
    main()
