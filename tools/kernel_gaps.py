#!/usr/bin/env python3
"""Kernel durations AND the gaps between consecutive kernels from a rocprofv3 --kernel-trace CSV (the stats summary only has
durations): tools/kernel_gaps.py <dir or *_kernel_trace.csv> [tail_fraction]
For every kernel name in the last `tail_fraction` of the trace (the timed, graph-replayed steps): count, mean duration, mean gap
from the END of the previous kernel to its start, and mean start-to-start period."""
import csv, glob, os, sys
from collections import defaultdict
src = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[int(len(rows) * (1 - frac)):]
dur, gap, n = defaultdict(float), defaultdict(float), defaultdict(int)
for (s0, e0, _), (s1, e1, k1) in zip(rows, rows[1:]):
    k = k1.split("(")[0][-60:]
    dur[k] += e1 - s1; gap[k] += s1 - e0; n[k] += 1
big = [(s1 - e0, k1.split("(")[0][-50:]) for (s0, e0, _), (s1, e1, k1) in zip(rows, rows[1:]) if s1 - e0 > 3000]
by = defaultdict(list)
for g, k in big: by[k].append(g)
for k, v in by.items():
    print(f"gaps > 3 us in front of {k}: {len(v)}, mean {sum(v) / len(v) / 1e3:.2f} us")
tot = rows[-1][1] - rows[0][0]
print(f"{len(rows)} kernels over {tot / 1e3:.1f} us")
for k in sorted(n, key=lambda k: -dur[k]):
    print(f"{n[k]:7d}  dur {dur[k] / n[k] / 1e3:7.2f} us  gap-before {gap[k] / n[k] / 1e3:7.2f} us   {k}")
