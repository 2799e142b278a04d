#!/usr/bin/env python3
"""What happens on the GPU between two decode steps of an advancing loop: from the END of a step's last merge kernel to the START of
the next step's first stage-1 kernel, kernel by kernel (and copy by copy), out of a rocprofv3 trace:

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python tools/replay.py ... --pipelined --no-warmup
    tools/step_boundary.py DIR [boundaries to print]

Prints the mean over the second half of the run -- idle time, every kernel's duration and the gap in front of it -- and a few
boundaries verbatim."""
import csv, glob, os, sys
from collections import defaultdict

src = sys.argv[1]
show = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = []
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]))
for f in glob.glob(os.path.join(src, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
is_s1 = lambda n: "stage1_np_kernel" in n or "stage1_kernel" in n
is_merge = lambda n: "merge_kernel" in n or "merge_coop" in n
bounds = []  # (index of the last merge of a step, index of the next step's first stage 1)
i = 0
while i + 1 < len(ev):
    if is_merge(ev[i][2]) and not is_s1(ev[i + 1][2]):
        j = i + 1
        while j < len(ev) and not is_s1(ev[j][2]):
            j += 1
        if j < len(ev):
            bounds.append((i, j))
        i = j
    else:
        i += 1
bounds = bounds[len(bounds) // 2:]
if not bounds:
    sys.exit("no step boundary found")
tot = idle = 0.0
per = defaultdict(lambda: [0, 0.0, 0.0])
forms = defaultdict(int)
for a, b in bounds:
    tot += ev[b][0] - ev[a][1]
    busy = 0
    prev_end = ev[a][1]
    names = []
    for k in range(a + 1, b):
        p = per[ev[k][2]]
        p[0] += 1
        p[1] += ev[k][1] - ev[k][0]
        p[2] += ev[k][0] - prev_end
        busy += ev[k][1] - ev[k][0]
        prev_end = ev[k][1]
        names.append(ev[k][2].split("::")[-1][:14])
    per["(first stage 1 of the next step)"][0] += 1
    per["(first stage 1 of the next step)"][2] += ev[b][0] - prev_end
    idle += (ev[b][0] - ev[a][1]) - busy
    forms[" > ".join(names)] += 1
n = len(bounds)
print(f"{n} step boundaries (second half of the trace): {tot / n / 1e3:.2f} us from a step's last merge to the next step's first stage 1, "
      f"{idle / n / 1e3:.2f} us of it idle")
for name, (c, d, g) in sorted(per.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
    print(f"  {c / n:5.2f} per step   dur {d / max(c, 1) / 1e3:6.2f} us   gap in front {g / max(c, 1) / 1e3:6.2f} us   {name}")
for form, c in sorted(forms.items(), key=lambda kv: -kv[1]):
    print(f"  {c:4d} x  {form}")
for a, b in bounds[:show]:
    t0 = ev[a][1]
    print("boundary:")
    for k in range(a + 1, b + 1):
        print(f"   +{(ev[k][0] - t0) / 1e3:7.2f} us  dur {(ev[k][1] - ev[k][0]) / 1e3:6.2f}  {ev[k][2]}")
