#!/usr/bin/env python3
"""Where the head kernels of a decode step (tree_md_scan .. qrows_fused) run relative to the layers, from a kernel trace
(rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/replay.py ... --pipelined --no-warmup): the gap between two
steps' layers, the head's position in it, mean kernel durations.  Written for round 4's two-stream session
(profiles/r4_two_stream_session_negative.txt); on the shipped one-graph step the head sits inside the gap.
   tools/trace_overlap.py DIR"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
body = [(s, e) for s, e, n in rows if "stage1_np" in n or "merge_kernel" in n]
head = [(s, e, n) for s, e, n in rows if "tree_md_scan" in n or "qrows_fused" in n]
# step boundaries of the body: gaps > 8 us between consecutive layer kernels
gaps = [(body[i][1], body[i + 1][0]) for i in range(len(body) - 1) if body[i + 1][0] - body[i][1] > 8000]
scans = [h for h in head if "tree_md_scan" in h[2]]
qrows = [h for h in head if "qrows_fused" in h[2]]
print(f"{len(body)} layer kernels, {len(gaps)} gaps > 8 us between them, {len(scans)} heads")
import statistics as st
glen = [(b - a) / 1000 for a, b in gaps]
print("gap between two steps' layers, us: median %.1f  p10 %.1f  p90 %.1f" % (st.median(glen), sorted(glen)[len(glen) // 10], sorted(glen)[9 * len(glen) // 10]))
# for each gap: when did the head that precedes the next body start and end, relative to the gap start
rel = []
for a, b in gaps[5:-5]:
    cands = [(s, e) for (s, e, n) in qrows if e <= b + 1000]
    if not cands: continue
    qs, qe = max(cands, key=lambda x: x[1])
    sc = max([(s, e) for (s, e, n) in scans if s <= qs], key=lambda x: x[0])
    rel.append(((sc[0] - a) / 1000, (qe - a) / 1000, (b - qe) / 1000))
if rel:
    print("head start - end of the previous step's layers, us: median %.1f" % st.median(r[0] for r in rel))
    print("head end   - end of the previous step's layers, us: median %.1f" % st.median(r[1] for r in rel))
    print("next step's first layer - head end, us:              median %.1f" % st.median(r[2] for r in rel))
s1 = [(e - s) / 1000 for s, e, n in rows if "stage1_np" in n]
mg = [(e - s) / 1000 for s, e, n in rows if "merge_kernel" in n]
inner = [(body[i + 1][0] - body[i][1]) / 1000 for i in range(len(body) - 1) if body[i + 1][0] - body[i][1] <= 8000]
print("stage 1 mean %.2f us, merge mean %.2f us, gaps inside a step: mean %.2f us (%d), sum per step %.1f us" % (st.mean(s1), st.mean(mg), st.mean(inner), len(inner), sum(inner) / max(len(scans), 1)))
print("span of the trace per step: %.1f us" % ((body[-1][1] - body[0][0]) / 1000 / max(len(scans), 1)))
