#!/usr/bin/env python3
"""GPU timeline of the graphed advancing-tree decode loop (bench.py end_to_end, graphed).

  run:      python tools/e2e_timeline.py run [workload]            (under rocprofv3 --kernel-trace --output-format csv)
  analyse:  python tools/e2e_timeline.py csv <kernel_trace.csv>    per step: span, busy time, the gaps and the kernels around them
"""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def run(name):
    import torch
    from bench import Bench
    from deft_amd.utils.workloads import WORKLOADS, GEOMETRY
    w = WORKLOADS[name]
    b = Bench(w, GEOMETRY[w.model][3], torch.device("cuda", 0)); b.prepare(use_graph=False)
    if w.trees > 1:  # a batch of trees: as ONE tree object through the session (bench.py forest_end_to_end)
        from bench import forest_end_to_end
        print(forest_end_to_end(b, w, torch.device("cuda", 0), 30))
    else:
        print(b.end_to_end(30, True))


def analyse(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    short = lambda n: n.split("(")[0].replace("deft::", "").replace("void ", "")[:40]
    starts = [i for i, r in enumerate(rows) if "index_elementwise" in r[2] or "index_put" in r[2]]  # the page-table write opens a step's layers
    print("kernels", len(rows), "steps", len(starts))
    for a, b in list(zip(starts, starts[1:]))[-6:]:
        seg = rows[a:b]
        span = (rows[b][0] - seg[0][0]) / 1e3
        busy = sum(e - s for s, e, _ in seg) / 1e3
        print(f"step: span {span:8.1f} us  busy {busy:8.1f}  kernels {len(seg)}")
        big = []
        for i in range(len(seg)):
            nxt = seg[i + 1][0] if i + 1 < len(seg) else rows[b][0]
            gap = (nxt - seg[i][1]) / 1e3
            if gap > 2.5:
                big.append((gap, short(seg[i][2]), short(seg[i + 1][2]) if i + 1 < len(seg) else "next step"))
        tot = sum(g for g, _, _ in big)
        print(f"   gaps > 2.5 us: {len(big)}, total {tot:.1f} us")
        for g, x, y in sorted(big, reverse=True)[:8]:
            print(f"     {g:7.1f} us after {x} before {y}")
        # kernels that ran beside another one (a later start before an earlier end): the prepared step under the layers
        over = 0.0
        end_max = seg[0][1]
        for s_, e_, n_ in seg[1:]:
            if s_ < end_max:
                over += (min(e_, end_max) - s_) / 1e3
            end_max = max(end_max, e_)
        print(f"   overlapped kernel time {over:.1f} us")
        names = {}
        for s, e, n in seg:
            k = short(n); names[k] = names.get(k, 0) + (e - s) / 1e3
        print("   " + ", ".join(f"{k} {v:.1f}" for k, v in sorted(names.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "northstar_4kx32")
    else:
        analyse(sys.argv[2])
