#!/usr/bin/env python3
"""Hunt for the intermittent slow run: the few-shot run (bench.run_level's loop) repeated; host time stamps per step (the host is at most
8 steps ahead of the GPU: the staging ring), no events in the stream.  Prints every run's ms per step and, for runs slower than 1.4 x the
median, where the time went (per 10 steps).     python tools/experiments/slow_run_hunt.py WORKLOAD GEN REPS [win_tiles]"""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench as B
import deft_amd
from deft_amd.utils.workloads import Workload
w = B.WORKLOADS[sys.argv[1]]
gen, reps = int(sys.argv[2]), int(sys.argv[3])
W = int(sys.argv[4]) if len(sys.argv) > 4 else None
dev = torch.device("cuda:0")
runs = []
if os.environ.get("NOGC"): gc.disable()
for rep in range(reps):
    b = B.Bench(Workload(**{**w.__dict__, "branch_len": 1}), 32, dev, seed=11, extra_steps=gen + 8)
    tree = b.forest.trees[0]
    sess = deft_amd.FlattenDecodeSession(tree, b.Hq, b.Hkv, b.D, 32, lambda l: (b.q[l], b.k_new[l], b.v_new[l]), win_tiles=W)
    torch.cuda.synchronize(dev)
    ts = [time.perf_counter()]
    for _ in range(gen - 1):
        for leaf in tree.leaves.values():
            leaf.append_token(7)
        sess.step()
        ts.append(time.perf_counter())
    torch.cuda.synchronize(dev)
    ts.append(time.perf_counter())
    runs.append((ts[-1] - ts[0], ts, dict(sess.step_kinds)))
    print(f"rep {rep}: {(ts[-1] - ts[0]) / (gen - 1) * 1e3:.4f} ms per step", flush=True)
    del sess, b, tree
    torch.cuda.empty_cache()
med = sorted(r[0] for r in runs)[len(runs) // 2]
for i, (tot, ts, kinds) in enumerate(runs):
    if tot > 1.4 * med:
        d = [round((ts[min(k + 10, len(ts) - 1)] - ts[k]) * 1e3, 2) for k in range(0, len(ts) - 1, 10)]
        print(f"SLOW rep {i}: {tot * 1e3:.1f} ms against a median of {med * 1e3:.1f}; ms per 10 steps (the last entry includes the final sync): {d}")
print("median ms per run", round(med * 1e3, 2), "slow runs", sum(1 for r in runs if r[0] > 1.4 * med), "of", len(runs))
