#!/usr/bin/env python3
"""In-process A/B of stage-1 launch knobs on the SAME pools (process-to-process placement noise is ~1 us):
   tools/ab.py [--branch-len 200] [--reps 6] [--rounds 3] VAR=a,b,c [VAR2=x,y]   (cartesian product)"""
import argparse, itertools, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import Bench
from deft_amd.utils.workloads import WORKLOADS, Workload, GEOMETRY
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="northstar_4kx32")
ap.add_argument("--branch-len", type=int, default=None)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("knobs", nargs="+")
a = ap.parse_args()
w = WORKLOADS[a.workload]
if a.branch_len is not None:
    w = Workload(**{**w.__dict__, "branch_len": a.branch_len})
b = Bench(w, GEOMETRY[w.model][3], torch.device("cuda", 0)); b.prepare(use_graph=False)
names = [k.split("=")[0] for k in a.knobs]
vals = [k.split("=")[1].split(",") for k in a.knobs]
res = {}
for rnd in range(a.rounds):
    for combo in itertools.product(*vals):
        for n, v in zip(names, combo): os.environ[n] = v
        r = b.time_stage1(reps=a.reps)
        res.setdefault(combo, []).append(r["mean_us"])
for combo, xs in res.items():
    print(" ".join(f"{n}={v}" for n, v in zip(names, combo)), "->", " ".join(f"{x:.2f}" for x in xs), f"| mean {np.mean(xs):.2f} min {np.min(xs):.2f}")
