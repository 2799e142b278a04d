#!/usr/bin/env python3
"""In-process A/B of plan / launch knobs on the WHOLE layer step (stage 1 + merge, any mode), hipGraph replay on the
same pools:   tools/ab_step.py --workload medusa64_node [--steps 60] VAR=a,b,c [VAR2=x,y]   (cartesian product)"""
import argparse, itertools, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import Bench
from deft_amd.utils.workloads import WORKLOADS, Workload, GEOMETRY
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="medusa64_node")
ap.add_argument("--mode", default=None, help="override the workload's mode (flatten / node / seq)")
ap.add_argument("--width", type=int, default=None, help="override the workload's branch count")
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("knobs", nargs="+")
a = ap.parse_args()
w = WORKLOADS[a.workload]
if a.mode:
    w = Workload(**{**w.__dict__, "mode": a.mode})
if a.width:
    w = Workload(**{**w.__dict__, "width": a.width})
b = Bench(w, GEOMETRY[w.model][3], torch.device("cuda", 0))
names = [k.split("=")[0] for k in a.knobs]
vals = [k.split("=")[1].split(",") for k in a.knobs]
res = {}
for rnd in range(a.rounds):
    for combo in itertools.product(*vals):
        for n, v in zip(names, combo):
            if v == "-": os.environ.pop(n, None)
            else: os.environ[n] = v
        b.graph = None
        b.md.__dict__.pop("_deft_step", None)  # the per-step fast path holds the plan it was built with
        b.prepare(use_graph=True)  # plans are cached per deft_plan_variant(): a new knob value rebuilds them
        for _ in range(10): b.step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps): b.step()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(combo, []).append(e0.elapsed_time(e1) * 1e3 / (a.steps * b.layers))
for combo, xs in res.items():
    print(" ".join(f"{n}={v}" for n, v in zip(names, combo)), "-> us/layer", " ".join(f"{x:.2f}" for x in xs), f"| mean {np.mean(xs):.2f} min {np.min(xs):.2f} ({b.launch})")
