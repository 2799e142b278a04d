#!/usr/bin/env python3
"""Time the Flatten stage-1 kernel (and the whole step) for a workload; used to
A/B kernel variants on the GPU box.  Prints one JSON line."""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import Bench, run_timed, HBM_PEAK_GBPS
from deft_amd.utils.workloads import WORKLOADS, Workload, GEOMETRY

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="northstar_4kx32")
ap.add_argument("--branch-len", type=int, nargs="*", default=[None])
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--tag", default="")
args = ap.parse_args()
dev = torch.device("cuda", 0)
out = {"tag": args.tag, "ablate": os.environ.get("DEFT_STAGE1_ABLATE", "0"), "variant": os.environ.get("DEFT_STAGE1_VARIANT", "")}
for bl in args.branch_len:
    w = WORKLOADS[args.workload]
    if bl is not None:
        w = Workload(**{**w.__dict__, "branch_len": bl})
    bl = w.branch_len
    b = Bench(w, GEOMETRY[w.model][3], dev)
    b.prepare(use_graph=True)
    dt = run_timed(b, args.steps, 5, False)
    s1 = b.time_stage1(reps=2)
    algo = b.algorithmic_bytes_per_layer()
    out[f"len{bl}"] = {"step_us_per_layer": round(dt / args.steps / b.layers * 1e6, 2),
                       "stage1_us": round(s1["mean_us"], 2), "stage1_TBps": round(algo / s1["mean_us"] / 1e6, 3),
                       "stage1_frac": round(algo / s1["mean_us"] / 1e3 / HBM_PEAK_GBPS, 4), "MB": round(algo / 1e6, 1)}
    del b
    torch.cuda.empty_cache()
print(json.dumps(out), flush=True)
