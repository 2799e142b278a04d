#!/usr/bin/env python3
"""bench.run_level (the few-shot run 1 -> 400 tokens per branch through the session) under the library named by DEFT_AMD_LIB:
   for lib in ...; do DEFT_AMD_LIB=$lib python tools/experiments/run_level_ab.py [workload [window-plans-only]]; done     (same box: an A/B of plan rules at run level)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench as B
w = B.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "northstar_4kx32"]
for inc in ((True, False) if len(sys.argv) < 3 else (True,)):
    for rep in range(2):
        r = B.run_level(w, 32, torch.device("cuda:0"), 400, inc)
        print(os.path.basename(os.environ.get("DEFT_AMD_LIB", "libdeft_amd.so")), "window plans" if inc else "rebuild every step", "rep", rep,
              "run_hbm_frac", r["run_hbm_frac"], "ms_per_step", r["ms_per_step"], r["step_kinds"], flush=True)
        torch.cuda.empty_cache()
