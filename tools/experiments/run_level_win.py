#!/usr/bin/env python3
"""bench.run_level (a few-shot tree growing 1 -> GEN tokens per branch through the session) by overflow tiles per window (win_tiles;
0 = rebuild every step), under the library named by DEFT_AMD_LIB:   python tools/experiments/run_level_win.py WORKLOAD GEN W [W ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench as B
w = B.WORKLOADS[sys.argv[1]]
gen = int(sys.argv[2])
for rep in range(2):
    for W in [int(x) for x in sys.argv[3:]]:
        r = B.run_level(w, 32, torch.device("cuda:0"), gen, W != 0, W or None)
        print(os.path.basename(os.environ.get("DEFT_AMD_LIB", "libdeft_amd.so")), sys.argv[1], "gen", gen, "win_tiles", W, "rep", rep,
              "run_hbm_frac", r["run_hbm_frac"], "ms_per_step", r["ms_per_step"], r["step_kinds"], flush=True)
        torch.cuda.empty_cache()
