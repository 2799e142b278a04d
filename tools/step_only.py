import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import Bench
from deft_amd.utils.workloads import WORKLOADS, Workload, GEOMETRY
bl = int(sys.argv[1]); mode = sys.argv[2]
w = Workload(**{**WORKLOADS["northstar_4kx32"].__dict__, "branch_len": bl})
b = Bench(w, 32, torch.device("cuda", 0)); b.prepare(use_graph=(mode == "step"))
if mode == "step":
    for _ in range(20): b.step()
else:
    b.time_stage1(reps=10)
torch.cuda.synchronize()
