import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from bench import Bench, run_timed
from deft_amd.utils.workloads import WORKLOADS, GEOMETRY
w = WORKLOADS["northstar_4kx32"]
b = Bench(w, 32, torch.device("cuda", 0)); b.prepare(use_graph=True)
print("A after prepare(graph):", b.end_to_end(8)["ms_per_step"])
run_timed(b, 50, 5, False)
print("B after run_timed:", b.end_to_end(8)["ms_per_step"])
b.time_stage1(reps=2)
print("C after time_stage1:", b.end_to_end(8)["ms_per_step"])
b.step_percentiles(50)
print("D after percentiles:", b.end_to_end(8)["ms_per_step"])
b.time_plan()
print("E after time_plan:", b.end_to_end(8)["ms_per_step"])
