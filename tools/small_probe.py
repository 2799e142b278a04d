import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import Bench
from deft_amd.utils.workloads import WORKLOADS, Workload, GEOMETRY
for prefix, width, bl in ((256, 2, 1), (1024, 32, 1), (4096, 32, 1)):
    w = Workload("x", "llama2-7b", "flatten", "few_shot", prefix, width, bl)
    b = Bench(w, 4, torch.device("cuda", 0)); b.prepare(use_graph=False)
    for _ in range(5): b.step_eager()
    torch.cuda.synchronize()
    print(prefix, width, bl, b.time_stage1(reps=2))
