import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import Bench
from deft_amd.utils.workloads import WORKLOADS, Workload, GEOMETRY
GEOMETRY["g4"] = (32, 8, 128, 32); GEOMETRY["g2"] = (16, 8, 128, 32); GEOMETRY["g1"] = (8, 8, 128, 32)
GEOMETRY["m32"] = (32, 32, 128, 32)
def run(name, w):
    b = Bench(w, 32, torch.device("cuda", 0)); b.prepare(True)
    for _ in range(10): b.step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): b.step()
    e1.record(); torch.cuda.synchronize()
    s1 = b.time_stage1(5)
    print(name, "step us/layer %.2f" % (e0.elapsed_time(e1) * 1e3 / (50 * 32)), "stage1", s1 and round(s1["median_us"], 2), flush=True)
    del b; torch.cuda.empty_cache()
for g in ("g4", "g2", "g1"):
    run("tot50 " + g, Workload("x", g, "flatten", "tot", 4096))
    run("ns4kx32 " + g, Workload("x", g, "flatten", "few_shot", 4096, 32, 200))
for wd in (64, 32):
    run("medusa node w%d" % wd, Workload("x", "m32", "node", "medusa", 1016, wd, 1))
    run("medusa flatten w%d" % wd, Workload("x", "m32", "flatten", "medusa", 1016, wd, 1))
