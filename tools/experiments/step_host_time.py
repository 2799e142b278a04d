#!/usr/bin/env python3
"""Where the host time of DecodeSession.step() goes (synchronised loop: the GPU is idle when the step starts).
   python tools/experiments/step_host_time.py [width] [steps]"""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import deft_amd

width = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
Hq, Hkv, D, layers = 32, 32, 128, 32
size = 4096 + width * (steps + 200) + 1024
req = deft_amd.ReqToTokenPool(width + 8, size, device="cuda")
pool = deft_amd.TokenToKVPool(size, torch.float16, Hkv, D, layers, device="cuda")
tree = deft_amd.TreeCache(torch.float16, Hkv, D, layers, req, pool, None, True, False)
tree.init_prompt(torch.arange(1, 4097, dtype=torch.int32))
tree.branch(tree.root, width)
q = torch.randn((layers, width, Hq * D), dtype=torch.float16, device="cuda")
k = torch.randn((layers, width, Hkv * D), dtype=torch.float16, device="cuda")
v = torch.randn((layers, width, Hkv * D), dtype=torch.float16, device="cuda")
sess = deft_amd.DecodeSession(tree, Hq, Hkv, D, layers, lambda l: (q[l], k[l], v[l]))
leaves = list(tree.leaves.values())


def one():
    for lf in leaves:
        lf.append_token(7)
    sess.step()
    torch.cuda.synchronize()


for _ in range(10):
    one()
t_app = t_step = 0.0
for _ in range(steps):
    t0 = time.perf_counter()
    for lf in leaves:
        lf.append_token(7)
    t1 = time.perf_counter()
    sess.step()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t_app += t1 - t0
    t_step += t2 - t1
print(f"width {width}: append_token x {width} {t_app / steps * 1e6:.1f} us, sess.step() {t_step / steps * 1e6:.1f} us (host, GPU idle at entry)", sess.step_kinds)
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    for lf in leaves:
        lf.append_token(7)
    sess.step()
    torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
