#!/usr/bin/env python3
"""DecodeSession.step() host time by part (synchronised loop), without a profiler: perf_counter around the pieces.
   python tools/experiments/step_host_parts.py [width] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import deft_amd
from deft_amd import session as S

width = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
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
acc = {}


def timed(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **kw):
        t = time.perf_counter()
        r = f(*a, **kw)
        acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, name, g)


timed(pool, "alloc_host", "pool.alloc_host")
timed(sess, "_stage", "_stage (ring slot, books, copy)")
timed(sess, "_staged", "_staged (event every 4th)")
timed(sess, "_moved", "_moved")
for _ in range(10):
    for lf in leaves:
        lf.append_token(7)
    sess.step(); torch.cuda.synchronize()
for kind in ("patch", "replan"):
    g = sess.graphs[kind]
    class G:  # (CUDAGraph.replay is a slot of a C type: wrap the object)
        def __init__(s, g): s.g = g
        def replay(s):
            t = time.perf_counter(); s.g.replay(); acc["graph.replay"] = acc.get("graph.replay", 0.0) + time.perf_counter() - t
    sess.graphs[kind] = G(g)
acc.clear()
tot = 0.0
per = []
for i in range(steps):
    for lf in leaves:
        lf.append_token(7)
    before = dict(acc)
    t = time.perf_counter()
    sess.step()
    dt = time.perf_counter() - t
    tot += dt
    per.append((dt, i, {kk: round((acc[kk] - before.get(kk, 0.0)) * 1e6, 1) for kk in acc}))
    torch.cuda.synchronize()
for dt, i, parts in sorted(per, reverse=True)[:4]:
    print(f"   slowest: step {i} {dt * 1e6:.1f} us", parts)
med = sorted(x[0] for x in per)[len(per) // 2]
print(f"   median step {med * 1e6:.1f} us")
print(f"width {width}: sess.step() {tot / steps * 1e6:.1f} us of host time per step", sess.step_kinds)
for kk, vv in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"   {vv / steps * 1e6:6.1f} us  {kk}")
print(f"   {(tot - sum(acc.values())) / steps * 1e6:6.1f} us  the rest of step() (alloc_step, journal_take, ctypes, Python)")
