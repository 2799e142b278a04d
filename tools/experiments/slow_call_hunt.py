#!/usr/bin/env python3
"""Which call of an epoch's set-up stalls on the boxes where runs are intermittently slow (profiles/r6_slow_run_hunt.txt): perf_counter
around DecodeSession._epoch_setup / _capture / graph replays / _stage / torch.empty / Tensor.pin_memory, every call over 5 ms printed.
   python tools/experiments/slow_call_hunt.py WORKLOAD GEN REPS"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench as B
import deft_amd
from deft_amd import session as S, tree_cache as TC
from deft_amd.utils.workloads import Workload
import gc
if os.environ.get('NOGC'): gc.disable()
gc_t = [0.0]
def _gc_cb(phase, info):
    if phase == 'start': gc_t[0] = time.perf_counter()
    else:
        d = time.perf_counter() - gc_t[0]
        if d > 2e-3: slow.append((f"GC gen {info['generation']}", round(d * 1e3, 2)))
gc.callbacks.append(_gc_cb)
slow = []
THRESH = float(os.environ.get('THRESH_MS', '5')) * 1e-3
depth = [0]


def timed(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **kw):
        t = time.perf_counter()
        try:
            return f(*a, **kw)
        finally:
            dt = time.perf_counter() - t
            if dt > THRESH:
                slow.append((label, round(dt * 1e3, 2)))
    setattr(obj, name, g)


for n in ("_epoch_setup", "_capture", "_stage", "_launch_step", "_launch_window_step", "_staged"):
    timed(S.DecodeSession, n, "DecodeSession." + n)
timed(TC._DeviceTree, "sync", "_DeviceTree.sync")
timed(TC._DeviceTree, "__init__", "_DeviceTree.__init__")
timed(torch, "empty", "torch.empty")
timed(torch, "zeros", "torch.zeros")
timed(torch.cuda.CUDAGraph, "replay", "CUDAGraph.replay")
timed(torch.cuda.CUDAGraph, "capture_begin", "CUDAGraph.capture_begin")
timed(torch.cuda.CUDAGraph, "capture_end", "CUDAGraph.capture_end")
timed(torch.Tensor, "pin_memory", "Tensor.pin_memory")
for n in ("deft_window_create", "deft_window_step", "deft_stage_copy", "deft_tree_alloc_step", "deft_tree_journal_take"):
    timed(deft_amd.lib, n, n)
w = B.WORKLOADS[sys.argv[1]]
gen, reps = int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
tot = []
for rep in range(reps):
    b = B.Bench(Workload(**{**w.__dict__, "branch_len": 1}), 32, dev, seed=11, extra_steps=gen + 8)
    tree = b.forest.trees[0]
    sess = deft_amd.FlattenDecodeSession(tree, b.Hq, b.Hkv, b.D, 32, lambda l: (b.q[l], b.k_new[l], b.v_new[l]))
    torch.cuda.synchronize(dev)
    slow.clear()
    t0 = time.perf_counter()
    for i in range(gen - 1):
        for leaf in tree.leaves.values():
            leaf.append_token(7)
        t = time.perf_counter()
        sess.step()
        dt = time.perf_counter() - t
        if dt > THRESH:
            slow.append((f"step {i} ({sess.step_kinds})", round(dt * 1e3, 1)))
    torch.cuda.synchronize(dev)
    tot.append(time.perf_counter() - t0)
    if slow:
        print(f"rep {rep}: {tot[-1] * 1e3:.1f} ms; calls over 5 ms:", slow, flush=True)
    del sess, b, tree
    torch.cuda.empty_cache()
print("median ms per run", round(sorted(tot)[len(tot) // 2] * 1e3, 2), "max", round(max(tot) * 1e3, 2))
