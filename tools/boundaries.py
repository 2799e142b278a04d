#!/usr/bin/env python3
"""Kernel-boundary anatomy of one Flatten call on the 100 MHz device clock:
plan kernel | gap | stream kernel (first WG start .. last WG end) | gap | merge kernel."""
import ctypes, os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import Bench
from deft_amd._lib import lib
from deft_amd.utils.workloads import WORKLOADS, Workload, GEOMETRY
import deft_amd
bl = int(sys.argv[1]) if len(sys.argv) > 1 else 200
w = Workload(**{**WORKLOADS["northstar_4kx32"].__dict__, "branch_len": bl})
b = Bench(w, 8, torch.device("cuda", 0)); b.prepare(use_graph=True)
workers = 512
lib.deft_debug_set_buffer.argtypes = [ctypes.c_void_p]
res = []
for rep in range(5):
    dbg = torch.zeros(workers * 128 + 8, dtype=torch.int64, device="cuda")
    dbg[65536] = 2**62; dbg[65538] = 2**62
    torch.cuda.synchronize()
    lib.deft_debug_set_buffer(dbg.data_ptr())
    l = rep % b.layers
    o = b.attn[l](b.q[l], b.k_new[l], b.v_new[l], b.meta)   # kv_append + plan + stream + merge, eager
    torch.cuda.synchronize()
    lib.deft_debug_set_buffer(None)
    raw = dbg.cpu().numpy()
    d = raw[: workers * 128].reshape(workers, 16, 8)
    st, en = d[:, 15, 6], d[:, 15, 7]
    ok = st > 0
    t0 = raw[65536]
    us = lambda x: round((int(x) - int(t0)) / 100.0, 2)
    res.append({"plan": [0.0, us(raw[65537])], "stream_first_start": us(st[ok].min()), "stream_last_start": us(st[ok].max()),
                "stream_first_end": us(en[ok].min()), "stream_median_end": us(np.median(en[ok])), "stream_last_end": us(en[ok].max()),
                "merge": [us(raw[65538]), us(raw[65539])]})
for r in res: print(json.dumps(r))
# per-position breakdown of the last rep: workgroup j covers units [5.125 j, ...) of head j // 16
dur = (en - st) / 100.0
pos = np.arange(workers) % 16
print("mean WG duration (us) by position within a head (0 = prefix start .. 15 = last leaf tiles):")
print([round(float(dur[pos == k].mean()), 1) for k in range(16)])
print("by head quartile:", [round(float(dur[(np.arange(workers) // 16) // 8 == q].mean()), 1) for q in range(4)])
print("by XCD (j % 8):", [round(float(dur[np.arange(workers) % 8 == x].mean()), 1) for x in range(8)])
print("by tiles (first 64 WGs have 6):", round(float(dur[:64].mean()), 1), round(float(dur[64:].mean()), 1))
print("sorted deciles:", [round(float(np.percentile(dur, q)), 1) for q in range(0, 101, 10)])
