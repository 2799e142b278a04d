#!/usr/bin/env python3
"""Kernel-boundary anatomy of one Flatten call on the 100 MHz device clock:
plan kernel | gap | stream kernel (first WG start .. last WG end) | gap | merge kernel."""
import ctypes, os, sys, json
os.environ.setdefault("DEFT_STAGE1_KERNEL", "stream")  # this tool reads the streaming form's stamps
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import Bench
from deft_amd._lib import lib
from deft_amd.utils.workloads import WORKLOADS, Workload, GEOMETRY
import deft_amd
bl = int(sys.argv[1]) if len(sys.argv) > 1 else 200
w = Workload(**{**WORKLOADS["northstar_4kx32"].__dict__, "branch_len": bl})
b = Bench(w, 8, torch.device("cuda", 0)); b.prepare(use_graph=True)
workers = 256 if os.environ.get("DEFT_STREAM_DB") == "1" else 512
lib.deft_debug_set_buffer.argtypes = [ctypes.c_void_p]
res = []
for rep in range(5):
    dbg = torch.zeros(65536 + 8, dtype=torch.int64, device="cuda")
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
hwid = d[:, 15, 5]
xcc = (hwid >> 32) & 0xf
hw = hwid & 0xffffffff
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = xcc * 64 + se * 8 + sh * 16 * 0 + cu  # coarse physical CU key
key = (xcc.astype(np.int64) << 20) | (se.astype(np.int64) << 10) | (sh.astype(np.int64) << 8) | cu.astype(np.int64)
import collections
groups = collections.defaultdict(list)
for j in range(workers):
    if ok[j]: groups[int(key[j])].append(float(dur[j]))
sizes = collections.Counter(len(v) for v in groups.values())
print("distinct CUs used:", len(groups), "WGs per CU histogram:", dict(sizes))
m1 = [v[0] for v in groups.values() if len(v) == 1]
m2 = [np.mean(v) for v in groups.values() if len(v) == 2]
m3 = [np.mean(v) for v in groups.values() if len(v) >= 3]
print("mean dur with 1 WG on CU:", round(float(np.mean(m1)), 1) if m1 else None, " 2 WGs:", round(float(np.mean(m2)), 1) if m2 else None, " >=3:", round(float(np.mean(m3)), 1) if m3 else None)
print("by xcc:", [round(float(dur[(xcc == x) & ok].mean()), 1) for x in range(8)], "counts", [int(((xcc == x) & ok).sum()) for x in range(8)])
