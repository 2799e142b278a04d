#!/usr/bin/env python3
"""Per-workgroup timeline of the tile-parallel stage-1 kernel: start, first K ready, epilogue, end (100 MHz clock)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("DEFT_STAGE1_KERNEL", "np")
from bench import Bench
from deft_amd._lib import lib
from deft_amd.utils.workloads import WORKLOADS, Workload
w0 = WORKLOADS[os.environ.get("WL", "northstar_4kx32")]
bl = int(sys.argv[1]) if len(sys.argv) > 1 else w0.branch_len
w = Workload(**{**w0.__dict__, "branch_len": bl})
b = Bench(w, 8, torch.device("cuda", 0)); b.prepare(use_graph=False)
lib.deft_debug_set_buffer.argtypes = [ctypes.c_void_p]
NW = 8192
for rep in range(3):
    dbg = torch.zeros(NW * 8 + 8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    lib.deft_debug_set_buffer(dbg.data_ptr())
    l = rep % b.layers
    ev0, ev1, ev2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    ev0.record()
    b.attn[l](b.q[l], b.k_new[l], b.v_new[l], b.meta)
    ev1.record()
    torch.cuda.synchronize()
    print(f"rep {rep}: HIP events around the layer call (stage 1 + merge): {ev0.elapsed_time(ev1) * 1e3:.1f} us")
    lib.deft_debug_set_buffer(None)
    if rep == 0: continue
    d = dbg.cpu().numpy()[: NW * 8].reshape(NW, 8)
    ok = d[:, 3] > 0
    d = d[ok]
    t0 = d[:, 0].min()
    st, k0, ep, en, n = [(d[:, i] - t0) / 100.0 for i in range(4)] + [d[:, 4]]
    pct = lambda x: [round(float(np.percentile(x, q)), 2) for q in (0, 10, 50, 90, 100)]
    print(f"rep {rep}: {len(d)} workgroups, span {en.max():.2f} us")
    for nn in sorted(set(n.tolist())):
        m = n == nn
        print(f"  n={int(nn)}: {int(m.sum()):4d} WGs  start {pct(st[m])}  ramp(start->K0) {pct((k0 - st)[m])}  body {pct((ep - k0)[m])}  "
              f"epilogue {pct((en - ep)[m])} (to barrier {pct(((d[:, 7] - t0) / 100.0 - ep)[m])})  total {pct((en - st)[m])}  end {pct(en[m])}")
    # single-launch decode: how long after its KV head's last stage-1 workgroup ended did a merge workgroup see the
    # counter complete, and how long did its merges take
    mg = n == 1000
    if mg.any():
        kvh = d[:, 6]
        head_done = {int(k): float(en[(~mg) & (kvh == k)].max()) for k in set(kvh[~mg].tolist())}
        lag = np.array([wt - head_done.get(int(k), 0.0) for wt, k in zip(ep[mg], kvh[mg])])
        print(f"  merge WGs: saw-complete minus head-done {pct(lag)}  merge duration {pct((en - ep)[mg])}  stage-1 last end {en[~mg].max():.2f}  head-done {pct(np.array(list(head_done.values())))}")
    # occupancy over time: workgroups alive per microsecond
    edges = np.arange(0, en.max() + 1, 2.0)
    alive = [(int(((st <= t) & (en > t)).sum())) for t in edges]
    print("  alive WGs every 2 us:", alive)
    cu = d[:, 5]
    xcc = (cu.astype(np.int64) >> 32) & 0xf
    se = (cu.astype(np.int64) >> 13) & 0x7
    cuid = (cu.astype(np.int64) >> 8) & 0xf
    for nn in sorted(set(n.tolist())):
        m = n == nn
        print(f"  n={int(nn)} total by XCC:", [round(float(np.mean((en - st)[m & (xcc == x)])), 1) if (m & (xcc == x)).any() else None for x in range(8)])
    # per-CU sum of busy time
    key = xcc * 1000 + se * 16 + cuid
    tot = {}
    for k_, t_ in zip(key.tolist(), (en - st).tolist()):
        tot[k_] = tot.get(k_, 0.0) + t_
    v = np.array(list(tot.values()))
    print("  per-CU summed WG time:", pct(v), " CUs:", len(v))
    last = {}
    for k_, t_ in zip(key.tolist(), en.tolist()):
        last[k_] = max(last.get(k_, 0.0), t_)
    print("  per-CU last end:", pct(np.array(list(last.values()))))
    print("  distinct (xcc,hw_id) CU slots:", len(set((int(x) >> 32, (int(x) >> 8) & 0xff, (int(x) >> 13) & 7) for x in cu)))
