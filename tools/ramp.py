#!/usr/bin/env python3
"""Per-workgroup timeline of the streaming stage-1 kernel (DB variant, 256 workgroups): start spread, ramp
(start -> tile 0 ready), per-tile period, tail.  s_memtime ticks are converted with the measured 2.39 GHz... the
ratio is re-derived here from the two clocks each workgroup stamps at start and end."""
import ctypes, os, sys, json
os.environ.setdefault("DEFT_STAGE1_KERNEL", "stream")  # this tool reads the streaming form's stamps
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import Bench
from deft_amd._lib import lib
from deft_amd.utils.workloads import WORKLOADS, Workload
bl = int(sys.argv[1]) if len(sys.argv) > 1 else 200
w = Workload(**{**WORKLOADS["northstar_4kx32"].__dict__, "branch_len": bl})
b = Bench(w, 8, torch.device("cuda", 0)); b.prepare(use_graph=True)
workers = 256
lib.deft_debug_set_buffer.argtypes = [ctypes.c_void_p]
for rep in range(4):
    dbg = torch.zeros(65536 + 8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    lib.deft_debug_set_buffer(dbg.data_ptr())
    l = rep % b.layers
    b.attn[l](b.q[l], b.k_new[l], b.v_new[l], b.meta)
    torch.cuda.synchronize()
    lib.deft_debug_set_buffer(None)
    d = dbg.cpu().numpy()[: workers * 128].reshape(workers, 16, 8)
    if rep == 0: continue
    ws, we = d[:, 15, 6].astype(np.float64), d[:, 15, 7].astype(np.float64)   # 100 MHz
    ms, me = d[:, 15, 0].astype(np.float64), d[:, 15, 1].astype(np.float64)   # s_memtime
    ok = (ws > 0) & (we > 0)
    ratio = np.median((me[ok] - ms[ok]) / ((we[ok] - ws[ok]) * 10.0))          # ticks per ns
    t0 = ws[ok].min()
    start = (ws - t0) / 100.0
    end = (we - t0) / 100.0
    tus = lambda x: x / ratio / 1000.0
    a0 = tus(d[:, 0, 1] - ms)                      # start -> barrier A of tile 0 passed
    print(f"rep {rep}: ticks/ns {ratio:.3f}  kernel span {end[ok].max():.2f} us")
    pct = lambda x: [round(float(np.percentile(x, q)), 2) for q in (0, 10, 50, 90, 100)]
    print("  WG start (us after first):", pct(start[ok]))
    print("  ramp start->tile0 ready   :", pct(a0[ok]))
    print("  WG end                    :", pct(end[ok]))
    # per-tile period: A(i+1) - A(i)
    for i in range(0, 11):
        v = (d[:, i + 1, 1] > 0) & ok
        if v.sum() == 0: break
        per = tus(d[v, i + 1, 1] - d[v, i, 1])
        ph = [tus(d[v, i, k + 1] - d[v, i, k]) for k in range(7)]
        print(f"  tile {i:2d}: n={int(v.sum()):3d} period {np.mean(per):5.2f} us   phases "
              + " ".join(f"{np.mean(x):4.2f}" for x in ph))
    lastA = np.array([d[j, :15, 7].max() for j in range(workers)], dtype=np.float64)
    print("  last H -> WG end:", pct(tus(me[ok] - lastA[ok])))
