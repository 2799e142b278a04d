#!/usr/bin/env python3
"""When the work items of ONE stage-1 launch start and end (experiments build's debug buffer, as tools/np_phases.py), by chunk length:
   WL=northstar_4kx32 python tools/experiments/np_timeline.py BRANCH_LEN [BRANCH_LEN ...]      (us from the launch's first item start)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
os.environ.setdefault("DEFT_AMD_LIB", os.path.join(ROOT, "deft_amd", "lib", "libdeft_amd_exp.so"))
import numpy as np, torch
from bench import Bench
from deft_amd._lib import lib
from deft_amd.utils.workloads import WORKLOADS, Workload
w0 = WORKLOADS[os.environ.get("WL", "northstar_4kx32")]
lib.deft_debug_set_buffer.argtypes = [ctypes.c_void_p]
NW = 8192
for bl in [int(x) for x in sys.argv[1:]]:
    w = Workload(**{**w0.__dict__, "branch_len": bl})
    b = Bench(w, 8, torch.device("cuda", 0)); b.prepare(use_graph=False)
    for rep in range(4):
        dbg = torch.zeros(2 * NW * 8 + 8, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        lib.deft_debug_set_buffer(dbg.data_ptr())
        l = rep % b.layers
        b.attn[l](b.q[l], b.k_new[l], b.v_new[l], b.meta)
        torch.cuda.synchronize()
        lib.deft_debug_set_buffer(None)
        if rep < 3: continue
        a = dbg.cpu().numpy()
        d = a[: NW * 8].reshape(NW, 8)
        ok = d[:, 3] > 0
        n, t0, t1 = d[ok, 4], d[ok, 1] / 100.0, d[ok, 2] / 100.0
        base = t0.min()
        t0, t1 = t0 - base, t1 - base
        print(f"L={bl}: {int(ok.sum())} work items, last end {t1.max():.1f} us")
        for nn in sorted(set(n.tolist())):
            m = n == nn
            q = lambda x: "%5.1f %5.1f %5.1f" % (np.min(x), np.median(x), np.max(x))
            print(f"   n={int(nn)}: {int(m.sum()):4d} items   start min/med/max {q(t0[m])}   end {q(t1[m])}   duration {q(t1[m] - t0[m])}")
        xcc = (d[ok, 5] >> 32) & 0xff
        ws, we = d[ok, 0] / 100.0 - base, d[ok, 3] / 100.0 - base  # the workgroup's whole stay on the item (prologue .. epilogue)
        print("   per XCD: items / tiles / first start / last end (us) / tile-loop us per tile (median):")
        for x in sorted(set(xcc.tolist())):
            m = xcc == x
            print(f"      XCD {int(x)}: {int(m.sum()):4d} items {int(n[m].sum()):5d} tiles   {ws[m].min():5.1f} .. {we[m].max():5.1f}   "
                  f"{np.median((t1[m] - t0[m]) / n[m]):.2f}   ends of its last 5 items: {' '.join('%.1f' % v for v in np.sort(we[m])[-5:])}")
        # how many items are running at 2-us marks
        marks = np.arange(0, t1.max() + 2, 2.0)
        print("   running at t =", " ".join(f"{int(((t0 <= t) & (t1 > t)).sum())}" for t in marks), "  (every 2 us)")
    del b
    torch.cuda.empty_cache()
