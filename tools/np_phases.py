#!/usr/bin/env python3
"""Where a stage-1 workgroup's tile loop spends its time (experiments build): per-phase wall-clock sums of wave 0 over the tiles
of every work item -- wait K(i) | masks + QK^T + maxima | next tile's offsets + K(i+1) / aux requests | softmax | wait V(i) |
PV + V(i+1) request -- printed per chunk length as us PER TILE (median over the items).  WL=<workload> [DEFT_STAGE1_ABLATE=..]."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import Bench
from deft_amd._lib import lib
from deft_amd.utils.workloads import WORKLOADS, Workload
w0 = WORKLOADS[os.environ.get("WL", "gqa_4kx32")]
bl = int(sys.argv[1]) if len(sys.argv) > 1 else w0.branch_len
w = Workload(**{**w0.__dict__, "branch_len": bl})
b = Bench(w, 8, torch.device("cuda", 0)); b.prepare(use_graph=False)
lib.deft_debug_set_buffer.argtypes = [ctypes.c_void_p]
NW = 8192
names = ["wait K", "masks+QK^T+max", "offsets+K/aux req", "softmax", "wait V", "PV+V req"]
for rep in range(3):
    dbg = torch.zeros(2 * NW * 8 + 8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    lib.deft_debug_set_buffer(dbg.data_ptr())
    l = rep % b.layers
    b.attn[l](b.q[l], b.k_new[l], b.v_new[l], b.meta)
    torch.cuda.synchronize()
    lib.deft_debug_set_buffer(None)
    if rep == 0: continue
    a = dbg.cpu().numpy()
    d, ph = a[: NW * 8].reshape(NW, 8), a[NW * 8 : 2 * NW * 8].reshape(NW, 8)
    ok = d[:, 3] > 0
    n = d[ok, 4]
    ph = ph[ok, :6] / 100.0  # us
    body = (d[ok, 2] - d[ok, 1]) / 100.0
    print(f"rep {rep}: {int(ok.sum())} work items ({w.name})")
    for nn in sorted(set(n.tolist())):
        m = n == nn
        per = np.median(ph[m] / nn, axis=0)
        print(f"  n={int(nn)}: {int(m.sum()):4d} items  us per tile: " + "  ".join(f"{nm} {v:.2f}" for nm, v in zip(names, per)) +
              f"  | sum {per.sum():.2f}  (body/n {np.median(body[m]) / nn:.2f})")
