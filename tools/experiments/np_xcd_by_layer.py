#!/usr/bin/env python3
"""Per-XCD speed of the headline stage-1 launch BY LAYER POOL (experiments build's debug buffer): is it always the odd XCDs that are slow,
whatever pool the K / V rows sit in?      python tools/experiments/np_xcd_by_layer.py [branch_len]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
os.environ.setdefault("DEFT_AMD_LIB", os.path.join(ROOT, "deft_amd", "lib", "libdeft_amd_exp.so"))
import numpy as np, torch
from bench import Bench
from deft_amd._lib import lib
from deft_amd.utils.workloads import WORKLOADS, Workload
w0 = WORKLOADS["northstar_4kx32"]
bl = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib.deft_debug_set_buffer.argtypes = [ctypes.c_void_p]
NW = 8192
b = Bench(Workload(**{**w0.__dict__, "branch_len": bl}), 32, torch.device("cuda", 0)); b.prepare(use_graph=False)
for l in range(32):
    b.attn[l](b.q[l], b.k_new[l], b.v_new[l], b.meta)  # warm
torch.cuda.synchronize()
print("layer: per XCD 0..7 us per tile (median over its items) | last end per XCD (us from the launch's first start)")
for l in range(0, 32, 3):
    dbg = torch.zeros(2 * NW * 8 + 8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    lib.deft_debug_set_buffer(dbg.data_ptr())
    # SUSTAINED=n: n launches back to back in front of the sampled one (every launch stamps the same buffer; the last one's stamps stay)
    for k in range(int(os.environ.get("SUSTAINED", "0")), 0, -1):
        j = (l - k) % 32
        b.attn[j](b.q[j], b.k_new[j], b.v_new[j], b.meta)
    b.attn[l](b.q[l], b.k_new[l], b.v_new[l], b.meta)
    torch.cuda.synchronize()
    lib.deft_debug_set_buffer(None)
    d = dbg.cpu().numpy()[: NW * 8].reshape(NW, 8)
    ok = d[:, 3] > 0
    n, t0, t1, we = d[ok, 4], d[ok, 1] / 100.0, d[ok, 2] / 100.0, d[ok, 3] / 100.0
    base = (d[ok, 0] / 100.0).min()
    xcc = (d[ok, 5] >> 32) & 0xff
    per = [np.median((t1[xcc == x] - t0[xcc == x]) / n[xcc == x]) for x in range(8)]
    end = [we[xcc == x].max() - base for x in range(8)]
    print(f"{l:2d}: " + " ".join(f"{v:4.2f}" for v in per) + "  |  " + " ".join(f"{v:4.1f}" for v in end) + f"   span {max(end):.1f}")
