#!/usr/bin/env python3
"""Per-workgroup timeline of the prefill kernel (experiments build: DEFT_AMD_LIB=deft_amd/lib/libdeft_amd_exp.so):
   start, tile loop entered, epilogue, end (100 MHz clock), per block length."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import deft_amd
from deft_amd._lib import lib
from deft_amd.utils.workloads import GEOMETRY
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
model = sys.argv[2] if len(sys.argv) > 2 else "llama2-7b"
Hq, Hkv, D, _ = GEOMETRY[model]
qkv = torch.randn((S, (Hq + 2 * Hkv) * D), dtype=torch.float16, device="cuda")
q, k, v = (t.view(S, -1, D) for t in qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1))
o = torch.empty((S, Hq, D), dtype=torch.float16, device="cuda")
start = torch.zeros(1, dtype=torch.int32, device="cuda"); lens = torch.tensor([S], dtype=torch.int32, device="cuda")
lib.deft_debug_set_buffer.argtypes = [ctypes.c_void_p]
for _ in range(3): deft_amd.context_attention_fwd(q, k, v, o, start, lens, S)
NW = 8192
for rep in range(2):
    dbg = torch.zeros(NW * 8, dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
    lib.deft_debug_set_buffer(dbg.data_ptr())
    deft_amd.context_attention_fwd(q, k, v, o, start, lens, S); torch.cuda.synchronize()
    lib.deft_debug_set_buffer(None)
    d = dbg.cpu().numpy().reshape(NW, 8); d = d[d[:, 3] > 0]
    t0 = d[:, 0].min()
    st, lp, ep, en = [(d[:, i] - t0) / 100.0 for i in range(4)]; n = d[:, 4]
    pct = lambda x: [round(float(np.percentile(x, p)), 1) for p in (0, 10, 50, 90, 100)]
    print(f"rep {rep}: S={S} {model}: {len(d)} workgroups, span {en.max():.1f} us; sum of tiles / 256 CUs = {n.sum() / 256:.1f}")
    print("  prologue (start -> loop)", pct(lp - st), " epilogue", pct(en - ep), " per tile (loop / tiles)", pct((ep - lp) / n))
    for nn in sorted(set(n.tolist()))[:: max(1, len(set(n.tolist())) // 8)]:
        m = n == nn
        print(f"  tiles={int(nn):3d}: {int(m.sum()):3d} WGs start {pct(st[m])} per-tile {pct(((ep - lp) / n)[m])} end {pct(en[m])}")
    late = en > np.percentile(en, 95)
    print("  last 5% to end: tiles", sorted(set(n[late].astype(int).tolist())), "start", pct(st[late]))
