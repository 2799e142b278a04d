#!/usr/bin/env python3
"""Shader cycles of the one-wave-per-SIMD prefill kernel (experiments build, DEFT_PREFILL_64=1): per wave of the first 1024
workgroups, cycles per woven step (64 MFMAs) and per step in wait + barrier.  tools/prefill64_cycles.py [S] [model]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import deft_amd
from deft_amd._lib import lib
from deft_amd.utils.workloads import GEOMETRY
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
model = sys.argv[2] if len(sys.argv) > 2 else "llama2-7b"
Hq, Hkv, D, _ = GEOMETRY[model]
qkv = torch.randn((S, (Hq + 2 * Hkv) * D), dtype=torch.float16, device="cuda")
q, k, v = (t.view(S, -1, D) for t in qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1))
o = torch.empty((S, Hq, D), dtype=torch.float16, device="cuda")
start = torch.zeros(1, dtype=torch.int32, device="cuda"); lens = torch.tensor([S], dtype=torch.int32, device="cuda")
lib.deft_debug_set_buffer.argtypes = [ctypes.c_void_p]
for _ in range(3): deft_amd.context_attention_fwd(q, k, v, o, start, lens, S)
NW, NP = 8192, 1024
dbg = torch.zeros(NW * 8 + NP * 8 * 8, dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
lib.deft_debug_set_buffer(dbg.data_ptr())
deft_amd.context_attention_fwd(q, k, v, o, start, lens, S); torch.cuda.synchronize()
lib.deft_debug_set_buffer(None)
ph = dbg.cpu().numpy()[NW * 8:].reshape(NP, 8, 8).astype(np.float64)
ok = ph[:, 3, 2] > 0
ph = ph[ok]
print(f"S={S} {model}: {int(ok.sum())} workgroups")
for wv in range(4):
    sync, woven, n, loop, ntw, nta = (ph[:, wv, i] for i in range(6))
    print(f"  wave {wv}: sub-tiles {np.median(ntw):.0f} of {np.median(nta):.0f}; cycles per woven step {np.median(woven / np.maximum(n, 1)):.0f} "
          f"(= {np.median(woven / np.maximum(n, 1)) / 64:.1f} per MFMA, wait + barrier included: {np.median(sync / np.maximum(nta, 1)):.0f} per step); whole loop {np.median(loop):.0f} "
          f"= {np.median(loop / np.maximum(ntw, 1)) / 64:.1f} cycles per MFMA of the wave's sub-tiles")
