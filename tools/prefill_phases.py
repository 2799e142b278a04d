#!/usr/bin/env python3
"""Where a prefill wave spends a tile (experiments build: DEFT_AMD_LIB=deft_amd/lib/libdeft_amd_exp.so): per-phase wall-clock
sums of every wave of the first 1024 workgroups -- wait + barrier | next tile's requests | QK^T | softmax | PV -- as us per tile
the wave actually folded (waves above the diagonal skip a tile's arithmetic but not its barrier).  tools/prefill_phases.py [S] [model]"""
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
a = dbg.cpu().numpy()
wg = a[: NW * 8].reshape(NW, 8)
ph = a[NW * 8 :].reshape(NP, 8, 8).astype(np.float64)
ok = ph[:, 0, 6] > 0
ph = ph[ok]
names = ["wait + barrier", "next tile's requests", "QK^T (32 MFMAs)", "softmax (64 scores/lane)", "PV (32 MFMAs)"]
print(f"S={S} {model}: {int(ok.sum())} workgroups (the longest ones: launched first)")
for wv in (0, 3, 4, 7):
    folded, total = ph[:, wv, 5], ph[:, wv, 6]
    per = ph[:, wv, :5] / 100.0 / np.maximum(folded[:, None], 1)
    med = np.median(per, axis=0)
    print(f"  wave {wv}: folded {np.median(folded):.0f} of {np.median(total):.0f} tiles; us per folded tile: " +
          "  ".join(f"{n} {x:.2f}" for n, x in zip(names, med)) + f"  | sum {med.sum():.2f}")
