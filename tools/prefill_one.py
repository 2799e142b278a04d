#!/usr/bin/env python3
"""One prefill shape a few times (for rocprofv3 --pmc passes): tools/prefill_one.py [S] [model] [reps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import deft_amd
from deft_amd.utils.workloads import GEOMETRY
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
Hq, Hkv, D, _ = GEOMETRY[sys.argv[2] if len(sys.argv) > 2 else "llama2-7b"]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
qkv = torch.randn((S, (Hq + 2 * Hkv) * D), dtype=torch.float16, device="cuda")
q, k, v = (t.view(S, -1, D) for t in qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1))
o = torch.empty((S, Hq, D), dtype=torch.float16, device="cuda")
start = torch.zeros(1, dtype=torch.int32, device="cuda"); lens = torch.tensor([S], dtype=torch.int32, device="cuda")
for _ in range(reps):
    deft_amd.context_attention_fwd(q, k, v, o, start, lens, S)
torch.cuda.synchronize()
