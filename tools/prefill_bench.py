#!/usr/bin/env python3
"""Prefill (causal) attention throughput: one sequence of S tokens, Llama geometry; TFLOP/s counts the causal half
(2 * 2 * S^2/2 * D * Hq) against the dense fp16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md)."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import deft_amd
from deft_amd.utils.workloads import GEOMETRY
rows = []
for model in ("llama2-7b", "llama3-8b"):
    Hq, Hkv, D, _ = GEOMETRY[model]
    for S in (1024, 4096, 8192, 16384):
        qkv = torch.randn((S, (Hq + 2 * Hkv) * D), dtype=torch.float16, device="cuda")
        q, k, v = (t.view(S, -1, D) for t in qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1))
        o = torch.empty((S, Hq, D), dtype=torch.float16, device="cuda")
        start = torch.zeros(1, dtype=torch.int32, device="cuda"); lens = torch.tensor([S], dtype=torch.int32, device="cuda")
        for _ in range(3):
            deft_amd.context_attention_fwd(q, k, v, o, start, lens, S)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            deft_amd.context_attention_fwd(q, k, v, o, start, lens, S)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        flops = 2.0 * 2.0 * (S * (S + 1) / 2) * D * Hq
        r = {"model": model, "S": S, "us": round(us, 1), "TFLOPs": round(flops / us / 1e6, 1), "frac_of_2.5PF": round(flops / us / 1e6 / 2500, 4)}
        rows.append(r); print(json.dumps(r), flush=True)
