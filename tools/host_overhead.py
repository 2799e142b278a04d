#!/usr/bin/env python3
"""Host cost of one attention call through the Python surface (launch-only, no sync in the loop)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import Bench
from deft_amd.utils.workloads import WORKLOADS
import deft_amd
for name in ("northstar_4kx32", "medusa64_node", "northstar_4kx32_seq"):
    b = Bench(WORKLOADS[name], 32, torch.device("cuda", 0))
    deft_amd.register_tree_metadata(b.md)
    b.step_eager(); torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        b.step_eager()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{name}: host {t_host / n / 32 * 1e6:.1f} us per layer call, wall {t_all / n / 32 * 1e6:.1f} us per layer")
    del b; torch.cuda.empty_cache()
