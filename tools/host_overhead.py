#!/usr/bin/env python3
"""Host cost of one attention call (launch-only, no sync in the loop): the Python surface end to end, and the bare
C-ABI call with prepared arguments (what remains is the HIP runtime's two kernel launches)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import Bench
from deft_amd._lib import lib
from deft_amd.tree_attention import _flatten_plan
from deft_amd.utils.workloads import WORKLOADS
import deft_amd
for name in ("northstar_4kx32", "medusa64_node"):
    b = Bench(WORKLOADS[name], 32, torch.device("cuda", 0))
    deft_amd.register_tree_metadata(b.md)
    b.step_eager(); torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        b.step_eager()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    line = f"{name}: python surface {t_host / n / 32 * 1e6:.1f} us per layer call"
    if b.w.mode == "flatten":
        md, pool = b.md, b.pool
        NB, P = md.block_q_cnts.shape[0], md.block_q.shape[0]
        ws_bytes = lib.deft_flatten_workspace_bytes(NB, P, b.nq, b.Hq, b.Hkv, b.D)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        mdl = [md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv, md.block_lens]
        q0 = b.q[0].view(b.nq, b.Hq, b.D)
        plan = _flatten_plan(mdl, NB, P, b.Hq, b.Hkv, (q0.stride(0), q0.stride(1)), pool.get_key_buffer(0).stride(0), st)
        out = torch.empty_like(q0)
        args = []
        for l in range(32):
            q = b.q[l].view(b.nq, b.Hq, b.D); kb, vb = pool.get_key_buffer(l), pool.get_value_buffer(l)
            args.append((q.data_ptr(), q.stride(0), q.stride(1), kb.data_ptr(), vb.data_ptr(), kb.stride(0), kb.stride(1),
                         out.data_ptr(), out.stride(0), out.stride(1), *[t.data_ptr() for t in mdl], NB, P, b.nq, b.Hq, b.Hkv,
                         b.D, b.D ** -0.5, plan.data_ptr(), ws.data_ptr(), ws_bytes, st))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            for a in args:
                lib.deft_flatten_decode_f16(*a)
        t_c = time.perf_counter() - t0
        torch.cuda.synchronize()
        line += f"; bare C-ABI call {t_c / n / 32 * 1e6:.1f} us"
    print(line)
    del b; torch.cuda.empty_cache()
