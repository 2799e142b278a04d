#!/usr/bin/env python3
"""Shape of the Flatten plan of the north-star tree by branch length and chunk length (experiments build: DEFT_NP_CHUNK read per
call): units R, chunk leaders NL (work items per KV head), tiles per leader -- to read profiles/r6_chunk_sweep_short.txt against.
   DEFT_AMD_LIB=deft_amd/lib/libdeft_amd_exp.so python tools/experiments/plan_shape.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
os.environ.setdefault("DEFT_AMD_LIB", os.path.join(ROOT, "deft_amd", "lib", "libdeft_amd_exp.so"))
import numpy as np, torch
import deft_amd
from deft_amd._lib import lib, check
from deft_amd.utils.workloads import WORKLOADS, Workload, build_tree

w0 = WORKLOADS["northstar_4kx32"]
Hq = Hkv = 32
D = 128
LENS = [int(x) for x in sys.argv[1:]] or [1, 10, 25, 40, 50, 75, 100, 125, 150, 175, 200, 300, 400]
for L in LENS:
    w = Workload(**{**w0.__dict__, "branch_len": L})
    tree, pool = build_tree(w, 1, "cuda:0")
    md = deft_amd.TreeMetadata.from_tree_cache(tree)
    mdl = [md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv, md.block_lens]
    NB, P = md.block_q_cnts.shape[0], md.block_q.shape[0]
    nbytes = lib.deft_flatten_plan_bytes(NB, P, Hq, Hkv)
    out = []
    for C in (0, 3, 4, 5, 6, 7, 8):
        os.environ["DEFT_NP_CHUNK"] = str(C)
        plan = torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0")
        check(lib.deft_flatten_build_plan(*[t.data_ptr() for t in mdl], NB, P, Hq, Hkv, Hq * D, D, pool.kv_data[0].stride(0), None, 0, 0,
                                          plan.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream), "plan")
        torch.cuda.synchronize()
        hdr = plan[:16].view(torch.int32).cpu().numpy()
        R, NL = int(hdr[0]), int(hdr[1])
        rec = plan[4096 : 4096 + 2048 * NL].view(-1, 2048)
        desc = rec[:, 1536:1568].contiguous().view(torch.int32).view(-1, 8).cpu().numpy()
        tiles = desc[:, 4]
        hist = {int(k): int(v) for k, v in zip(*np.unique(tiles, return_counts=True))}
        out.append(f"C={C or 'rule'}: NL {NL:3d} (x32 = {NL * 32:4d}) tiles/leader {hist}")
    print(f"L={L:3d} NB={NB:3d}  " + " | ".join(out), flush=True)
    del tree, pool
