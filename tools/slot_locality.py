#!/usr/bin/env python3
"""Does the pool-slot pattern of the leaf tokens matter?  The reference's allocator hands each decode step's slots to
the leaves in turn, so a branch's tokens sit 32 slots (512 KB) apart; here the SAME metadata is timed with the slot ids
remapped so that every branch is contiguous (data content is irrelevant for timing)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import Bench
from deft_amd.utils.workloads import WORKLOADS, Workload, GEOMETRY
bl = int(sys.argv[1]) if len(sys.argv) > 1 else 200
w = Workload(**{**WORKLOADS["northstar_4kx32"].__dict__, "branch_len": bl})
b = Bench(w, GEOMETRY[w.model][3], torch.device("cuda", 0)); b.prepare(use_graph=False)
base = b.time_stage1(reps=4)
P, W = w.prefix, w.width
kv = b.md.block_kv.clone()
live = kv >= P
s = kv[live] - P                      # step * W + leaf
kv[live] = P + (s % W) * bl + (s // W)  # leaf * branch_len + step
orig = b.md.block_kv
b.md.block_kv = kv
con = b.time_stage1(reps=4)
b.md.block_kv = orig
again = b.time_stage1(reps=4)
print(f"branch_len {bl}: interleaved slots {base['mean_us']:.2f} us | branch-contiguous slots {con['mean_us']:.2f} us | interleaved again {again['mean_us']:.2f} us")
