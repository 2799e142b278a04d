#!/usr/bin/env python3
"""Rotary embedding as its own launch in front of the decode step vs fused into stage 1, whole
32-layer step from one hipGraph each, same pools:   tools/rope_fused_ab.py [workload ...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import deft_amd
from bench import Bench
from deft_amd.utils.workloads import WORKLOADS, GEOMETRY

def timed(fn, steps=60, rounds=3):
    g = torch.cuda.CUDAGraph(); side = torch.cuda.Stream()
    fn(); fn(); torch.cuda.synchronize()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side): fn()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(10): g.replay()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps): g.replay()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / steps)
    return out

for name in (sys.argv[1:] or ["northstar_4kx32", "medusa64_node", "tot50_4k", "gqa_4kx32"]):
    w = WORKLOADS[name]
    b = Bench(w, GEOMETRY[w.model][3], torch.device("cuda", 0)); b.prepare(use_graph=False)
    rope = deft_amd.get_rope(b.D, b.D, 16384, 10000.0, True).cuda()
    leaves = sorted(b.forest.trees[0].leaves.values(), key=lambda n: n.id)
    pos = torch.tensor([len(b.forest.trees[0].leaf_path_slots(lf)) - 1 for lf in leaves], dtype=torch.int64, device="cuda")
    def plain():
        for l in range(b.layers): b.attn[l](b.q[l], b.k_new[l], b.v_new[l], b.meta)
    def separate():
        for l in range(b.layers):
            rope(pos, b.q[l], b.k_new[l]); b.attn[l](b.q[l], b.k_new[l], b.v_new[l], b.meta)
    def fused():
        for l in range(b.layers): b.attn[l](b.q[l], b.k_new[l], b.v_new[l], b.meta, rotary_emb=rope, positions=pos, fuse_rope=True)
    r = {k: timed(f) for k, f in (("attention only", plain), ("rope launch + attention", separate), ("fused", fused))}
    print(name, " | ".join(f"{k}: {np.min(v) / b.layers:.2f} us/layer" for k, v in r.items()), flush=True)
    del b; torch.cuda.empty_cache()
