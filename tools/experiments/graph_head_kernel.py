#!/usr/bin/env python3
"""What does ONE small kernel in front of (or behind) the 32 layers of a captured step cost?  (tools/step_boundary.py: ~20 us of idle
queue in front of an advancing step's first kernel -- a lone workgroup -- against ~8.6 in front of a frozen step's first stage 1.)
   python tools/experiments/graph_head_kernel.py [workload] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench as B
from deft_amd._lib import lib, check

wl = sys.argv[1] if len(sys.argv) > 1 else "northstar_4kx32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
w = B.WORKLOADS[wl]
b = B.Bench(w, B.GEOMETRY[w.model][3], dev, seed=0)
b.prepare(use_graph=True)
x = torch.zeros(64, device=dev)
ring = torch.zeros(4 * 4096, dtype=torch.uint8).pin_memory()
dst = torch.zeros(4096, dtype=torch.uint8, device=dev)
ctr = torch.zeros(1, dtype=torch.int32, device=dev)


def tiny():
    x.add_(1.0)


def fetch():
    check(lib.deft_stage_fetch(ring.data_ptr(), 4096, 4, dst.data_ptr(), ctr.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "fetch")


def capture(head=None, tail=None):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            if head:
                head()
            b.step_eager()
            if tail:
                tail()
    torch.cuda.current_stream(dev).wait_stream(side)
    g.replay()
    torch.cuda.synchronize(dev)
    return g


variants = {"layers only": capture(), "tiny kernel at the head": capture(head=tiny), "tiny kernel at the tail": capture(tail=tiny),
            "stage_fetch at the head": capture(head=fetch), "two tiny kernels at the head": capture(head=lambda: (tiny(), tiny()))}


def timed(g):
    for i in range(20):
        g.replay()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        g.replay()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e6


for rep in range(3):
    print(wl, {k: round(timed(g), 1) for k, g in variants.items()}, flush=True)
