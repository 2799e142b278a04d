#!/usr/bin/env python3
"""Does re-launching ONE hipGraphExec back to back cost more than alternating between TWO captures of the same step?
(The advancing loop shows ~20 us of idle queue in front of every step's first kernel, a frozen step ~8.6: tools/step_boundary.py.)
   python tools/experiments/graph_alternation.py [workload] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench as B

wl = sys.argv[1] if len(sys.argv) > 1 else "northstar_4kx32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
w = B.WORKLOADS[wl]
b = B.Bench(w, B.GEOMETRY[w.model][3], dev, seed=0)
b.prepare(use_graph=True)


def capture():
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            b.step_eager()
    torch.cuda.current_stream(dev).wait_stream(side)
    g.replay()
    torch.cuda.synchronize(dev)
    return g


graphs = [b.graph] + [capture() for _ in range(3)]


def timed(n_graphs):
    for i in range(20):
        graphs[i % n_graphs].replay()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        graphs[i % n_graphs].replay()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e6


for rep in range(3):
    print(wl, "us per step:", {n: round(timed(n), 1) for n in (1, 2, 4)}, flush=True)
