#!/usr/bin/env python3
"""Host-side profile of the advancing-tree decode loop (bench.py end_to_end): where the host time per step goes."""
import cProfile, os, pstats, sys, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import Bench
from deft_amd.utils.workloads import WORKLOADS, GEOMETRY
w = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "northstar_4kx32"]
b = Bench(w, GEOMETRY[w.model][3], torch.device("cuda", 0)); b.prepare(use_graph=False)
b.end_to_end(10)
pr = cProfile.Profile(); pr.enable()
r = b.end_to_end(40)
pr.disable()
print(r)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
