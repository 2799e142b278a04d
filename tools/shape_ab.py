"""Whole-layer latency of arbitrary tree shapes under the library named by DEFT_AMD_LIB -- the rules of the plan are tuned on
the BASELINE shapes; this is how they are checked elsewhere (tools/shape_ab.sh runs it under several builds on one box).

    python tools/shape_ab.py <model> <mode> <kind>:<prefix>:<width>:<branch_len>[:<trees>] ...
    model: llama2-7b | llama3-8b | mha-d64;  mode: flatten | node | seq;  kind: few_shot | medusa | tot
Prints one line per shape: us per layer of the captured 32-layer step (100 timed steps).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from deft_amd.utils.workloads import GEOMETRY, Workload  # noqa: E402


def main():
    model, mode = sys.argv[1], sys.argv[2]
    for spec in sys.argv[3:]:
        f = spec.split(":")
        kind, prefix, width, blen = f[0], int(f[1]), int(f[2]), int(f[3])
        trees = int(f[4]) if len(f) > 4 else 1
        w = Workload(spec, model, mode, kind, prefix, width, blen, trees)
        torch.cuda.empty_cache()
        b = bench.Bench(w, GEOMETRY[model][3], "cuda:0", seed=1)
        b.prepare(use_graph=True)
        dt = bench.run_timed(b, 100, 10, False)
        print("%-28s %-10s %-8s %7.2f us per layer   (nq %d, %d KV tokens)" % (spec, model, mode, dt / 100 * 1e6 / b.layers, b.nq, b.n_kv), flush=True)
        del b


if __name__ == "__main__":
    main()
