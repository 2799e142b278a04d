#!/usr/bin/env python3
"""Per-phase cycle breakdown of the streaming stage-1 kernel (s_memtime stamps)."""
import ctypes, os, sys, json
os.environ.setdefault("DEFT_STAGE1_KERNEL", "stream")  # this tool reads the streaming form's stamps
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import Bench
from deft_amd._lib import lib
from deft_amd.utils.workloads import WORKLOADS, Workload, GEOMETRY
bl = int(sys.argv[1]) if len(sys.argv) > 1 else 200
w = Workload(**{**WORKLOADS["northstar_4kx32"].__dict__, "branch_len": bl})
b = Bench(w, 4, torch.device("cuda", 0))
workers = 256 if os.environ.get("DEFT_STREAM_DB") == "1" else 512
dbg = torch.zeros(workers * 16 * 8 + workers * 2, dtype=torch.int64, device="cuda")
lib.deft_debug_set_buffer.argtypes = [ctypes.c_void_p]
b.time_stage1(reps=1)
lib.deft_debug_set_buffer(dbg.data_ptr())
b.layers = 1
b.time_stage1(reps=1)
torch.cuda.synchronize()
lib.deft_debug_set_buffer(None)
raw = dbg.cpu().numpy()
d = raw[: workers * 128].reshape(workers, 16, 8).astype(np.float64)
print("dbg wg0 slot15", raw[15*8:16*8].tolist(), "slot0", raw[0:8].tolist()); rt = d[:, 15, 6:8].copy(); d[:, 15, :] = 0
names = ["A wait K", "B qk+max", "C barrier", "D+E issueK+softmax", "F wait V", "G pv", "H barrier", "I+J issueV+out -> next A"]
valid = d[:, :, 0] > 0
n_tiles = valid.sum(1)
out = {"tiles_per_wg": [int(n_tiles.min()), int(n_tiles.max())]}
seg = {}
for k in range(7):
    x = (d[:, :, k + 1] - d[:, :, k])[valid]
    seg[names[k]] = [round(float(np.mean(x))), round(float(np.median(x))), round(float(np.percentile(x, 90)))]
nxt = []
for wg in range(workers):
    for i in range(int(n_tiles[wg]) - 1):
        nxt.append(d[wg, i + 1, 0] - d[wg, i, 7])
seg[names[7]] = [round(float(np.mean(nxt))), round(float(np.median(nxt))), round(float(np.percentile(nxt, 90)))]
out["cycles mean/median/p90"] = seg
tot = [(d[wg, int(n_tiles[wg]) - 1, 7] - d[wg, 0, 0]) / max(1, n_tiles[wg]) for wg in range(workers) if n_tiles[wg] > 0]
out["cycles_per_tile_mean"] = round(float(np.mean(tot)))
span = d[:, :, 7][valid].max() - d[:, :, 0][valid].min()
out["kernel_span_cycles"] = float(span)
st = (rt[:, 0] - rt[:, 0].min()) / 100.0
en = (rt[:, 1] - rt[:, 0].min()) / 100.0
out["wg_start_us pct0/50/90/100"] = [round(float(np.percentile(st, q)), 2) for q in (0, 50, 90, 100)]
out["wg_end_us pct0/50/90/100"] = [round(float(np.percentile(en, q)), 2) for q in (0, 50, 90, 100)]
out["wg_dur_us pct0/50/90/100"] = [round(float(np.percentile(en - st, q)), 2) for q in (0, 50, 90, 100)]
print(json.dumps(out, indent=1))
