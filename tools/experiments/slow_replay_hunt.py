#!/usr/bin/env python3
"""The pipelined speculative-decoding replay (tools/replay.py --task speculative_decoding --tree-size 64 --pipelined) repeated, with
perf_counter around the calls that can stall (as tools/experiments/slow_call_hunt.py): which call, when a replay runs 2-3 x slower.
   python tools/experiments/slow_replay_hunt.py MODE REPS"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import deft_amd
from deft_amd import replay as rp, session as S, tree_cache as TC
THRESH = float(os.environ.get("THRESH_MS", "3")) * 1e-3
slow = []


def timed(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **kw):
        t = time.perf_counter()
        try:
            return f(*a, **kw)
        finally:
            dt = time.perf_counter() - t
            if dt > THRESH:
                slow.append((label, round(dt * 1e3, 2)))
    setattr(obj, name, g)


for n in ("_epoch_setup", "_capture", "_stage", "_launch_step", "_launch_window_step", "_staged", "step"):
    timed(S.DecodeSession, n, "DecodeSession." + n)
timed(TC._DeviceTree, "sync", "_DeviceTree.sync")
timed(torch, "empty", "torch.empty")
timed(torch, "zeros", "torch.zeros")
timed(torch.cuda.CUDAGraph, "replay", "CUDAGraph.replay")
timed(torch.Tensor, "pin_memory", "Tensor.pin_memory")
timed(torch.cuda.Event, "synchronize", "Event.synchronize")
for n in ("deft_window_create", "deft_window_step", "deft_stage_copy", "deft_tree_alloc_step", "deft_tree_journal_take"):
    timed(deft_amd.lib, n, n)
mode, reps = sys.argv[1], int(sys.argv[2])
tot = []
for rep in range(reps):
    tpl = rp.synthetic_speculative_template(64, 100)
    r = rp.TemplateReplay(32, 32, 128, 32, mode=mode, device="cuda", attention=True)
    slow.clear()
    out = r.run(tpl, "speculative_decoding", rp.default_prompt_len(tpl, "speculative_decoding", from_file=False), 400, max_rows=512, pipelined=True)
    s = out.summary()
    tot.append(s["attention_us_per_step"])
    flag = "SLOW " if s["attention_us_per_step"] > 1.3 * sorted(tot)[len(tot) // 2] else ""
    print(f"{flag}rep {rep}: {s['attention_us_per_step']:.1f} us per step, wall {s['wall_ms']:.1f} ms; calls over {THRESH * 1e3:.0f} ms: {slow}"[:1500], flush=True)
    del r
    torch.cuda.empty_cache()
print("median us per step", sorted(tot)[len(tot) // 2], "max", max(tot))
