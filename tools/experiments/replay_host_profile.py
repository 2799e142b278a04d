#!/usr/bin/env python3
"""cProfile of the SYNCHRONISED speculative-decoding replay (where the host time between two steps goes).
   python tools/experiments/replay_host_profile.py [mode]"""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from deft_amd import replay as rp
mode = sys.argv[1] if len(sys.argv) > 1 else "flatten"
for rep in range(2):
    tpl = rp.synthetic_speculative_template(64, 100)
    r = rp.TemplateReplay(32, 32, 128, 32, mode=mode, device="cuda", attention=True)
    pr = cProfile.Profile()
    if rep:
        pr.enable()
    out = r.run(tpl, "speculative_decoding", rp.default_prompt_len(tpl, "speculative_decoding", from_file=False), 400, max_rows=512, pipelined=False)
    if rep:
        pr.disable()
        s = out.summary()
        print({k: s[k] for k in ("steps", "attention_latency_ms", "metadata_ms", "branch_ms", "wall_ms")})
        pstats.Stats(pr).sort_stats("tottime").print_stats(28)
    del r
