#!/usr/bin/env python3
"""Host profile (cProfile) of a template replay: where the per-step host time of alloc / metadata / branch / launches goes.
   tools/replay_profile.py [task] [mode]"""
import cProfile, io, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deft_amd import replay as rp
from deft_amd.utils.workloads import GEOMETRY

task = sys.argv[1] if len(sys.argv) > 1 else "speculative_decoding"
mode = sys.argv[2] if len(sys.argv) > 2 else "flatten"
Hq, Hkv, D, L = GEOMETRY["llama2-7b"]
if task == "speculative_decoding":
    tpl, plen, gen = rp.synthetic_speculative_template(64, 100), 1016, 400
elif task == "reasoning":
    tpl, plen, gen = rp.synthetic_reasoning_template(), 4096, 400
else:
    tpl, plen, gen = rp.synthetic_few_shot_template(32), 4096, 200
r = rp.TemplateReplay(Hq, Hkv, D, L, mode=mode)
r.run(tpl, task, plen, gen)  # warm
pr = cProfile.Profile(); pr.enable()
rep = r.run(tpl, task, plen, gen)
pr.disable()
print({k: v for k, v in rep.summary().items() if not isinstance(v, (list, dict))} if hasattr(rep, "summary") else rep)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
