#!/usr/bin/env python3
"""Phases of the fused step-head kernel (experiments build: DEFT_AMD_LIB=deft_amd/lib/libdeft_amd_exp.so), us from the kernel's
start: scan + query lists | units + record order | row_q | row lists.   tools/step_head_phases.py [few_shot|speculative|tot]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import deft_amd
from deft_amd import replay as rp
which = sys.argv[1] if len(sys.argv) > 1 else "few_shot"
Hq, Hkv, D, L = (32, 32, 128, 4)
if which == "tot":
    Hq, Hkv = 32, 8
tpl, task, plen, gen = {"few_shot": (rp.synthetic_few_shot_template(32), "few_shot", 4096, 60),
                        "speculative": (rp.synthetic_speculative_template(64, 40), "speculative_decoding", 1016, 400),
                        "tot": (rp.synthetic_reasoning_template(), "reasoning", 4096, 400)}[which]
r = rp.TemplateReplay(Hq, Hkv, D, L, mode="flatten", device="cuda", attention=True)
stamps = []
def hook(tree, q, o):
    torch.cuda.synchronize()
    d = tree._device_tree.dims()
    stamps.append([x / 100.0 for x in d[10:14]])
r.step_hook = hook
r.run(tpl, task, plen, gen)
a = np.asarray(stamps[5:])
names = ["scan + query lists", "units + record order", "row_q", "row lists"]
prev = np.zeros(len(a))
for i, nm in enumerate(names):
    print(f"{nm:22s} ends at {np.median(a[:, i]):6.2f} us   (phase {np.median(a[:, i] - prev):6.2f} us)")
    prev = a[:, i]
