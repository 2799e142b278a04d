#!/usr/bin/env python3
"""Golden vectors for the prefill operator: tests/golden/prefill.npz, made by running the REFERENCE's
`context_attention_fwd` (DeFT/deft/layers/attention/context_flashattention_nopad.py:130-195) on CPU under
TRITON_INTERPRET=1.  Build-container only.  Shims (test side, like tools/gen_golden.py): the module asks
`torch.cuda.get_device_capability()` at import (:10) -- answered with (8, 0), which selects the reference's BLOCK = 128
path, the one it runs on A100/H100-class parts.  Inputs are regenerated from seeds by deft_amd.utils.synthetic.

Usage:  PYTHONDONTWRITEBYTECODE=1 TRITON_INTERPRET=1 python tools/gen_golden_prefill.py
"""
import os
import sys
import time

os.environ.setdefault("TRITON_INTERPRET", "1")
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/DeFT")
torch.cuda.get_device_capability = lambda *a, **k: (8, 0)  # type: ignore[assignment]
import deft.layers.attention.context_flashattention_nopad as ref  # noqa: E402
from deft_amd.utils.synthetic import dyadic_normal  # noqa: E402

CASES = {"single_300": ([300], (4, 4, 128)), "single_513_gqa": ([513], (8, 2, 128)), "batch_ragged": ([130, 77, 256, 1], (4, 4, 128)),
         # the other head dimensions the reference takes (:134): 64 on the MFMA kernel, 32 / 16 on the small-head-dim kernel
         "d64_gqa_385": ([385], (4, 2, 64)), "d64_batch_ragged": ([130, 77, 256, 1], (4, 4, 64)), "d32_gqa_200": ([200, 3], (4, 2, 32)),
         "d16_150": ([150], (2, 2, 16))}
out = {}
for name, (lens, (Hq, Hkv, D)) in CASES.items():
    t0 = time.time()
    T = sum(lens)
    q = torch.from_numpy(dyadic_normal((T, Hq, D), 101))
    k = torch.from_numpy(dyadic_normal((T, Hkv, D), 102))
    v = torch.from_numpy(dyadic_normal((T, Hkv, D), 103))
    o = torch.zeros((T, Hq, D), dtype=torch.float16)
    b_seq_len = torch.tensor(lens, dtype=torch.int32)
    b_start_loc = torch.zeros(len(lens), dtype=torch.int32)
    b_start_loc[1:] = torch.cumsum(b_seq_len[:-1], dim=0)
    ref.context_attention_fwd(q, k, v, o, b_start_loc, b_seq_len, max(lens))
    # a sample of rows keeps the fixture small: every 5th token plus everything around the 128-token block edges
    rows = sorted(set(range(0, T, 5)) | {r for e in range(0, T + 129, 128) for r in range(e - 3, e + 3) if 0 <= r < T} | {T - 1})
    out[name + "_rows"] = np.asarray(rows, dtype=np.int32)
    out[name + "_o"] = o.numpy()[rows].copy()
    out[name + "_lens"] = np.asarray(lens, dtype=np.int32)
    out[name + "_geom"] = np.asarray([Hq, Hkv, D], dtype=np.int32)
    print(name, lens, (Hq, Hkv, D), f"{time.time() - t0:.1f}s", flush=True)
path = os.path.join(ROOT, "tests", "golden", "prefill.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes")
