#!/usr/bin/env python3
"""Generate tests/golden/rope.npz by running the REFERENCE's own rotary embedding on CPU.

Build-container only.  DeFT/deft/layers/rotary_embedding.py imports `flashinfer.rope` at module level (:31) for its
CUDA path; flashinfer is a third-party package that is not part of the reference tree and not installed here, so an
empty stand-in MODULE OBJECT is put into sys.modules for the import only -- no flashinfer code is written or run.
What runs is the reference's in-tree torch path: `get_rope(...)` (:647-690) builds the fp32 cos|sin cache
(`_compute_cos_sin_cache`, :119-127) and `forward_native` (:129-155, `_apply_rotary_emb` :47-72) rotates q and k.

Stored per case (data only; inputs come from the seeded integer PRNG of deft_amd.utils.synthetic):
  cache        the reference's cos_sin_cache rows at the sampled positions (fp32)
  q_f32/k_f32  forward_native on the fp32 UPCAST of the fp16 inputs, rounded to fp16: fp32 arithmetic, one rounding --
               the arithmetic of the decode path's kernel (flashinfer computes in fp32) and of deft_rope_qk_f16
  q_f16/k_f16  forward_native on the fp16 tensors as they are (cos / sin and every product rounded to fp16, :62-69):
               what the reference returns for fp16 CPU tensors

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_rope.py
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/DeFT")

stub = types.ModuleType("flashinfer.rope")
stub.apply_rope_with_cos_sin_cache_inplace = None  # never called: CPU tensors take forward_native (:179-190)
sys.modules.setdefault("flashinfer", types.ModuleType("flashinfer"))
sys.modules["flashinfer.rope"] = stub

import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_rotary_embedding", "/root/reference/DeFT/deft/layers/rotary_embedding.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

from deft_amd.utils.synthetic import dyadic_normal  # noqa: E402

# (Hq, Hkv, D, rotary_dim, neox, max_position, base)
CASES = [(32, 32, 128, 128, True, 4400, 10000.0), (32, 8, 128, 128, True, 8300, 500000.0), (8, 2, 64, 64, True, 2048, 10000.0),
         (4, 4, 128, 64, True, 1024, 10000.0), (4, 4, 128, 128, False, 1024, 10000.0)]
N = 16


def main():
    out = {"cases": np.asarray([[a, b, c, d, int(e), f, g] for a, b, c, d, e, f, g in CASES], dtype=np.float64)}
    for ci, (Hq, Hkv, D, rot, neox, maxpos, base) in enumerate(CASES):
        rope = ref.get_rope(D, rot, maxpos, base, neox, None, torch.float32)  # llama2.py:86-93 passes dtype=float32
        pos = np.random.default_rng(ci).integers(0, maxpos, size=N)
        q = dyadic_normal((N, Hq * D), 100 + ci)
        k = dyadic_normal((N, Hkv * D), 200 + ci)
        p = torch.from_numpy(pos)
        q32, k32 = rope.forward_native(p, torch.from_numpy(q).float(), torch.from_numpy(k).float())
        q16, k16 = rope.forward_native(p, torch.from_numpy(q), torch.from_numpy(k))
        out[f"pos_{ci}"] = pos.astype(np.int64)
        out[f"cache_{ci}"] = rope.cos_sin_cache.index_select(0, p).numpy().astype(np.float32)
        out[f"q_f32_{ci}"] = q32.half().numpy()
        out[f"k_f32_{ci}"] = k32.half().numpy()
        out[f"q_f16_{ci}"] = q16.numpy()
        out[f"k_f16_{ci}"] = k16.numpy()
    path = os.path.join(ROOT, "tests", "golden", "rope.npz")
    np.savez_compressed(path, **out)
    print(f"rope.npz: {len(CASES)} cases -> {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
