"""Oracle: rotary position embedding of q / k (numpy).  Test infrastructure only (see oracle/__init__.py).

Restates RotaryEmbedding as the reference's decode path runs it:
  cache      DeFT/deft/layers/rotary_embedding.py:103-127  inv_freq = base^(-2i/rot), cache[p] = cos(p f) | sin(p f),
             fp32 (get_rope(..., dtype=torch.float32), DeFT/deft/models/llama2.py:86-93)
  rotation   rotary_embedding.py:157-177 forward_cuda -> flashinfer.rope.apply_rope_with_cos_sin_cache_inplace.
             flashinfer is a third-party dependency that is NOT part of /root/reference (imported at
             rotary_embedding.py:31, no version pinned anywhere in the reference tree), so its published algorithm
             is restated: NeoX pairing o1 = x1 cos - x2 sin, o2 = x2 cos + x1 sin over (d, d + rot/2) [GPT-J:
             (2i, 2i+1)], fp32 arithmetic on fp16 inputs, one rounding back to fp16, dims >= rot untouched.
             `forward_native` (:129-155) is the same formula with cos / sin first rounded to the activation dtype; it
             is the module's CPU path, not the decode path.
Parity: PINNED on outputs of the reference itself -- tools/gen_golden_rope.py runs the reference's `get_rope` +
`forward_native` in the build container (an empty stand-in module object satisfies the flashinfer import; none of its
code exists or runs) and tests/golden/rope.npz holds the cache rows and both results: this function is bit-exact
against forward_native on fp32-upcast inputs (fp32 arithmetic, one rounding: the decode kernel's arithmetic) and
within a few fp16 ulps of forward_native on the fp16 tensors (tests/test_rope.py).  flashinfer's own binary remains
unavailable; what it has in common with forward_native -- cache, pairing, signs, fp32 products -- is what is pinned.
"""
from __future__ import annotations

import numpy as np


def cos_sin_cache(rotary_dim: int, max_position: int, base: float = 10000.0) -> np.ndarray:
    inv_freq = (1.0 / (np.float32(base) ** (np.arange(0, rotary_dim, 2, dtype=np.float32) / np.float32(rotary_dim)))).astype(np.float32)
    t = np.arange(max_position, dtype=np.float32)
    freqs = np.einsum("i,j->ij", t, inv_freq).astype(np.float32)
    return np.concatenate([np.cos(freqs), np.sin(freqs)], axis=-1).astype(np.float32)


def apply_rope(x: np.ndarray, positions: np.ndarray, cache: np.ndarray, rotary_dim: int, neox: bool = True) -> np.ndarray:
    """x [n, H, D] fp16 -> rotated copy, fp16."""
    half = rotary_dim // 2
    cs = cache[positions]  # [n, rot]
    cos, sin = cs[:, None, :half], cs[:, None, half:]
    xf = x.astype(np.float32)
    out = x.copy()
    if neox:
        x1, x2 = xf[..., :half], xf[..., half:rotary_dim]
        out[..., :half] = (x1 * cos - x2 * sin).astype(np.float16)
        out[..., half:rotary_dim] = (x2 * cos + x1 * sin).astype(np.float16)
    else:
        x1, x2 = xf[..., 0:rotary_dim:2], xf[..., 1:rotary_dim:2]
        out[..., 0:rotary_dim:2] = (x1 * cos - x2 * sin).astype(np.float16)
        out[..., 1:rotary_dim:2] = (x2 * cos + x1 * sin).astype(np.float16)
    return out
