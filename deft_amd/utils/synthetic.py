"""Seeded synthetic fp16 inputs that are bit-identical on every machine.

Values are sums of three uniform integers scaled by 1/512: bell-shaped with
standard deviation ~1 (like the `torch.randn` inputs of the reference's kernel
script, DeFT/tests/model/test_DeFT_kernel.py:52-54, :80-85), all multiples of
2^-9 with |x| <= 3, hence exact in fp16.  Only integer PRNG output is used, so
the build container (where golden vectors are generated) and the GPU box
(where they are checked) regenerate the same bytes and only outputs need to be
stored as fixtures.
"""
from __future__ import annotations

import numpy as np


def dyadic_normal(shape, seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(int(seed)))
    n = int(np.prod(shape))
    acc = np.zeros(n, dtype=np.int32)
    for _ in range(3):
        acc += rng.integers(-512, 513, size=n, dtype=np.int32)
    return (acc.astype(np.float32) / np.float32(512.0)).astype(np.float16).reshape(shape)


def fill_kv_rows(kv_data: np.ndarray, slots, seed: int) -> None:
    """Fill `kv_data[slots]` ([n, 2, Hkv, D]) with seeded values; other slots untouched."""
    slots = np.asarray(slots, dtype=np.int64)
    kv_data[slots] = dyadic_normal((len(slots),) + kv_data.shape[1:], seed)


def permutation_scores(it: int, rows: int, vocab: int = 4096) -> np.ndarray:
    """Next-token "probabilities" for a replayed decode step: row r of iteration `it` is a PERMUTATION of (1 .. vocab) / sum --
    positive, no two equal (so top-k and argmax have no ties to break, in float32 and after a log), integer arithmetic only.
    The golden replays (tools/gen_golden_replay.py) feed the reference's branch functions with these; the tests feed
    deft_amd.replay with the same."""
    assert vocab & (vocab - 1) == 0, "vocab must be a power of two (odd multipliers are bijections mod 2^k)"
    j = np.arange(vocab, dtype=np.int64)[None, :]
    r = np.arange(rows, dtype=np.int64)[:, None]
    a = 2 * ((int(it) * 7919 + r * 104729 + 12345) % (vocab // 2)) + 1
    b = (int(it) * 31 + r * 17 + 5) % vocab
    perm = (a * j + b) % vocab
    return ((perm + 1).astype(np.float64) / (vocab * (vocab + 1) / 2)).astype(np.float32)
