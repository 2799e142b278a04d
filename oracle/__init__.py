"""CPU oracle for the DeFT paged tree-attention decode path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a plain numpy / pure-Python restatement of the reference algorithm
(LINs-lab/DeFT @ 2025-07-04) for the one path this repository accelerates:
paged DeFT-Flatten / DeFT-Node tree attention at decode time.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it, and only as the checker.  Nothing under `deft_amd/` imports it; the product
path fails loudly when the HIP library is missing instead of falling back here.

Pinning: the reference's own tests hold no golden vectors for this path
(DeFT/tests/model/test_DeFT_kernel.py exits at :208 before its parity section),
so the oracle is pinned against outputs of the reference itself, run in the
build container under Triton's CPU interpreter by `tools/gen_golden.py`, and
committed as fixtures under `tests/golden/` (see tests/test_oracle_golden.py).

Modules
  tree_model.py   token pool, page table, tree (init_prompt/branch/alloc/cut/...)
  metadata.py     TreeMetadata.from_tree_cache restated (KV-guided grouping +
                  flattened-tree split)
  attention.py    Flatten / Node stage-1, LSE merge, per-leaf sequential truth
  cpu_baseline.py the PyTorch-CPU sequential-attention baseline timed by bench.py
"""
