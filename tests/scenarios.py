"""Scripted decoding-tree scenarios shared by the golden generator and the tests.

Each script drives a tree object through the reference's own mutation API
(`init_prompt / branch / alloc / cut / merge_nodes / reset_node_KV`,
DeFT/deft/tree_decoding/tree_cache.py:192-403) in the order the reference's
decode loop does (DeFT/deft/tree_decoding/generation/tree_generate.py:92-169:
the branch function appends a token to every leaf, the next step calls
`tree.alloc()` and then builds the metadata).  The same script is replayed on

  * the reference `TreeCache`             (tools/gen_golden.py, build container only)
  * the oracle `OracleTree`               (tests, CPU)
  * the product `deft_amd.TreeCache`      (tests, CPU and GPU)

so slot numbers, page tables and metadata must agree bit for bit.
"""
from __future__ import annotations

from typing import Callable, Dict, NamedTuple


def _step(tree, n: int = 1, tok: int = 7) -> None:
    """n decode steps: every live leaf gets a token, then one pool slot."""
    for _ in range(n):
        for leaf in list(tree.leaves.values()):
            leaf.append_token(tok)
        tree.alloc()


def _leaves(tree):
    return sorted(tree.leaves.values(), key=lambda n: n.id)


def s_cfgA(tree, ids):  # BASELINE config 1: 256-token prefix, 2 branches
    tree.init_prompt(ids(256))
    tree.branch(tree.root, 2)
    _step(tree, 1)


def s_chain(tree, ids):  # no branching: the root is the only leaf
    tree.init_prompt(ids(300))
    _step(tree, 5)


def s_wide40(tree, ids):  # 40 leaves > 32 -> query chunking / block duplication
    tree.init_prompt(ids(200))
    tree.branch(tree.root, 40)
    _step(tree, 3)


def s_edge128(tree, ids):  # root ends exactly on a 128-slot boundary
    tree.init_prompt(ids(128))
    tree.branch(tree.root, 4)
    _step(tree, 2)


def s_edge_fill(tree, ids):  # leaf pieces straddle / exactly fill blocks
    tree.init_prompt(ids(250))
    tree.branch(tree.root, 3)
    _step(tree, 2)  # 250 + 3*2 = 256 slots: the last block is exactly full


def s_multilevel(tree, ids):  # ToT-like: root -> 3 -> 6 leaves
    tree.init_prompt(ids(300))
    tree.branch(tree.root, 3)
    _step(tree, 40)
    for leaf in _leaves(tree):
        tree.branch(leaf, 2)
    _step(tree, 10)


def s_after_cut(tree, ids):  # pruning frees ancestors whose leaf set empties
    s_multilevel(tree, ids)
    lv = _leaves(tree)
    tree.cut(lv[0])
    tree.cut(lv[1])  # both children of the first mid node -> mid node freed too
    tree.cut(lv[4])
    _step(tree, 2)  # freed slots are reused lowest-first


def s_node_chunk(tree, ids):  # --mode node_chunk: MAX_BLOCK_LEN = 128
    tree.init_prompt(ids(300))
    tree.branch(tree.root, 5)
    _step(tree, 4)


def s_spec_mock(tree, ids):
    """Speculative-decoding mock (branch_func_example.py:374-442): leaves keep one
    token; accepted leaves' KV is merged into the root, then every leaf is reset."""
    tree.init_prompt(ids(100))
    tree.branch(tree.root, 8)
    for leaf in list(tree.leaves.values()):
        leaf.append_token(7)
    tree.alloc()
    for accepted in (2, 1):
        leaves = list(tree.leaves.values())
        before = len(tree.root.kv_indices)
        for i in range(accepted):
            tree.merge_nodes(tree.root, leaves[i], pruneB_flag=False)
        diff = len(tree.root.kv_indices) - before
        for leaf in leaves:
            tree.reset_node_KV(leaf, diff)
        tree.alloc()


def s_appendix_d(tree, ids):  # SURVEY.md Appendix D, BLOCK_LEN=4, max_q_len=2
    tree.init_prompt(ids(5))
    kids = tree.branch(tree.root, 3)
    _step(tree, 2)
    tree.branch(kids[1], 2)
    _step(tree, 1)


def s_medusa64(tree, ids):  # BASELINE config 3 as the reference mocks it
    tree.init_prompt(ids(1016))
    tree.branch(tree.root, 64)
    _step(tree, 1)


def s_medusa64_tree(tree, ids):
    """BASELINE config 3 read literally (VERDICT r4 missing #3): the depth-4 width-10 Medusa token tree of
    dataset/generation/Speculative_Decoding/tree_size64.json (`Tree_Structure`: 63 one-token nodes, 42 leaves) below a
    1016-token prompt.  The topology comes from tests/golden/templates.json (the reference file's own copy)."""
    import json
    import os

    from deft_amd.utils.workloads import build_token_tree

    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "templates.json")))
    tree.init_prompt(ids(1016))
    build_token_tree(tree, gold["speculative"]["tree_size64"]["Tree_Structure"])


def s_tot50(tree, ids):  # BASELINE config 4: 4096 root -> 7 x 128 -> 42 x 64
    tree.init_prompt(ids(4096))
    tree.branch(tree.root, 7)
    _step(tree, 128)
    for leaf in _leaves(tree):
        tree.branch(leaf, 6)
    _step(tree, 64)


def s_fewshot_1k(tree, ids):  # BASELINE config 2 at branch length 1
    tree.init_prompt(ids(1024))
    tree.branch(tree.root, 32)
    _step(tree, 1)


def s_fewshot_1k_len200(tree, ids):  # BASELINE config 2 at the branch length bench.py measures it at
    tree.init_prompt(ids(1024))
    tree.branch(tree.root, 32)
    _step(tree, 200)


def s_fewshot_4k(tree, ids):  # north-star tree at branch length 1
    tree.init_prompt(ids(4096))
    tree.branch(tree.root, 32)
    _step(tree, 1)


def s_fewshot_4k_len200(tree, ids):  # north-star tree at the benchmarked branch length (BASELINE north_star)
    tree.init_prompt(ids(4096))
    tree.branch(tree.root, 32)
    _step(tree, 200)


def s_forest_tree_8kx8(tree, ids):  # BASELINE config 5: ONE of the 64 trees (8192-token prefix, 8 branches x 64 tokens)
    tree.init_prompt(ids(8192))
    tree.branch(tree.root, 8)
    _step(tree, 64)


class Scenario(NamedTuple):
    script: Callable
    max_q_len: int = 32
    block_len: int = 128
    max_block_len: int = -1
    pool_size: int = 8192
    kernels: bool = True  # also emit kernel goldens for this tree


SCENARIOS: Dict[str, Scenario] = {
    "cfgA_256x2": Scenario(s_cfgA, pool_size=512),
    "chain_300": Scenario(s_chain, pool_size=512),
    "wide40": Scenario(s_wide40, pool_size=512),
    "edge128": Scenario(s_edge128, pool_size=256),
    "edge_fill": Scenario(s_edge_fill, pool_size=512),
    "multilevel": Scenario(s_multilevel, pool_size=1024),
    "after_cut": Scenario(s_after_cut, pool_size=1024),
    "node_chunk": Scenario(s_node_chunk, max_block_len=128, pool_size=512),
    "spec_mock": Scenario(s_spec_mock, pool_size=256),
    "appendix_d": Scenario(s_appendix_d, max_q_len=2, block_len=4, pool_size=32, kernels=False),
    "medusa64": Scenario(s_medusa64, pool_size=2048, kernels=False),
    "medusa64_tree": Scenario(s_medusa64_tree, pool_size=2048, kernels=False),
    "tot50": Scenario(s_tot50, pool_size=8192, kernels=False),
    "fewshot_1k": Scenario(s_fewshot_1k, pool_size=2048, kernels=False),
    "fewshot_1k_len200": Scenario(s_fewshot_1k_len200, pool_size=7680, kernels=False),
    "fewshot_4k": Scenario(s_fewshot_4k, pool_size=8192, kernels=False),
    "fewshot_4k_len200": Scenario(s_fewshot_4k_len200, pool_size=10752, kernels=False),
    "forest_tree_8kx8": Scenario(s_forest_tree_8kx8, pool_size=8832, kernels=False),
}

# (Hq, Hkv, D) used for the small kernel goldens
SMALL_GEOMETRIES = ((4, 4, 128), (8, 2, 128), (4, 4, 64))
# full Llama-2-7B geometry goldens (F3): scenario -> (Hq, Hkv, D).  BASELINE configs[1] (1k x 32), the north-star tree
# at branch length 1 and at the benchmarked length 200, configs[2] (Medusa-64) at the model BASELINE names for it
# (and configs[0], the 256-prefix x 2-branch plumbing case, at its own model's geometry too)
FULL_GEOMETRY = {"cfgA_256x2": (32, 32, 128), "fewshot_1k": (32, 32, 128), "fewshot_1k_len200": (32, 32, 128), "fewshot_4k": (32, 32, 128),
                 "fewshot_4k_len200": (32, 32, 128), "medusa64": (32, 32, 128), "medusa64_tree": (32, 32, 128)}
# Llama-3-8B GQA geometry: the Medusa tree, configs[3] (ToT-50) and one tree of configs[4] (8k x 8 x 64)
GQA_GEOMETRY = {"medusa64": (32, 8, 128), "medusa64_tree": (32, 8, 128), "tot50": (32, 8, 128), "forest_tree_8kx8": (32, 8, 128)}


# the reference's other head dims (`assert Lk in {16, 32, 64, 128}`, tree_attention.py:100, :582): two trees each
SMALL_D_GEOMETRY = {"multilevel": [(4, 4, 32), (4, 2, 16)], "wide40": [(4, 4, 32), (4, 2, 16)]}


def small_d_cases():
    return [(name, geom) for name, geoms in SMALL_D_GEOMETRY.items() for geom in geoms]


def big_cases():
    """(scenario, geometry) pairs of the full-size goldens, each once."""
    seen = []
    for d in (FULL_GEOMETRY, GQA_GEOMETRY):
        for name, geom in d.items():
            if (name, geom) not in seen:
                seen.append((name, geom))
    return seen


def input_seeds(name: str, geom) -> Dict[str, int]:
    """Deterministic seeds for q / kv of one (scenario, geometry) pair."""
    base = sum(ord(c) for c in name) * 131 + geom[0] * 17 + geom[1] * 5 + geom[2]
    return {"q": base + 1, "kv": base + 2}
