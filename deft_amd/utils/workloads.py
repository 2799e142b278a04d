"""Synthetic decoding trees of the shapes BASELINE.json names, built with the
product TreeCache in the reference's allocation order (prompt slots contiguous,
then one `alloc()` per decode step handing consecutive slots to different leaves).

Shapes follow the reference's branch functions
(DeFT/deft/tree_decoding/generation/branch_func_example.py):
  few_shot   SimpleTree :12-62 — root branches once into `width` leaves after
             prefill, then every leaf grows one token per step
  medusa     SpeculativeDecoding mock :374-442 — root + `tree_size` one-token leaves
  tot        FromTreeTemplate :293-371 shaped like SURVEY §8d cfg4(i): root ->
             7 x 128-token nodes -> 42 x 64-token leaves (50 live nodes)
  medusa_tree  BASELINE configs[2] read literally: the depth-4 width-10 Medusa token tree itself (`Tree_Structure` of
             dataset/generation/Speculative_Decoding/tree_size64.json: 63 one-token nodes, 42 of them leaves) below the prompt --
             the shape a speculative-decoding caller that keeps its token tree AS a tree hands the operator
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import torch

from ..memory_pool import ReqToTokenPool, TokenToKVPool
from ..tree_cache import TreeCache

# (Hq, Hkv, D, layers)
GEOMETRY: Dict[str, Tuple[int, int, int, int]] = {
    "llama2-7b": (32, 32, 128, 32),
    "llama3-8b": (32, 8, 128, 32),
    # a head_dim-64 geometry (GPT-2-XL / Pythia-class heads) for the tile-per-workgroup kernel that serves head dims < 128
    "mha-d64": (32, 32, 64, 32),
}


# The 63 node paths of the reference's tree_size64 token tree (Medusa's `mc_sim_7b_63` choices: a node is the path of top-k
# indices that leads to it).  tests/golden/templates.json holds the reference file's own copy; tests/test_workloads.py compares.
MEDUSA_TREE_SIZE64 = (
    (0,), (0, 0), (1,), (0, 1), (2,), (0, 0, 0), (1, 0), (0, 2), (3,), (0, 3), (4,), (0, 4), (2, 0), (0, 5), (0, 0, 1), (5,), (0, 6),
    (6,), (0, 7), (0, 1, 0), (1, 1), (7,), (0, 8), (0, 0, 2), (3, 0), (0, 9), (8,), (9,), (1, 0, 0), (0, 2, 0), (1, 2), (0, 0, 3),
    (4, 0), (2, 1), (0, 0, 4), (0, 0, 5), (0, 0, 0, 0), (0, 1, 1), (0, 0, 6), (0, 3, 0), (5, 0), (1, 3), (0, 0, 7), (0, 0, 8),
    (0, 0, 9), (6, 0), (0, 4, 0), (1, 4), (7, 0), (0, 1, 2), (2, 0, 0), (3, 1), (2, 2), (8, 0), (0, 5, 0), (1, 5), (1, 0, 1),
    (0, 2, 1), (9, 0), (0, 6, 0), (0, 0, 0, 1), (1, 6), (0, 7, 0),
)


def build_token_tree(tree, paths, tok: int = 7) -> dict:
    """Grow a tree of ONE-TOKEN nodes below `tree.root` (which holds the prompt), one node per path, level by level: the nodes of
    a level that have children branch (children in ascending order of their last index), every new node takes one token and
    one pool slot.  Written against the reference's own API (`branch`, `append_token`, `append_index`, the pools) so that the
    same function drives the reference's TreeCache (tools/gen_golden.py), the oracle's and deft_amd's.  The slot step is
    `tree.alloc()` (tree_cache.py:261-277) restricted to the level's NEW nodes: `alloc()` itself would hand a second slot to
    the childless nodes of the levels above, which are live leaves too.  Returns {path: node}."""
    paths = sorted(tuple(p) for p in paths)
    by_path = {(): tree.root}
    for depth in range(1, max(len(p) for p in paths) + 1):
        level = [p for p in paths if len(p) == depth]
        new = []
        for parent in sorted({p[:-1] for p in level}):
            kids = [p for p in level if p[:-1] == parent]
            for p, node in zip(kids, tree.branch(by_path[parent], len(kids))):
                by_path[p] = node
                new.append(node)
        for node in new:
            node.append_token(tok)
        loc = tree.token_to_kv_pool.alloc(len(new))
        assert loc is not None
        table = tree.req_to_token_pool.req_to_token
        for i, node in enumerate(sorted(new, key=lambda n: n.id)):
            slot = int(loc[i])
            node.append_index(slot)
            table[tree.leaf_to_req[node.id], node.positions[-1]] = slot
    return by_path


@dataclass
class Workload:
    name: str
    model: str
    mode: str  # "flatten" | "node" | "node_chunk" | "seq" (the sequential per-leaf comparator, `--mode seq`)
    kind: str  # few_shot | medusa | tot
    prefix: int
    width: int = 32
    branch_len: int = 1
    trees: int = 1  # independent trees per GPU, decoded as one batch (deft_amd.Forest)


WORKLOADS: Dict[str, Workload] = {
    # north-star: Llama-2-7B, 4k shared prefix x 32 branches, DeFT-Flatten
    "northstar_4kx32": Workload("northstar_4kx32", "llama2-7b", "flatten", "few_shot", 4096, 32, 200),
    # BASELINE configs[1]: 1k shared prefix x 32 branches
    "fewshot_1kx32": Workload("fewshot_1kx32", "llama2-7b", "flatten", "few_shot", 1024, 32, 200),
    # the same two trees through the sequential comparator (token_attention_fwd): what DeFT is measured against
    "northstar_4kx32_seq": Workload("northstar_4kx32_seq", "llama2-7b", "seq", "few_shot", 4096, 32, 200),
    "fewshot_1kx32_seq": Workload("fewshot_1kx32_seq", "llama2-7b", "seq", "few_shot", 1024, 32, 200),
    # the north-star tree through DeFT-Node (the reference's other tree mode: one entry per node, no bit masks)
    "northstar_4kx32_node": Workload("northstar_4kx32_node", "llama2-7b", "node", "few_shot", 4096, 32, 200),
    # ... and through --mode node_chunk (DeFT-Node with every node cut into 128-token entries, run_DeFT_llama_paged.py:145-150)
    "northstar_4kx32_node_chunk": Workload("northstar_4kx32_node_chunk", "llama2-7b", "node_chunk", "few_shot", 4096, 32, 200),
    # configs[2]: Medusa depth-4 width-10 template as the reference mocks it (tree_size64), DeFT-Node
    "medusa64_node": Workload("medusa64_node", "llama2-7b", "node", "medusa", 1016, 64, 1),
    # configs[2] read literally: the 63-node depth-4 width-10 token tree itself (42 leaves = 42 queries), both modes
    "medusa64_tree_node": Workload("medusa64_tree_node", "llama2-7b", "node", "medusa_tree", 1016, 42, 1),
    "medusa64_tree_flatten": Workload("medusa64_tree_flatten", "llama2-7b", "flatten", "medusa_tree", 1016, 42, 1),
    # the north-star tree on a GQA model (Llama-3-8B: 32 branches x 4 query heads per KV head = 128 rows per tile)
    "gqa_4kx32": Workload("gqa_4kx32", "llama3-8b", "flatten", "few_shot", 4096, 32, 200),
    # configs[3]: Llama-3-8B ToT tree, 4k prefix, 50 nodes, DeFT-Flatten
    "tot50_4k": Workload("tot50_4k", "llama3-8b", "flatten", "tot", 4096),
    # configs[4]: 64 independent 8k-prefix trees (8 branches x 64 tokens) over 8 GPUs, Llama-3-8B:
    # one GPU's share = 8 trees, decoded as ONE batch; and a single tree of the same shape for comparison
    "forest_8kx8": Workload("forest_8kx8", "llama3-8b", "flatten", "few_shot", 8192, 8, 64, 8),
    "forest_8kx8_single": Workload("forest_8kx8_single", "llama3-8b", "flatten", "few_shot", 8192, 8, 64),
    # a long shared prefix (not a BASELINE configuration: where the kernel goes when the launch is long)
    "fewshot_16kx32": Workload("fewshot_16kx32", "llama2-7b", "flatten", "few_shot", 16384, 32, 200),
    "fewshot_64kx8_gqa": Workload("fewshot_64kx8_gqa", "llama3-8b", "flatten", "few_shot", 65536, 8, 200),
    # head_dim 64 (not a Llama geometry; measured once so the number exists): the north-star tree shape
    "northstar_4kx32_d64": Workload("northstar_4kx32_d64", "mha-d64", "flatten", "few_shot", 4096, 32, 200),
}


def tree_tokens(w: Workload) -> int:
    if w.kind == "few_shot":
        return w.prefix + w.width * w.branch_len
    if w.kind == "medusa":
        return w.prefix + w.width
    if w.kind == "tot":
        return w.prefix + 7 * 128 + 42 * 64
    if w.kind == "medusa_tree":
        return w.prefix + len(MEDUSA_TREE_SIZE64)
    raise ValueError(w.kind)


def build_tree(w: Workload, layers: int, device: str, extra_slots: int = 256, pools=None):
    """Returns (tree, kv_pool).  The pool holds `layers` distinct per-layer KV arrays.
    `pools` = (req_pool, kv_pool) grows the tree inside existing pools (forests)."""
    Hq, Hkv, D, _ = GEOMETRY[w.model]
    if pools is None:
        size = tree_tokens(w) + extra_slots
        req = ReqToTokenPool(max(w.width, 64) + 8, size + 8, device=device)
        pool = TokenToKVPool(size, torch.float16, Hkv, D, layers, device=device)
    else:
        req, pool = pools
    tree = TreeCache(torch.float16, Hkv, D, layers, req, pool, None, True, False)
    tree.init_prompt(torch.arange(1, w.prefix + 1, dtype=torch.int32))

    def step(n):
        for _ in range(n):
            for leaf in list(tree.leaves.values()):
                leaf.append_token(7)
            tree.alloc()

    if w.kind == "few_shot":
        tree.branch(tree.root, w.width)
        step(w.branch_len)
    elif w.kind == "medusa":
        tree.branch(tree.root, w.width)
        step(1)
    elif w.kind == "medusa_tree":
        build_token_tree(tree, MEDUSA_TREE_SIZE64)
    elif w.kind == "tot":
        tree.branch(tree.root, 7)
        step(128)
        for leaf in sorted(tree.leaves.values(), key=lambda n: n.id):
            tree.branch(leaf, 6)
        step(64)
    else:
        raise ValueError(w.kind)
    return tree, pool


def algorithmic_bytes(n_kv_tokens: int, nq: int, Hq: int, Hkv: int, D: int) -> int:
    """unique KV read once (K and V, fp16) + Q read + O written."""
    return 4 * n_kv_tokens * Hkv * D + 4 * nq * Hq * D


def build_forest(w: Workload, n_trees: int, layers: int, device: str, extra_slots: int = 256):
    """`n_trees` independent trees of shape `w` in ONE pool (BASELINE configs[4]: a GPU's share of the 64-tree
    batch).  Trees are built one after the other, so each tree's prompt is contiguous and its decode-step slots
    interleave across its own leaves only.  Returns (Forest, kv_pool)."""
    from ..forest import Forest

    Hq, Hkv, D, _ = GEOMETRY[w.model]
    size = n_trees * tree_tokens(w) + extra_slots
    req = ReqToTokenPool(n_trees * (max(w.width, 64) + 8), tree_tokens(w) + 16, device=device)
    pool = TokenToKVPool(size, torch.float16, Hkv, D, layers, device=device)
    trees = [build_tree(w, layers, device, pools=(req, pool))[0] for _ in range(n_trees)]
    return Forest(trees), pool


def build_forest_tree(w: Workload, n_trees: int, layers: int, device: str, extra_slots: int = 256):
    """The same batch as ONE tree object (`TreeCache.init_forest`: a root without tokens, the trees below it), which is what
    the device-side metadata kernels and `DecodeSession` take: per decode step only the new slot numbers cross PCIe, where
    `Forest.metadata()` rebuilds and concatenates every tree's metadata on the host.  few_shot shapes.  Returns (tree, pool)."""
    assert w.kind == "few_shot"
    Hq, Hkv, D, _ = GEOMETRY[w.model]
    size = n_trees * tree_tokens(w) + extra_slots
    req = ReqToTokenPool(n_trees * (w.width + 8) + 8, tree_tokens(w) + 512, device=device)
    pool = TokenToKVPool(size, torch.float16, Hkv, D, layers, device=device)
    tree = TreeCache(torch.float16, Hkv, D, layers, req, pool, None, True, False)
    tree.init_forest([torch.arange(1, w.prefix + 1, dtype=torch.int32) for _ in range(n_trees)])
    for root in sorted(tree.leaves.values(), key=lambda n: n.id):
        tree.branch(root, w.width)
    for _ in range(w.branch_len):
        for leaf in tree.leaves.values():
            leaf.append_token(7)
        tree.alloc()
    return tree, pool
