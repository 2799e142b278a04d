"""A batch of independent decoding trees in ONE paged pool, attended by ONE operator call.

The reference decodes a single tree per process (`TreeCache` has one root,
DeFT/deft/tree_decoding/tree_cache.py:94-130; `run_DeFT_llama_paged.py` loops over trees).  BASELINE.json's
batched configuration ("64 independent 8k-prefix trees") needs several trees per GPU: trees share nothing,
so their operator metadata simply concatenates — block lists / node entries one after the other, query rows
offset per tree — and `tree_attention_subtree_fwd` / `tree_attention_fwd` run unchanged on the
result.  One launch over 8 trees streams 8x the KV of one tree, which is what lets a small-tree batch reach the
same HBM efficiency as one large tree.

Query-row order of the batch: tree 0's leaves (sorted by node id, as `TreeMetadata.from_tree_cache` orders
them, tree_cache.py:640-651), then tree 1's, ...  `Forest.alloc()` returns the new slots in the same order.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .tree_cache import BLOCK_CONFIG, KVCacheUpdater, TreeCache, TreeMetadata, _FIELDS, build_metadata_host

# which arrays are shifted by what when tree t is appended to the batch
_SHIFT_BY_QUERIES = ("node_q", "block_q")          # values are query rows
_SHIFT_BY_NODE_Q = ("node_q_offset",)              # offsets into node_q
_SHIFT_BY_NODE_KV = ("node_kv_offset",)            # offsets into node_kv
_SHIFT_BY_BLOCK_Q = ("block_q_offset",)            # offsets into block_q


def concat_metadata_host(hosts: Sequence[Dict[str, object]]) -> Dict[str, object]:
    """Concatenate per-tree host metadata (dicts from `build_metadata_host`) into one batch; numpy only."""
    lens = {k: sum(int(h["_lens"][k]) for h in hosts) for k in _FIELDS}
    packed = np.empty(sum(lens.values()), dtype=np.int64)
    views, off = {}, 0
    for k in _FIELDS:
        views[k] = packed[off : off + lens[k]]
        off += lens[k]
    pos = {k: 0 for k in _FIELDS}
    q_base = node_q_base = node_kv_base = block_q_base = 0
    leaf_to_q: Dict[Tuple[int, int], int] = {}
    q_bases: List[int] = []
    for t, h in enumerate(hosts):
        for k in _FIELDS:
            n = int(h["_lens"][k])
            dst = views[k][pos[k] : pos[k] + n]
            dst[:] = h[k]
            if k in _SHIFT_BY_QUERIES:
                dst += q_base
            elif k in _SHIFT_BY_NODE_Q:
                dst += node_q_base
            elif k in _SHIFT_BY_NODE_KV:
                dst += node_kv_base
            elif k in _SHIFT_BY_BLOCK_Q:
                dst += block_q_base
            pos[k] += n
        for leaf, qi in h["leaf_to_q"].items():
            leaf_to_q[(t, int(leaf))] = q_base + int(qi)
        q_bases.append(q_base)
        q_base += int(h["query_num"])
        node_q_base += int(h["_lens"]["node_q"])
        node_kv_base += int(h["_lens"]["node_kv"])
        block_q_base += int(h["_lens"]["block_q"])
    out: Dict[str, object] = dict(views)
    out.update(query_num=q_base, node_num=sum(int(h["node_num"]) for h in hosts),
               total_kv_len=sum(int(h["total_kv_len"]) for h in hosts), block_len=hosts[0]["block_len"],
               leaf_to_q=leaf_to_q, q_bases=q_bases, _packed=packed, _lens=lens)
    return out


class Forest:
    """Independent `TreeCache`s over one `TokenToKVPool` / `ReqToTokenPool`."""

    def __init__(self, trees: Sequence[TreeCache]) -> None:
        assert len(trees) > 0
        pool = trees[0].token_to_kv_pool
        for t in trees:
            assert t.token_to_kv_pool is pool, "all trees of a forest live in one KV pool"
            assert t.root is not None
        self.trees: List[TreeCache] = list(trees)
        self.token_to_kv_pool = pool

    @property
    def query_num(self) -> int:
        return sum(len(t.leaves) for t in self.trees)

    def alloc(self) -> KVCacheUpdater:
        """One decode step for every tree: a new slot per live leaf (tree_cache.py:261-283 per tree); the
        returned updater's `cache_loc` is in batch query-row order."""
        locs = [t.alloc().cache_loc for t in self.trees]
        return KVCacheUpdater(True, self.token_to_kv_pool, torch.cat(locs), None, False)

    def metadata(self, max_q_len: int = 32, max_block_len: int = -1, device: Optional[str] = None) -> TreeMetadata:
        block_len = BLOCK_CONFIG["BLOCK_LEN"]
        if max_block_len == -1:
            max_block_len = BLOCK_CONFIG["MAX_BLOCK_LEN"]
        host = concat_metadata_host([build_metadata_host(t, max_q_len, block_len, max_block_len) for t in self.trees])
        dev = torch.device(device) if device is not None else self.token_to_kv_pool.device
        packed = torch.from_numpy(host["_packed"])
        if dev.type != "cpu":
            packed = packed.pin_memory().to(dev, non_blocking=True)  # one H2D copy for the whole batch
        views, off = {}, 0
        for k in _FIELDS:
            n = host["_lens"][k]
            views[k] = packed[off : off + n]
            off += n
        md = TreeMetadata(query_num=host["query_num"], node_num=host["node_num"], total_kv_len=host["total_kv_len"],
                          leaf_to_q=host["leaf_to_q"], block_len=block_len, **views)
        md.q_bases = host["q_bases"]  # first query row of every tree
        return md

    def leaf_paths(self) -> List[List[int]]:
        """Root->leaf pool slots of every query row, in batch order (for sequential-attention comparators)."""
        out: List[List[int]] = []
        for t in self.trees:
            for leaf in sorted(t.leaves.values(), key=lambda n: n.id):
                out.append(t.leaf_path_slots(leaf))
        return out
