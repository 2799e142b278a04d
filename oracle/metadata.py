"""Oracle: TreeMetadata.from_tree_cache restated (CPU, pure Python + numpy).

Test infrastructure only (see oracle/__init__.py).  Follows
DeFT/deft/tree_decoding/tree_cache.py:618-881 step by step:

  * leaf -> query row: leaves sorted by node id                    (:650-652)
  * DFS from the root, children in creation order                  (:725-791)
  * per node: sorted KV slots, sorted query rows of live leaves    (:736-743)
  * KV-guided grouping -> node entries, q chunks of `max_q_len`
    (outer) x KV chunks of `max_block_len` (inner)                 (:744-758)
  * flattened-tree split -> `block_len`-slot blocks, padded with
    -1, one int64 query bitmask per slot, blocks with more than
    `max_q_len` queries emitted once per query chunk               (:661-723, :763-799)
  * offsets are exclusive prefix sums                              (:821-843)

All arrays are int64, as the reference tensors are (:813-857).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

from .tree_model import OracleNode, OracleTree

BLOCK_LEN = 128  # BLOCK_CONFIG["BLOCK_LEN"], tree_cache.py:587
MAX_Q_LEN = 32  # from_tree_cache default, tree_cache.py:623


def _excl_cumsum(xs: List[int]) -> np.ndarray:
    out = np.zeros(len(xs), dtype=np.int64)
    if len(xs) > 1:
        out[1:] = np.cumsum(np.asarray(xs[:-1], dtype=np.int64))
    return out


def build_metadata(
    tree: OracleTree,
    max_q_len: int = MAX_Q_LEN,
    block_len: int = BLOCK_LEN,
    max_block_len: int = -1,
) -> Dict[str, object]:
    leaf_to_q = {leaf.id: i for i, leaf in enumerate(tree.leaf_order())}

    node_q: List[int] = []
    node_kv: List[int] = []
    node_q_len: List[int] = []
    node_kv_len: List[int] = []
    block_q: List[int] = []
    block_q_cnts: List[int] = []
    block_bitmasks: List[int] = []
    block_kv: List[int] = []
    block_lens: List[int] = []
    total_kv_len = 0

    # running (not yet emitted) block: slots, and per-segment (q-set, length)
    cur_kv: List[int] = []
    cur_sets: List[set] = []
    cur_seg_lens: List[int] = []
    cur_union: set = set()

    def pack_block() -> None:  # tree_cache.py:661-723
        cur_len = len(cur_kv)
        if cur_len < block_len:
            cur_kv.extend([-1] * (block_len - cur_len))
            cur_seg_lens.append(block_len - cur_len)
            cur_sets.append(set())
        qs = sorted(cur_union)
        for lo in range(0, len(qs), max_q_len):
            chunk = qs[lo : lo + max_q_len]
            row = {q: i for i, q in enumerate(chunk)}
            block_q.extend(chunk)
            block_q_cnts.append(len(chunk))
            block_kv.extend(cur_kv)
            block_lens.append(cur_len)
            for qset, seg_len in zip(cur_sets, cur_seg_lens):
                mask = sum(1 << row[q] for q in qset if q in row)
                block_bitmasks.extend([mask] * seg_len)
        cur_kv.clear()
        cur_sets.clear()
        cur_seg_lens.clear()
        cur_union.clear()

    def dfs(node: OracleNode) -> None:  # tree_cache.py:725-791
        nonlocal total_kv_len
        assert len(node.refs) > 0 and len(node.token_ids) > 0
        kv = sorted(node.kv_indices)
        total_kv_len += len(kv)
        q = sorted(leaf_to_q[r] for r in node.refs)

        step = len(kv) if max_block_len == -1 else max_block_len
        kv_chunks = [kv[i : i + step] for i in range(0, len(kv), step)]  # step==0 raises, as upstream
        for i in range(0, len(q), max_q_len):
            q_chunk = q[i : i + max_q_len]
            for kv_chunk in kv_chunks:
                node_q.extend(q_chunk)
                node_q_len.append(len(q_chunk))
                node_kv.extend(kv_chunk)
                node_kv_len.append(len(kv_chunk))

        room = block_len - len(cur_kv)
        done = 0
        while done < len(kv):
            if len(kv) - done < room:
                piece = kv[done:]
                cur_kv.extend(piece)
                cur_union.update(q)
                cur_sets.append(set(q))
                cur_seg_lens.append(len(piece))
                break
            piece = kv[done : done + room]
            cur_kv.extend(piece)
            cur_union.update(q)
            cur_sets.append(set(q))
            cur_seg_lens.append(len(piece))
            pack_block()
            done += room
            room = block_len

        for child in node.children.values():
            dfs(child)

    assert tree.root is not None
    dfs(tree.root)
    if cur_seg_lens:  # tree_cache.py:797-798
        pack_block()

    i64 = lambda xs: np.asarray(xs, dtype=np.int64)  # noqa: E731
    return {
        "query_num": len(leaf_to_q),
        "node_num": len(node_q_len),
        "total_kv_len": total_kv_len,
        "leaf_to_q": leaf_to_q,
        "node_q": i64(node_q),
        "node_kv": i64(node_kv),
        "node_q_len": i64(node_q_len),
        "node_kv_len": i64(node_kv_len),
        "node_q_offset": _excl_cumsum(node_q_len),
        "node_kv_offset": _excl_cumsum(node_kv_len),
        "block_len": block_len,
        "block_q": i64(block_q),
        "block_q_cnts": i64(block_q_cnts),
        "block_q_offset": _excl_cumsum(block_q_cnts),
        "block_bitmasks": i64(block_bitmasks),
        "block_kv": i64(block_kv),
        "block_lens": i64(block_lens),
    }


ARRAY_FIELDS = (
    "node_q",
    "node_kv",
    "node_q_len",
    "node_kv_len",
    "node_q_offset",
    "node_kv_offset",
    "block_q",
    "block_q_cnts",
    "block_q_offset",
    "block_bitmasks",
    "block_kv",
    "block_lens",
)
SCALAR_FIELDS = ("query_num", "node_num", "total_kv_len", "block_len")
