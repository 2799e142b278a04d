"""Oracle: token-paged KV pool, page table and decoding tree (CPU, numpy).

Test infrastructure only (see oracle/__init__.py).  Restates, in plain
Python/numpy, the state machine of the reference:

  * TokenToKVPool   DeFT/deft/memory_pool.py:48-108   (int16 refcounts,
                    first-free allocation `nonzero(mem_state == 0)[:n]`)
  * ReqToTokenPool  DeFT/deft/memory_pool.py:11-45    (leaf -> slot page table)
  * TreeNode        DeFT/deft/tree_decoding/tree_cache.py:94-130
  * TreeCache       DeFT/deft/tree_decoding/tree_cache.py:147-403, :504-516
                    (paged branch only; unpaged / tree_index modes are out of scope)

KV payload is not stored here: the pool only hands out slot numbers.  The KV
bytes live in a caller-owned array shaped like the reference's per-layer
`kv_data[layer]` = [size, 2, Hkv, D] (memory_pool.py:61-66).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np


class OracleTokenPool:
    """memory_pool.py:48-108 — refcounted slots, lowest free index first."""

    def __init__(self, size: int) -> None:
        self.mem_state = np.zeros(size, dtype=np.int16)
        self.alloc_ct = 0

    def alloc(self, need: int) -> Optional[np.ndarray]:
        free = np.flatnonzero(self.mem_state == 0)[:need]  # :74-77
        if free.shape[0] < need:
            return None
        self.add_refs(free)
        return free.astype(np.int32)

    def add_refs(self, idx) -> None:  # :91-93
        idx = np.asarray(idx, dtype=np.int64)
        self.alloc_ct += len(idx)
        np.add.at(self.mem_state, idx, 1)

    def free(self, idx) -> int:  # :82-83, :95-104
        idx = np.asarray(idx, dtype=np.int64)
        self.alloc_ct -= len(idx)
        np.subtract.at(self.mem_state, idx, 1)
        return int(np.sum(self.mem_state[idx] == 0))

    def used_size(self) -> int:
        return int(np.count_nonzero(self.mem_state))


class OracleReqTable:
    """memory_pool.py:11-45 — one page-table row per live leaf."""

    def __init__(self, size: int, max_context_len: int) -> None:
        self.mem_state = np.ones(size, dtype=bool)
        self.req_to_token = np.zeros((size, max_context_len), dtype=np.int32)

    def alloc(self) -> int:
        free = np.flatnonzero(self.mem_state)
        assert free.shape[0] >= 1, "request table exhausted"
        self.mem_state[free[0]] = False
        return int(free[0])

    def free(self, req: int) -> None:
        self.mem_state[req] = True

    def copy(self, src: int, dst: int, n: int) -> None:  # :38-41
        self.req_to_token[dst, :n] = self.req_to_token[src, :n]


class OracleNode:
    """tree_cache.py:94-130."""

    def __init__(self, nid: int) -> None:
        self.id = nid
        self.children: Dict[int, "OracleNode"] = {}  # insertion order = creation order
        self.token_ids: List[int] = []
        self.positions: List[int] = []
        self.position_offset = 0
        self.kv_indices: List[int] = []
        self.parent: Optional["OracleNode"] = None
        self.refs: set = set()  # ids of live leaves at/under this node

    def append_token(self, token: int) -> None:  # :119-123
        self.positions.append(self.position_offset + len(self.token_ids))
        self.token_ids.append(token)

    def append_index(self, index: int) -> None:  # :125-129
        self.kv_indices.append(int(index))


class OracleTree:
    """tree_cache.py:147-403 (paged memory only)."""

    def __init__(self, pool: OracleTokenPool, reqs: OracleReqTable) -> None:
        self.pool = pool
        self.reqs = reqs
        self.token_to_kv_pool, self.req_to_token_pool = pool, reqs  # (the reference's names, tree_cache.py:168-169)
        self.node_cnt = 1
        self.root: Optional[OracleNode] = None
        self.nodes: Dict[int, OracleNode] = {}
        self.leaves: Dict[int, OracleNode] = {}
        self.leaf_to_req: Dict[int, int] = {}

    # -- refs (:504-516) -------------------------------------------------
    def _add_ref(self, node: OracleNode) -> None:
        ref = node.id
        cur: Optional[OracleNode] = node
        while cur is not None:
            cur.refs.add(ref)
            cur = cur.parent

    def _remove_ref(self, node: OracleNode) -> None:
        ref = node.id
        cur: Optional[OracleNode] = node
        while cur is not None:
            cur.refs.remove(ref)
            cur = cur.parent

    # -- :192-230 --------------------------------------------------------
    def init_prompt(self, prompt_ids) -> np.ndarray:
        prompt_ids = [int(t) for t in np.asarray(prompt_ids).reshape(-1)]
        root = OracleNode(0)
        self.root = root
        self.nodes[0] = root
        root.token_ids = prompt_ids
        root.positions = list(range(len(prompt_ids)))
        self.leaves[0] = root
        self._add_ref(root)
        req = self.reqs.alloc()
        self.leaf_to_req[0] = req
        loc = self.pool.alloc(len(prompt_ids))
        assert loc is not None, "token pool exhausted"
        root.kv_indices = [int(x) for x in loc]
        self.reqs.req_to_token[req, : len(prompt_ids)] = loc
        return loc

    # -- :242-259 --------------------------------------------------------
    def _new_node(self, parent: OracleNode) -> OracleNode:
        node = OracleNode(self.node_cnt)
        self.node_cnt += 1
        node.parent = parent
        node.position_offset = parent.position_offset + len(parent.positions)
        parent.children[node.id] = node
        self.nodes[node.id] = node
        return node

    # -- :261-283 --------------------------------------------------------
    def alloc(self) -> np.ndarray:
        """One new slot per live leaf, handed out in leaf-id order."""
        loc = self.pool.alloc(len(self.leaves))
        assert loc is not None, "token pool exhausted"
        for i, leaf in enumerate(sorted(self.leaves.values(), key=lambda n: n.id)):
            slot = int(loc[i])
            leaf.kv_indices.append(slot)
            self.reqs.req_to_token[self.leaf_to_req[leaf.id], leaf.positions[-1]] = slot
        return loc

    # -- :338-370 --------------------------------------------------------
    def branch(self, node: OracleNode, branch_cnt: int) -> List[OracleNode]:
        assert node.id in self.leaves
        self.leaves.pop(node.id)
        req = self.leaf_to_req.pop(node.id)
        path_len = node.positions[-1] + 1
        out: List[OracleNode] = []
        for i in range(branch_cnt):
            child = self._new_node(node)
            out.append(child)
            self.leaves[child.id] = child
            if i == 0:
                self.leaf_to_req[child.id] = req
            else:
                new_req = self.reqs.alloc()
                self.reqs.copy(req, new_req, path_len)
                self.leaf_to_req[child.id] = new_req
        self._remove_ref(node)
        for child in out:
            self._add_ref(child)
        return out

    # -- :373-403 --------------------------------------------------------
    def cut(self, node: OracleNode) -> List[OracleNode]:
        assert len(node.children) == 0 and node.id in self.leaves
        self.leaves.pop(node.id)
        self._remove_ref(node)
        self.reqs.free(self.leaf_to_req.pop(node.id))
        deleted = []
        cur: Optional[OracleNode] = node
        while cur is not None and len(cur.refs) == 0:
            deleted.append(self.nodes.pop(cur.id))
            self.pool.free(cur.kv_indices)
            parent = cur.parent
            if parent is not None:
                parent.children.pop(cur.id)
            cur = parent
        return deleted

    # -- :300-336 (speculative-decoding mock) ----------------------------
    def merge_nodes(self, node_A: OracleNode, node_B: OracleNode, pruneB_flag: bool = True) -> None:
        a, b, prune_b = node_A, node_B, pruneB_flag
        for tok in b.token_ids:
            # the reference appends the position here AND inside append_token
            a.positions.append(a.position_offset + len(a.token_ids))
            a.append_token(tok)
        for slot in b.kv_indices:
            a.kv_indices.append(slot)
        self.pool.add_refs(b.kv_indices)
        if prune_b:
            self.cut(b)

    def reset_node_KV(self, node: OracleNode, diff: int) -> None:
        self.pool.free(node.kv_indices)
        node.kv_indices = []
        node.position_offset += diff
        node.positions = [p + diff for p in node.positions]

    # -- helpers used by tests / ground truth ----------------------------
    def leaf_order(self) -> List[OracleNode]:
        return sorted(self.leaves.values(), key=lambda n: n.id)

    def path_slots(self, leaf: OracleNode) -> List[int]:
        """Root->leaf KV slots of one leaf (what sequential attention reads)."""
        chain = []
        cur: Optional[OracleNode] = leaf
        while cur is not None:
            chain.append(cur)
            cur = cur.parent
        out: List[int] = []
        for node in reversed(chain):
            out.extend(node.kv_indices)
        return out
