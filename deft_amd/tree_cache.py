"""Decoding tree over the paged KV pool + the attention metadata contract.

Mirrors the paged half of DeFT/deft/tree_decoding/tree_cache.py with the same
names, argument meaning and error behaviour:

  KVCacheUpdater (:52-91)    new K/V rows -> pool slots (here: one fused HIP kernel)
  TreeNode       (:94-130)
  TreeCache      (:147-403)  init_prompt / new_node / alloc / merge_nodes /
                             reset_node_KV / branch / cut / add_ref / remove_ref
  BLOCK_CONFIG   (:587)
  TreeMetadata   (:591-881)  from_tree_cache = KV-guided grouping + flattened split
  register_* / get_global_* (:1021-1052)

Out of scope (and rejected loudly): unpaged KV (`use_paged_memory=False`) and the
WIP tree-index mode (`use_tree_index=True`).

MI355X-side differences (results identical):
  * `from_tree_cache` runs in the native builder of libdeft_amd.so
    (`deft_md_build`, deft_amd/csrc/host.cpp) and ships all twelve int64 arrays
    to the GPU in ONE host-to-device copy; the reference walks Python sets and
    issues ~12 separate `torch.tensor(..., device="cuda")` copies per step.
  * `alloc()` takes slots from the host-side allocator and writes the page table
    with one batched index_put instead of a `.item()` + scalar store per leaf.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Set

import numpy as np
import torch

from ._lib import check, lib
from .memory_pool import ReqToTokenPool, TokenToKVPool

BLOCK_CONFIG = {"BLOCK_LEN": 128, "MAX_BLOCK_LEN": -1}  # tree_cache.py:587
TRAVERSAL_CONFIG = {"METHOD": "dfs"}  # tree_cache.py:588 (only DFS exists upstream)


class KVCacheUpdater:
    """tree_cache.py:52-91, paged branch: key_buffer[cache_loc] = k; value_buffer[cache_loc] = v."""

    def __init__(
        self,
        use_paged_memory: bool,
        token_to_kv_pool: Optional[TokenToKVPool],
        cache_loc: Optional[torch.Tensor],
        leaf_data=None,
        is_prompt: bool = False,
    ) -> None:
        assert use_paged_memory, "deft_amd covers --mem paged only"
        self.use_paged_memory = use_paged_memory
        self.token_to_kv_pool = token_to_kv_pool
        self.cache_loc = cache_loc
        self.unpaged_cache = leaf_data
        self.is_prompt = is_prompt

    def update(self, layer_id: int, cache_k: torch.Tensor, cache_v: torch.Tensor) -> None:
        assert self.token_to_kv_pool is not None
        assert self.cache_loc is not None
        from .tree_attention import kv_append

        kv_append(self.token_to_kv_pool.kv_data[layer_id], self.cache_loc, cache_k, cache_v)


class TreeNode:
    """tree_cache.py:94-130."""

    def __init__(self, id: int, node_indices_id: Optional[int] = None, node_indices=None) -> None:
        self.id = id
        self.children: Dict[int, "TreeNode"] = {}
        self.token_ids: List[int] = []
        self.positions: List[int] = []
        self.position_offset = 0
        self.kv_indices: List[int] = []
        self.kv_data = None
        self.parent: Optional["TreeNode"] = None
        self.refs: Set["TreeNode"] = set()
        self.paused = False
        self.node_indices_id = node_indices_id
        self.node_indices = node_indices
        self.cumulative_logprob = 0.0

    def get_len(self) -> int:
        return len(self.token_ids)

    def append_token(self, token: int, logprob: Optional[float] = None) -> None:
        self.positions.append(self.position_offset + len(self.token_ids))
        self.token_ids.append(token)
        if logprob is not None:
            self.cumulative_logprob += logprob

    def append_index(self, index: int) -> None:
        self.kv_indices.append(index)


class TreeCache:
    def __init__(
        self,
        dtype: torch.dtype,
        head_num: int,
        head_dim: int,
        layer_num: int,
        req_to_token_pool: Optional[ReqToTokenPool],
        token_to_kv_pool: Optional[TokenToKVPool],
        tree_index_pool=None,
        use_paged_memory: bool = True,
        use_tree_index: bool = False,
    ) -> None:
        if not use_paged_memory:
            raise NotImplementedError("deft_amd covers --mem paged only (unpaged KV is out of scope)")
        if use_tree_index:
            raise NotImplementedError("tree_index mode is WIP upstream and out of scope here")
        assert token_to_kv_pool is not None
        assert req_to_token_pool is not None
        self.node_cnt = 1
        self.root: Optional[TreeNode] = None
        self.nodes: Dict[int, TreeNode] = {}
        self.leaves: Dict[int, TreeNode] = {}
        self.leaf_to_req: Dict[int, int] = {}
        self.paused_nodes: Set[int] = set()
        self.leaf_to_q: Dict[int, int] = {}
        self.req_to_token_pool = req_to_token_pool
        self.token_to_kv_pool = token_to_kv_pool
        self.tree_index_pool = None
        self.use_paged_memory = True
        self.use_tree_index = False
        self.layer_num = layer_num
        self.deleted_token_num = 0
        # native mirror of the tree (csrc/host.cpp): mutations are forwarded so that from_tree_cache does not
        # re-marshal every node's slot list each decode step
        self._native = int(lib.deft_tree_create())

    def __del__(self):  # noqa: D105
        h = getattr(self, "_native", 0)
        if h:
            try:
                lib.deft_tree_free(h)
            except Exception:
                pass
            self._native = 0

    def _mirror(self, rc: int, what: str) -> None:
        check(rc, what)

    # ---- :192-230 -------------------------------------------------------------
    def init_prompt(self, prompt_ids) -> KVCacheUpdater:
        ids = [int(t) for t in torch.as_tensor(prompt_ids).reshape(-1).tolist()]
        self.root = TreeNode(0)
        self.nodes[0] = self.root
        self.root.token_ids = ids
        self.root.position_offset = 0
        self.root.positions = list(range(len(ids)))
        self.leaves[self.root.id] = self.root
        self.add_ref(self.root)

        req = self.req_to_token_pool.alloc(1)
        assert req is not None
        req_id = int(req[0])
        self.leaf_to_req[self.root.id] = req_id
        loc = self.token_to_kv_pool.alloc_host(len(ids))
        assert loc is not None
        self.root.kv_indices = loc.tolist()
        self._mirror(lib.deft_tree_add_node(self._native, 0, -1), "deft_tree_add_node")
        self._mirror(lib.deft_tree_set_leaf(self._native, 0, 1), "deft_tree_set_leaf")
        loc64 = np.ascontiguousarray(loc, dtype=np.int64)
        self._mirror(lib.deft_tree_extend_node(self._native, 0, len(loc64), loc64.ctypes.data_as(C.c_void_p)),
                     "deft_tree_extend_node")
        cache_loc = torch.from_numpy(loc).to(self.token_to_kv_pool.device)
        self.req_to_token_pool.req_to_token[req_id, : len(ids)] = cache_loc.to(self.req_to_token_pool.req_to_token.device)
        return KVCacheUpdater(True, self.token_to_kv_pool, cache_loc, None, True)

    # ---- :242-259 -------------------------------------------------------------
    def new_node(self, parent: TreeNode) -> TreeNode:
        node = TreeNode(self.node_cnt)
        self.node_cnt += 1
        node.parent = parent
        node.position_offset = parent.position_offset + len(parent.positions)
        parent.children[node.id] = node
        self.nodes[node.id] = node
        self._mirror(lib.deft_tree_add_node(self._native, node.id, parent.id), "deft_tree_add_node")
        return node

    # ---- :261-283 -------------------------------------------------------------
    def alloc(self) -> KVCacheUpdater:
        loc = self.token_to_kv_pool.alloc_host(len(self.leaves))
        assert loc is not None
        reqs, poss, ids = [], [], []
        for idx, leaf in enumerate(sorted(self.leaves.values(), key=lambda x: x.id)):
            leaf.append_index(int(loc[idx]))
            reqs.append(self.leaf_to_req[leaf.id])
            poss.append(leaf.positions[-1])
            ids.append(leaf.id)
        ids64 = np.asarray(ids, dtype=np.int64)
        loc64 = np.ascontiguousarray(loc, dtype=np.int64)
        self._mirror(lib.deft_tree_append_slots(self._native, len(ids), ids64.ctypes.data_as(C.c_void_p),
                                                loc64.ctypes.data_as(C.c_void_p)), "deft_tree_append_slots")
        cache_loc = torch.from_numpy(loc).to(self.token_to_kv_pool.device, non_blocking=True)
        table = self.req_to_token_pool.req_to_token
        idx = torch.from_numpy(np.asarray([reqs, poss], dtype=np.int64)).to(table.device, non_blocking=True)
        table[idx[0], idx[1]] = cache_loc.to(table.device)  # one batched page-table write
        return KVCacheUpdater(True, self.token_to_kv_pool, cache_loc, None, False)

    # ---- :300-336 -------------------------------------------------------------
    def merge_nodes(self, node_A: TreeNode, node_B: TreeNode, pruneB_flag: Optional[bool] = True) -> None:
        for token_id in node_B.token_ids:
            node_A.positions.append(node_A.position_offset + len(node_A.token_ids))
            node_A.append_token(token=token_id)
        for kv_idx in node_B.kv_indices:
            node_A.append_index(index=kv_idx)
        b64 = np.asarray(node_B.kv_indices, dtype=np.int64)
        self._mirror(lib.deft_tree_extend_node(self._native, node_A.id, len(b64), b64.ctypes.data_as(C.c_void_p)),
                     "deft_tree_extend_node")
        self.token_to_kv_pool.add_refs(node_B.kv_indices)
        if pruneB_flag:
            self.cut(node_B)

    def reset_node_KV(self, node: TreeNode, diff: int) -> None:
        self.token_to_kv_pool.free(node.kv_indices)
        node.kv_indices = []
        self._mirror(lib.deft_tree_clear_node_kv(self._native, node.id), "deft_tree_clear_node_kv")
        node.position_offset += diff
        node.positions = [pos + diff for pos in node.positions]

    # ---- :338-370 -------------------------------------------------------------
    def branch(self, node: TreeNode, branch_cnt: int) -> List[TreeNode]:
        assert node.id in self.leaves
        self.leaves.pop(node.id)
        self._mirror(lib.deft_tree_set_leaf(self._native, node.id, 0), "deft_tree_set_leaf")
        path_len = node.positions[-1] + 1
        req = self.leaf_to_req.pop(node.id)
        is_first = True
        new_nodes: List[TreeNode] = []
        for _ in range(branch_cnt):
            child = self.new_node(node)
            new_nodes.append(child)
            self.leaves[child.id] = child
            self._mirror(lib.deft_tree_set_leaf(self._native, child.id, 1), "deft_tree_set_leaf")
            if is_first:
                self.leaf_to_req[child.id] = req
                is_first = False
            else:
                new_req = self.req_to_token_pool.alloc(1)
                assert new_req is not None
                new_req_id = int(new_req[0])
                self.req_to_token_pool.copy(req, new_req_id, path_len)
                self.leaf_to_req[child.id] = new_req_id
        self.remove_ref(node)
        for child in new_nodes:
            self.add_ref(child)
        return new_nodes

    # ---- :373-403 -------------------------------------------------------------
    def cut(self, node: TreeNode, record_deleted: bool = False) -> List[TreeNode]:
        assert len(node.children) == 0
        assert node.id in self.leaves
        self.leaves.pop(node.id)
        self.remove_ref(node)
        req = self.leaf_to_req.pop(node.id)
        self.req_to_token_pool.free(req)
        assert len(node.refs) == 0
        deleted_nodes = []
        cur: Optional[TreeNode] = node
        while cur is not None and len(cur.refs) == 0:
            deleted_nodes.append(self.nodes.pop(cur.id))
            self._mirror(lib.deft_tree_remove_node(self._native, cur.id), "deft_tree_remove_node")
            self.token_to_kv_pool.free(cur.kv_indices)
            parent = cur.parent
            if parent is not None:
                parent.children.pop(cur.id)
            cur = parent
        if record_deleted:
            for deleted in deleted_nodes:
                self.deleted_token_num += len(deleted.token_ids)
        return deleted_nodes

    # ---- :504-516 -------------------------------------------------------------
    def add_ref(self, node: TreeNode) -> None:
        ref = node
        node.refs.add(ref)
        while node.parent is not None:
            node = node.parent
            node.refs.add(ref)

    def remove_ref(self, node: TreeNode) -> None:
        ref = node
        node.refs.remove(ref)
        while node.parent is not None:
            node = node.parent
            node.refs.remove(ref)

    def free(self) -> None:  # :518-523
        self.root = None
        self.nodes.clear()
        self.leaves.clear()
        self.node_cnt = 0
        lib.deft_tree_free(self._native)
        self._native = int(lib.deft_tree_create())

    def get_tree_token_number(self) -> int:  # :569-584
        return sum(len(n.token_ids) for n in self.nodes.values()) + self.deleted_token_num

    def leaf_path_slots(self, leaf: TreeNode) -> List[int]:
        """Root->leaf pool slots (what sequential attention over this leaf reads)."""
        chain = []
        cur: Optional[TreeNode] = leaf
        while cur is not None:
            chain.append(cur)
            cur = cur.parent
        out: List[int] = []
        for n in reversed(chain):
            out.extend(n.kv_indices)
        return out


_FIELDS = (
    "node_q", "node_kv", "node_q_len", "node_kv_len", "node_q_offset", "node_kv_offset",
    "block_q", "block_q_cnts", "block_q_offset", "block_bitmasks", "block_kv", "block_lens",
)


def _mirror_consistent(tree: TreeCache) -> bool:
    """The mirror sees every mutation made through TreeCache's methods; code that edits `node.kv_indices` or
    `tree.leaves` directly (the reference's scripts are free to) is caught here by counts and sends the build
    down the marshalling path."""
    if not getattr(tree, "_native", 0):
        return False
    stats = np.zeros(3, dtype=np.int64)
    if lib.deft_tree_stats(tree._native, stats.ctypes.data_as(C.c_void_p)) != 0:
        return False
    return (int(stats[0]) == len(tree.nodes) and int(stats[1]) == len(tree.leaves)
            and int(stats[2]) == sum(len(nd.kv_indices) for nd in tree.nodes.values()))


def _marshal_and_build(tree: TreeCache, max_q_len: int, block_len: int, max_block_len: int) -> int:
    nodes = list(tree.nodes.values())
    n = len(nodes)
    node_id = np.fromiter((nd.id for nd in nodes), dtype=np.int64, count=n)
    parent_id = np.fromiter((nd.parent.id if nd.parent is not None else -1 for nd in nodes), dtype=np.int64, count=n)
    is_leaf = np.fromiter((nd.id in tree.leaves for nd in nodes), dtype=np.uint8, count=n)
    kv_offset = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(nd.kv_indices) for nd in nodes], out=kv_offset[1:])
    kv_slots = np.empty(int(kv_offset[-1]), dtype=np.int64)
    for i, nd in enumerate(nodes):
        kv_slots[kv_offset[i] : kv_offset[i + 1]] = nd.kv_indices

    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    return int(lib.deft_md_build(n, ptr(node_id), ptr(parent_id), ptr(is_leaf), ptr(kv_offset), ptr(kv_slots),
                                 int(max_q_len), int(block_len), int(max_block_len)))


def build_metadata_host(tree: TreeCache, max_q_len: int, block_len: int, max_block_len: int,
                        use_mirror: bool = True, alloc=None) -> Dict[str, object]:
    """Run the native builder; returns numpy int64 arrays that alias ONE packed buffer."""
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    if use_mirror and _mirror_consistent(tree):
        handle = int(lib.deft_tree_build_md(tree._native, int(max_q_len), int(block_len), int(max_block_len)))
    else:
        handle = _marshal_and_build(tree, max_q_len, block_len, max_block_len)
    if handle <= 0:
        check(int(handle), "deft_md_build")
    try:
        sizes = np.zeros(8, dtype=np.int64)
        check(lib.deft_md_sizes(handle, ptr(sizes)), "deft_md_sizes")
        query_num, NE, total_kv, n_node_q, n_node_kv, NB, P, n_block_kv = (int(x) for x in sizes)
        lens = {
            "node_q": n_node_q, "node_kv": n_node_kv, "node_q_len": NE, "node_kv_len": NE,
            "node_q_offset": NE, "node_kv_offset": NE,
            "block_q": P, "block_q_cnts": NB, "block_q_offset": NB,
            "block_bitmasks": n_block_kv, "block_kv": n_block_kv, "block_lens": NB,
        }
        total = sum(lens.values())
        packed = np.empty(total, dtype=np.int64) if alloc is None else alloc(total)  # e.g. a pinned staging buffer
        views, off = {}, 0
        for k in _FIELDS:
            views[k] = packed[off : off + lens[k]]
            off += lens[k]
        leaf_ids = np.empty(query_num, dtype=np.int64)
        check(lib.deft_md_fetch(handle, *[ptr(views[k]) for k in _FIELDS], ptr(leaf_ids)), "deft_md_fetch")
    finally:
        lib.deft_md_free(handle)
    out: Dict[str, object] = dict(views)
    out.update(query_num=query_num, node_num=NE, total_kv_len=total_kv, block_len=block_len,
               leaf_to_q={int(l): i for i, l in enumerate(leaf_ids)}, _packed=packed, _lens=lens)
    return out


@dataclass
class TreeMetadata:
    """tree_cache.py:591-616 — field names, dtypes (int64) and meaning unchanged."""

    query_num: int
    node_num: int
    total_kv_len: int
    leaf_to_q: Dict[int, int]
    node_q: torch.Tensor
    node_kv: torch.Tensor
    node_q_len: torch.Tensor
    node_kv_len: torch.Tensor
    node_q_offset: torch.Tensor
    node_kv_offset: torch.Tensor

    block_len: int

    block_q: torch.Tensor
    block_q_cnts: torch.Tensor
    block_q_offset: torch.Tensor
    block_bitmasks: torch.Tensor
    block_kv: torch.Tensor
    block_lens: torch.Tensor

    @classmethod
    def from_tree_cache(
        cls,
        tree: TreeCache,
        tile_num: int = 8,
        max_q_len: int = 32,
        max_block_len: int = -1,
        device: Optional[str] = None,
    ) -> "TreeMetadata":
        assert tree.root is not None
        block_len = BLOCK_CONFIG["BLOCK_LEN"]
        if max_block_len == -1:
            max_block_len = BLOCK_CONFIG["MAX_BLOCK_LEN"]
        dev = torch.device(device) if device is not None else tree.token_to_kv_pool.device
        if dev.type == "cpu":
            host = build_metadata_host(tree, max_q_len, block_len, max_block_len)
            packed = torch.from_numpy(host["_packed"])
        else:
            # the builder writes straight into a pinned staging buffer kept on the tree (two, alternating, each
            # guarded by the event of its last upload); ONE H2D copy for all twelve arrays
            stages = tree.__dict__.setdefault("_md_stages", [None, None])
            k = tree.__dict__["_md_stage_idx"] = 1 - tree.__dict__.get("_md_stage_idx", 0)

            def alloc(total: int):
                st = stages[k]
                if st is None or st[0].numel() < total:
                    st = stages[k] = [torch.empty(max(2 * total, 1 << 14), dtype=torch.int64).pin_memory(), None]
                if st[1] is not None:
                    st[1].synchronize()
                return st[0].numpy()[:total]

            host = build_metadata_host(tree, max_q_len, block_len, max_block_len, alloc=alloc)
            st = stages[k]
            packed = st[0][: host["_packed"].shape[0]].to(dev, non_blocking=True)
            st[1] = torch.cuda.Event()
            st[1].record(torch.cuda.current_stream(dev))
        views, off = {}, 0
        for k in _FIELDS:
            n = host["_lens"][k]
            views[k] = packed[off : off + n]
            off += n
        return cls(
            query_num=host["query_num"], node_num=host["node_num"], total_kv_len=host["total_kv_len"],
            leaf_to_q=host["leaf_to_q"], block_len=block_len, **views,
        )


GLOBAL_TREE_METADATA: Optional[TreeMetadata] = None
GLOBAL_TREE_CACHE: Optional[TreeCache] = None


def register_tree_metadata(tree_metadata: TreeMetadata) -> None:
    global GLOBAL_TREE_METADATA
    GLOBAL_TREE_METADATA = tree_metadata


def unregister_tree_metadata() -> None:
    global GLOBAL_TREE_METADATA
    GLOBAL_TREE_METADATA = None


def get_global_tree_metadata() -> TreeMetadata:
    assert GLOBAL_TREE_METADATA is not None
    return GLOBAL_TREE_METADATA


def register_tree_cache(tree_cache: TreeCache) -> None:
    global GLOBAL_TREE_CACHE
    GLOBAL_TREE_CACHE = tree_cache


def unregister_tree_cache() -> None:
    global GLOBAL_TREE_CACHE
    GLOBAL_TREE_CACHE = None


def get_global_tree_cache() -> TreeCache:
    assert GLOBAL_TREE_CACHE is not None
    return GLOBAL_TREE_CACHE
