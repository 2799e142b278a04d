"""Decoding tree over the paged KV pool + the attention metadata contract.

The drop-in surface of the paged half of DeFT/deft/tree_decoding/tree_cache.py -- same
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

How it is built here (results identical, checked bit for bit against reference outputs):

  * The tree itself -- parents, children, pool slots, which nodes are live leaves, how many
    live leaves hang below every node -- is ONE native object (`deft_tree_*`,
    deft_amd/csrc/tree.cpp).  `TreeCache` translates the reference's calls into operations
    on it; `TreeNode.kv_indices`, `.refs` are views into it, the `nodes` / `leaves` dicts
    hold the Python handles.  Token ids and positions are host-side bookkeeping of the
    decode loop and stay on the handles.
  * `alloc()` takes its slots from the host-side allocator, appends them to all leaves in
    one native call and writes the page table with one batched index_put (the reference:
    a `.item()` + scalar store per leaf).
  * `TreeMetadata.from_tree_cache` on a GPU pool runs ON THE GPU: the tree has a compact
    device copy (`_DeviceTree`, deft_amd/csrc/tree_plan.h) that is uploaded when the
    STRUCTURE changes (branch / cut / merge) and advanced by a kernel on every `alloc()`;
    three kernels then emit the reference's twelve int64 arrays.  Per decode step nothing
    but the nq new slot numbers crosses PCIe.  On CPU pools (and with
    `device_build=False`) the native host builder `deft_md_build` produces the same
    arrays; the reference walks Python sets and issues ~12 H2D copies per step.
"""
from __future__ import annotations

import ctypes as C
from collections.abc import MutableSequence
from dataclasses import dataclass
from typing import Dict, List, Optional, Set

import math

import numpy as np
import torch

from ._lib import check, lib
from .memory_pool import ReqToTokenPool, TokenToKVPool

BLOCK_CONFIG = {"BLOCK_LEN": 128, "MAX_BLOCK_LEN": -1}  # tree_cache.py:587
TRAVERSAL_CONFIG = {"METHOD": "dfs"}  # tree_cache.py:588 (only DFS exists upstream)
DEVICE_METADATA = True  # GPU pools: from_tree_cache builds on the GPU (False: host builder + one upload, for A/B)

_ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731


class KVCacheUpdater:
    """tree_cache.py:52-91, paged branch: key_buffer[cache_loc] = k; value_buffer[cache_loc] = v."""

    def __init__(
        self,
        use_paged_memory: bool,
        token_to_kv_pool: Optional[TokenToKVPool],
        cache_loc: Optional[torch.Tensor],
        leaf_data,
        is_prompt: bool,
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


class _Slots(MutableSequence):
    """`node.kv_indices` of a node that belongs to a tree: the pool slots the native tree holds for it, in the
    order they were appended.  Reads fetch them, edits (scripts written against the reference append to and
    assign this list directly) are written through, so the tree never sees a stale slot list."""

    __slots__ = ("_h", "_id")

    def __init__(self, handle: int, node_id: int) -> None:
        self._h, self._id = handle, node_id

    def _get(self) -> np.ndarray:
        n = int(lib.deft_tree_node_len(self._h, self._id))
        if n < 0:
            check(n, "deft_tree_node_len")
        out = np.empty(n, dtype=np.int64)
        if n:
            lib.deft_tree_node_kv(self._h, self._id, _ptr(out), n)
        return out

    def _put(self, values) -> None:
        arr = np.ascontiguousarray(values, dtype=np.int64).reshape(-1)
        check(lib.deft_tree_set_node_kv(self._h, self._id, len(arr), _ptr(arr)), "deft_tree_set_node_kv")

    def __len__(self) -> int:
        n = int(lib.deft_tree_node_len(self._h, self._id))
        if n < 0:
            check(n, "deft_tree_node_len")
        return n

    def __getitem__(self, i):
        v = self._get()[i]
        return v.tolist() if isinstance(i, slice) else int(v)

    def __setitem__(self, i, value) -> None:
        cur = self._get().tolist()
        cur[i] = value
        self._put(cur)

    def __delitem__(self, i) -> None:
        cur = self._get().tolist()
        del cur[i]
        self._put(cur)

    def insert(self, i: int, value) -> None:
        cur = self._get().tolist()
        cur.insert(i, int(value))
        self._put(cur)

    def append(self, value) -> None:  # (the common edit: one native call, no round trip of the list)
        v = np.asarray([int(value)], dtype=np.int64)
        check(lib.deft_tree_extend_node(self._h, self._id, 1, _ptr(v)), "deft_tree_extend_node")

    def extend(self, values) -> None:
        v = np.ascontiguousarray(list(values), dtype=np.int64)
        check(lib.deft_tree_extend_node(self._h, self._id, len(v), _ptr(v)), "deft_tree_extend_node")

    def __iter__(self):
        return iter(self._get().tolist())

    def tolist(self) -> List[int]:
        return self._get().tolist()

    def __eq__(self, other) -> bool:
        try:
            return self._get().tolist() == list(other)
        except TypeError:
            return NotImplemented

    def __repr__(self) -> str:
        return repr(self._get().tolist())


class _ShiftedPositions(MutableSequence):
    """`node.positions` after `reset_nodes_KV`: the reference shifts every position of every leaf by the accepted length at every
    speculative-decoding step (`[pos + diff for pos in node.positions]`, tree_cache.py:332-336) -- lists that grow by a token per
    step, so the loop costs O(steps x leaves) per step.  Same values, the shift kept as ONE offset."""

    __slots__ = ("_base", "_off")

    def __init__(self, values, offset: int = 0) -> None:
        self._base, self._off = list(values), int(offset)

    def shift(self, d: int) -> None:
        self._off += int(d)

    def __len__(self) -> int:
        return len(self._base)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [v + self._off for v in self._base[i]]
        return self._base[i] + self._off

    def __setitem__(self, i, value) -> None:
        if isinstance(i, slice):
            self._base[i] = [int(v) - self._off for v in value]
        else:
            self._base[i] = int(value) - self._off

    def __delitem__(self, i) -> None:
        del self._base[i]

    def insert(self, i: int, value) -> None:
        self._base.insert(i, int(value) - self._off)

    def append(self, value) -> None:
        self._base.append(int(value) - self._off)

    def __iter__(self):
        off = self._off
        return (v + off for v in self._base)

    def __eq__(self, other) -> bool:
        try:
            return list(self) == list(other)
        except TypeError:
            return NotImplemented

    def __add__(self, other):
        return list(self) + list(other)

    def __repr__(self) -> str:
        return repr(list(self))


class TreeNode:
    """tree_cache.py:94-130: a handle.  Token ids / positions live here; the slots and the leaf set are read from
    the tree the node belongs to."""

    def __init__(self, id: int, node_indices_id: Optional[int] = None, node_indices=None) -> None:
        self.id = id
        self.children: Dict[int, "TreeNode"] = {}
        self.token_ids: List[int] = []
        self.positions: List[int] = []
        self.position_offset = 0
        self.kv_data = None
        self.parent: Optional["TreeNode"] = None
        self.paused = False
        self.node_indices_id = node_indices_id
        self.node_indices = node_indices
        self.cumulative_logprob = 0.0
        self._tree: Optional["TreeCache"] = None
        self._local_kv: List[int] = []  # a node outside any tree keeps an ordinary list

    def get_len(self) -> int:
        return len(self.token_ids)

    def append_token(self, token: int, logprob: Optional[float] = None) -> None:
        self.positions.append(self.position_offset + len(self.token_ids))
        self.token_ids.append(token)
        if logprob is not None:
            self.cumulative_logprob += logprob

    def append_index(self, index: int) -> None:
        self.kv_indices.append(index)

    @property
    def kv_indices(self):
        t = self._tree
        return _Slots(t._native, self.id) if t is not None and t._native else self._local_kv

    @kv_indices.setter
    def kv_indices(self, values) -> None:
        t = self._tree
        if t is not None and t._native:
            _Slots(t._native, self.id)._put(list(values))
        else:
            self._local_kv = list(values)

    @property
    def refs(self) -> Set["TreeNode"]:
        """The live leaves below (or at) this node (tree_cache.py:504-516 keeps them as a set per node)."""
        t = self._tree
        if t is None or not t._native:
            return set()
        n = int(lib.deft_tree_node_refs(t._native, self.id, None, 0))
        if n < 0:
            return set()
        out = np.empty(max(n, 1), dtype=np.int64)
        lib.deft_tree_node_refs(t._native, self.id, _ptr(out), n)
        return {t.nodes[int(i)] for i in out[:n] if int(i) in t.nodes}


class BranchSequence:
    """tree_cache.py:132-144: one finished branch of the decoding tree (tokens, cumulative log-probability, perplexity)."""

    def __init__(self, id: int):
        self.id = id
        self.token_ids: List[int] = []
        self.cumulative_logprob = 0.0
        self.PPL = 0.0

    def get_len(self) -> int:
        return len(self.token_ids)

    def append_tokens(self, tokens: List[int]) -> None:
        self.token_ids.extend(tokens)


class TreeCache:
    def __init__(
        self,
        dtype: torch.dtype,
        head_num: int,
        head_dim: int,
        layer_num: int,
        req_to_token_pool: Optional[ReqToTokenPool],
        token_to_kv_pool: Optional[TokenToKVPool],
        tree_index_pool,
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
        self.all_finished_seqs: List[BranchSequence] = []  # finished branches, in the order they were output (:186-188)
        self._native = int(lib.deft_tree_create())
        self._device_tree: Optional["_DeviceTree"] = None

    def __del__(self):  # noqa: D105
        h = getattr(self, "_native", 0)
        if h:
            try:
                lib.deft_tree_free(h)
            except Exception:
                pass
            self._native = 0

    # ---- helpers ------------------------------------------------------------------------------------
    def _handle(self, node_id: int, parent: Optional[TreeNode]) -> TreeNode:
        node = TreeNode(node_id)
        node._tree = self
        node.parent = parent
        if parent is not None:
            node.position_offset = parent.position_offset + len(parent.positions)
            parent.children[node_id] = node
        self.nodes[node_id] = node
        return node

    def _epoch(self) -> int:
        st = np.zeros(4, dtype=np.int64)
        check(lib.deft_tree_stats(self._native, _ptr(st)), "deft_tree_stats")
        return int(st[3])

    def _consistent(self) -> bool:
        """The dicts are handles onto the native tree; code that pops / adds entries behind TreeCache's back is caught
        by the counts (slot lists cannot go stale: `kv_indices` is a view)."""
        st = np.zeros(4, dtype=np.int64)
        if not self._native or lib.deft_tree_stats(self._native, _ptr(st)) != 0:
            return False
        return int(st[0]) == len(self.nodes) and int(st[1]) == len(self.leaves)

    # ---- :192-230 -------------------------------------------------------------
    def init_prompt(self, prompt_ids) -> KVCacheUpdater:
        ids = [int(t) for t in torch.as_tensor(prompt_ids).reshape(-1).tolist()]
        check(lib.deft_tree_add_node(self._native, 0, -1), "deft_tree_add_node")
        check(lib.deft_tree_set_leaf(self._native, 0, 1), "deft_tree_set_leaf")
        self.root = self._handle(0, None)
        self.root.token_ids = ids
        self.root.positions = list(range(len(ids)))
        self.leaves[0] = self.root

        req = self.req_to_token_pool.alloc(1)
        assert req is not None
        req_id = int(req[0])
        self.leaf_to_req[0] = req_id
        loc = self.token_to_kv_pool.alloc_host(len(ids))
        assert loc is not None
        self.root.kv_indices = loc
        cache_loc = torch.from_numpy(loc).to(self.token_to_kv_pool.device)
        table = self.req_to_token_pool.req_to_token
        table[req_id, : len(ids)] = cache_loc.to(table.device)
        return KVCacheUpdater(True, self.token_to_kv_pool, cache_loc, None, True)

    def init_forest(self, prompts) -> KVCacheUpdater:
        """A batch of INDEPENDENT trees as one tree object (not in the reference, which decodes one tree per process): a root
        without tokens whose children are the trees' roots, child t holding `prompts[t]`.  Trees share nothing -- no KV
        slot lies on two trees' paths -- so every operator, the metadata kernels on the GPU and `DecodeSession` serve the
        batch unchanged; the Flatten blocks simply keep packing across tree boundaries (the per-slot query masks keep a
        query off another tree's keys).  Query rows: live leaves by ascending node id, as always.  Returns the updater
        of all prompt tokens (tree 0's slots, then tree 1's, ...), like `init_prompt`."""
        self.init_prompt(torch.empty(0, dtype=torch.int32))
        kids = self.branch(self.root, len(prompts))
        locs = []
        for kid, prompt in zip(kids, prompts):
            locs.append(self.extend_leaf(kid, prompt).cache_loc)
        return KVCacheUpdater(True, self.token_to_kv_pool, torch.cat(locs) if locs else None, None, True)

    def extend_leaf(self, leaf: TreeNode, token_ids) -> KVCacheUpdater:
        """`token_ids` appended to a live leaf in one go, with their pool slots and page-table entries (what `init_prompt`
        does for the root; `append_token` + `alloc` would take a decode step per token)."""
        assert leaf.id in self.leaves
        ids = [int(t) for t in torch.as_tensor(token_ids).reshape(-1).tolist()]
        n = len(ids)
        start = leaf.position_offset + len(leaf.token_ids)
        loc = self.token_to_kv_pool.alloc_host(n)
        assert loc is not None
        loc64 = loc.astype(np.int64)
        check(lib.deft_tree_extend_node(self._native, leaf.id, n, _ptr(loc64)), "deft_tree_extend_node")
        leaf.token_ids.extend(ids)
        leaf.positions.extend(range(start, start + n))
        cache_loc = torch.from_numpy(loc).to(self.token_to_kv_pool.device)
        table = self.req_to_token_pool.req_to_token
        table[self.leaf_to_req[leaf.id], start : start + n] = cache_loc.to(table.device)
        return KVCacheUpdater(True, self.token_to_kv_pool, cache_loc, None, True)

    # ---- :242-259 -------------------------------------------------------------
    def new_node(self, parent: TreeNode) -> TreeNode:
        node_id = self.node_cnt
        self.node_cnt += 1
        check(lib.deft_tree_add_node(self._native, node_id, parent.id), "deft_tree_add_node")
        return self._handle(node_id, parent)

    # ---- :261-283 -------------------------------------------------------------
    def _staging(self, n: int):
        """Pinned staging for one step's uploads (slot numbers, page-table coordinates): a pageable source would make
        the copy wait for the stream to drain -- the host would run in lock-step with the GPU.  Four buffers in rotation,
        each guarded by the event of its last upload."""
        ring = self.__dict__.setdefault("_pin_ring", [])
        k = self.__dict__["_pin_idx"] = (self.__dict__.get("_pin_idx", -1) + 1) % 4
        while len(ring) <= k:
            ring.append(None)
        ent = ring[k]
        if ent is None or ent[0].numel() < n:
            cap = max(64, 2 * n)
            ent = ring[k] = [torch.empty(cap, dtype=torch.int32).pin_memory(), torch.empty((2, cap), dtype=torch.int64).pin_memory(), None]
        if ent[2] is not None:
            ent[2].synchronize()
        return ent

    def alloc(self) -> KVCacheUpdater:
        """One pool slot per live leaf, leaves in id order (what the reference's loop over sorted leaves does)."""
        n = len(self.leaves)
        loc = self.token_to_kv_pool.alloc_host(n)
        assert loc is not None
        dev = self._device_tree
        synced = dev is not None and dev.epoch == self._epoch()
        loc64 = loc.astype(np.int64)
        check(lib.deft_tree_alloc_step(self._native, n, _ptr(loc64)), "deft_tree_alloc_step")
        order = sorted(self.leaves)
        table = self.req_to_token_pool.req_to_token
        pool_dev = self.token_to_kv_pool.device
        if pool_dev.type == "cuda":
            ent = self._staging(n)
            ent[0].numpy()[:n] = loc
            idx_h = ent[1].numpy()
            idx_h[0, :n] = [self.leaf_to_req[i] for i in order]
            idx_h[1, :n] = [self.leaves[i].positions[-1] for i in order]
            cache_loc = ent[0][:n].to(pool_dev, non_blocking=True)
            idx = ent[1][:, :n].to(table.device, non_blocking=True)
            ent[2] = torch.cuda.Event()
            ent[2].record(torch.cuda.current_stream(pool_dev))
        else:
            cache_loc = torch.from_numpy(loc)
            idx = torch.from_numpy(np.asarray([[self.leaf_to_req[i] for i in order],
                                               [self.leaves[i].positions[-1] for i in order]], dtype=np.int64)).to(table.device)
        if synced and dev.epoch == self._epoch():
            # absorbed changes since the last step -- merge_nodes / reset_node_KV -- come first; then the device copy of the tree
            # appends the same slots itself, UNLESS the journal was too long to replay and the copy was uploaded afresh: that image
            # was made after deft_tree_alloc_step and already ends in this step's slots
            if dev.apply_journal() >= 0:
                dev.advance(cache_loc)
        table[idx[0], idx[1]] = cache_loc.to(table.device)  # one batched page-table write
        return KVCacheUpdater(True, self.token_to_kv_pool, cache_loc, None, False)

    # ---- :300-336 -------------------------------------------------------------
    def merge_nodes(self, node_A: TreeNode, node_B: TreeNode, pruneB_flag: Optional[bool] = True) -> None:
        """node_A takes over node_B's tokens and slots (speculative decoding: accepted tokens move into the root);
        the slots gain a reference, so pruning B afterwards does not free them."""
        moved = node_B.kv_indices.tolist()
        for token_id in node_B.token_ids:
            # (the reference records the position twice -- here and inside append_token, tree_cache.py:307-311 --
            #  and later position arithmetic on the node depends on it)
            node_A.positions.append(node_A.position_offset + len(node_A.token_ids))
            node_A.append_token(token=token_id)
        node_A.kv_indices.extend(moved)
        self.token_to_kv_pool.add_refs(moved)
        if pruneB_flag:
            self.cut(node_B)

    def reset_node_KV(self, node: TreeNode, diff: int) -> None:
        self.token_to_kv_pool.free(node.kv_indices.tolist())
        node.kv_indices = []
        node.position_offset += diff
        node.positions = [pos + diff for pos in node.positions]

    def reset_nodes_KV(self, nodes: List[TreeNode], diff: int) -> None:
        """`reset_node_KV` for many nodes at once (not in the reference, whose speculative-decoding mock calls it leaf by
        leaf, branch_func_example.py:430-436): the slots of all nodes are released by ONE refcount update -- per node the
        work is a slot fetch and a clear in the native tree, no numpy call."""
        if not nodes:
            return
        ids = np.fromiter((n.id for n in nodes), dtype=np.int64, count=len(nodes))
        st = np.zeros(4, dtype=np.int64)
        check(lib.deft_tree_stats(self._native, _ptr(st)), "deft_tree_stats")
        cap = max(int(st[2]), 1)  # every slot of the tree: no subset of nodes holds more
        buf = np.empty(cap, dtype=np.int64)
        total = int(lib.deft_tree_take_nodes_kv(self._native, len(nodes), _ptr(ids), _ptr(buf), cap))
        if total < 0:
            check(total, "deft_tree_take_nodes_kv")
        assert total <= cap, "slot lists longer than the tree"
        for n in nodes:
            n.position_offset += diff
            if isinstance(n.positions, _ShiftedPositions):
                n.positions.shift(diff)
            else:
                n.positions = _ShiftedPositions(n.positions, diff)
        if total:
            self.token_to_kv_pool.free(buf[:total])

    # ---- :338-370 -------------------------------------------------------------
    def branch(self, node: TreeNode, branch_cnt: int) -> List[TreeNode]:
        assert node.id in self.leaves
        first = self.node_cnt
        check(lib.deft_tree_branch(self._native, node.id, branch_cnt, first), "deft_tree_branch")
        self.node_cnt += branch_cnt
        self.leaves.pop(node.id)
        req = self.leaf_to_req.pop(node.id)
        path_len = node.positions[-1] + 1 if node.positions else node.position_offset  # (a node without tokens: a forest's virtual root)
        kids = [self._handle(first + i, node) for i in range(branch_cnt)]
        for i, child in enumerate(kids):
            self.leaves[child.id] = child
            if i == 0:  # the first child inherits the parent's page-table row, the others get a copy of it
                self.leaf_to_req[child.id] = req
                continue
            row = self.req_to_token_pool.alloc(1)
            assert row is not None
            self.req_to_token_pool.copy(req, int(row[0]), path_len)
            self.leaf_to_req[child.id] = int(row[0])
        return kids

    # ---- :373-403 -------------------------------------------------------------
    def cut(self, node: TreeNode, record_deleted: bool = False) -> List[TreeNode]:
        assert len(node.children) == 0
        assert node.id in self.leaves
        st = np.zeros(4, dtype=np.int64)
        check(lib.deft_tree_stats(self._native, _ptr(st)), "deft_tree_stats")
        ids = np.empty(max(int(st[0]), 1), dtype=np.int64)
        slots = np.empty(max(int(st[2]), 1), dtype=np.int64)
        n_ids, n_slots = C.c_int(0), C.c_int64(0)
        check(lib.deft_tree_cut(self._native, node.id, _ptr(ids), len(ids), C.byref(n_ids), _ptr(slots), len(slots),
                                C.byref(n_slots)), "deft_tree_cut")
        self.leaves.pop(node.id)
        self.req_to_token_pool.free(self.leaf_to_req.pop(node.id))
        self.token_to_kv_pool.free(slots[: n_slots.value])
        gone = []
        for i in ids[: n_ids.value].tolist():  # the leaf, then every ancestor it was the last live leaf of
            h = self.nodes.pop(i)
            if h.parent is not None:
                h.parent.children.pop(i, None)
            h._local_kv = []
            h._tree = None
            gone.append(h)
            if record_deleted:
                self.deleted_token_num += len(h.token_ids)
        return gone

    # ---- :504-516 -------------------------------------------------------------
    def add_ref(self, node: TreeNode) -> None:
        """The reference adds `node` to the `refs` set of every ancestor; here: the node counts as a live leaf."""
        check(lib.deft_tree_set_leaf(self._native, node.id, 1), "deft_tree_set_leaf")

    def remove_ref(self, node: TreeNode) -> None:
        check(lib.deft_tree_set_leaf(self._native, node.id, 0), "deft_tree_set_leaf")

    def free(self) -> None:  # :518-523
        self.root = None
        self.nodes.clear()
        self.leaves.clear()
        self.node_cnt = 0
        lib.deft_tree_free(self._native)
        self._native = int(lib.deft_tree_create())
        self._device_tree = None

    def output_branch(self, dstnode: TreeNode) -> None:
        """tree_cache.py:525-541: record the branch root -> `dstnode` (the root's own tokens -- the prompt -- excluded, as
        `_find_path_to_node` stops below the root) with its cumulative log-probability and perplexity."""
        branch_seq = BranchSequence(len(self.all_finished_seqs))
        for node in self._find_path_to_node(dstnode):
            branch_seq.append_tokens(node.token_ids)
            branch_seq.cumulative_logprob += node.cumulative_logprob
        branch_seq.PPL = math.exp(-branch_seq.cumulative_logprob / len(branch_seq.token_ids))
        self.all_finished_seqs.append(branch_seq)

    def _find_path_to_node(self, dstnode: TreeNode) -> List[TreeNode]:  # :543-550
        path = []
        node = dstnode
        while node.parent is not None:
            path.append(node)
            node = node.parent
        path.reverse()
        return path

    def print_finished_branches(self, tokenizer) -> None:
        """tree_cache.py:552-567: every finished branch decoded and printed (`tokenizer` = anything with the Hugging Face
        `decode(token_ids, skip_special_tokens=True)`)."""
        print(f"Total number of generated branches={len(self.all_finished_seqs)}! \n")
        for branch in self.all_finished_seqs:
            generated_text = tokenizer.decode(branch.token_ids, skip_special_tokens=True)
            print(f" Branch ID: {branch.id}\n", f"Generated Text: {generated_text}\n", f"Tokens in this path:{branch.token_ids}\n",
                  f"Token length : {len(branch.token_ids)}\n", f"Perplexity: {branch.PPL}\n")

    def get_tree_token_number(self) -> int:  # :569-584
        return sum(len(n.token_ids) for n in self.nodes.values()) + self.deleted_token_num

    def leaf_path_slots(self, leaf: TreeNode) -> List[int]:
        """Root->leaf pool slots (what sequential attention over this leaf reads)."""
        n = int(lib.deft_tree_path_slots(self._native, leaf.id, None, 0))
        if n < 0:
            check(n, "deft_tree_path_slots")
        out = np.empty(max(n, 1), dtype=np.int64)
        lib.deft_tree_path_slots(self._native, leaf.id, _ptr(out), n)
        return out[:n].tolist()


_FIELDS = (
    "node_q", "node_kv", "node_q_len", "node_kv_len", "node_q_offset", "node_kv_offset",
    "block_q", "block_q_cnts", "block_q_offset", "block_bitmasks", "block_kv", "block_lens",
)


def _lens_from_sizes(sizes) -> Dict[str, int]:
    query_num, NE, total_kv, n_node_q, n_node_kv, NB, P, n_block_kv = (int(x) for x in sizes[:8])
    return {
        "node_q": n_node_q, "node_kv": n_node_kv, "node_q_len": NE, "node_kv_len": NE,
        "node_q_offset": NE, "node_kv_offset": NE,
        "block_q": P, "block_q_cnts": NB, "block_q_offset": NB,
        "block_bitmasks": n_block_kv, "block_kv": n_block_kv, "block_lens": NB,
    }


def _mirror_consistent(tree: TreeCache) -> bool:
    return tree._consistent()


def _marshal_and_build(tree: TreeCache, max_q_len: int, block_len: int, max_block_len: int) -> int:
    """deft_md_build on arrays marshalled from the Python-visible tree (the stateless entry point of the C ABI)."""
    nodes = list(tree.nodes.values())
    n = len(nodes)
    node_id = np.fromiter((nd.id for nd in nodes), dtype=np.int64, count=n)
    parent_id = np.fromiter((nd.parent.id if nd.parent is not None else -1 for nd in nodes), dtype=np.int64, count=n)
    is_leaf = np.fromiter((nd.id in tree.leaves for nd in nodes), dtype=np.uint8, count=n)
    lists = [nd.kv_indices.tolist() if isinstance(nd.kv_indices, _Slots) else list(nd.kv_indices) for nd in nodes]
    kv_offset = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(x) for x in lists], out=kv_offset[1:])
    kv_slots = np.empty(int(kv_offset[-1]), dtype=np.int64)
    for i, x in enumerate(lists):
        kv_slots[kv_offset[i] : kv_offset[i + 1]] = x
    return int(lib.deft_md_build(n, _ptr(node_id), _ptr(parent_id), _ptr(is_leaf), _ptr(kv_offset), _ptr(kv_slots),
                                 int(max_q_len), int(block_len), int(max_block_len)))


def build_metadata_host(tree: TreeCache, max_q_len: int, block_len: int, max_block_len: int,
                        use_mirror: bool = True, alloc=None) -> Dict[str, object]:
    """Run the native HOST builder; returns numpy int64 arrays that alias ONE packed buffer."""
    if not tree._consistent():
        raise RuntimeError("tree.nodes / tree.leaves were edited behind TreeCache's back: mutate the tree through "
                           "init_prompt / branch / alloc / cut / merge_nodes / reset_node_KV")
    if use_mirror:
        handle = int(lib.deft_tree_build_md(tree._native, int(max_q_len), int(block_len), int(max_block_len)))
    else:
        handle = _marshal_and_build(tree, max_q_len, block_len, max_block_len)
    if handle <= 0:
        check(int(handle), "deft_md_build")
    try:
        sizes = np.zeros(8, dtype=np.int64)
        check(lib.deft_md_sizes(handle, _ptr(sizes)), "deft_md_sizes")
        lens = _lens_from_sizes(sizes)
        total = sum(lens.values())
        packed = np.empty(total, dtype=np.int64) if alloc is None else alloc(total)  # e.g. a pinned staging buffer
        views, off = {}, 0
        for k in _FIELDS:
            views[k] = packed[off : off + lens[k]]
            off += lens[k]
        leaf_ids = np.empty(int(sizes[0]), dtype=np.int64)
        check(lib.deft_md_fetch(handle, *[_ptr(views[k]) for k in _FIELDS], _ptr(leaf_ids)), "deft_md_fetch")
    finally:
        lib.deft_md_free(handle)
    out: Dict[str, object] = dict(views)
    out.update(query_num=int(sizes[0]), node_num=int(sizes[1]), total_kv_len=int(sizes[2]), block_len=block_len,
               leaf_to_q={int(l): i for i, l in enumerate(leaf_ids)}, _packed=packed, _lens=lens)
    return out


class _DeviceTree:
    """The compact copy of a tree on the GPU and the buffers the metadata kernels write (deft_amd/csrc/tree_plan.h).

    One upload per structural EPOCH of the tree (branch / cut / merge / a leaf outgrowing its room): node table,
    leaf sets as bit sets, every node's slots.  Within an epoch `alloc()` advances the device copy with a kernel, and
    `build()` launches the three metadata kernels into buffers sized for the largest tree the epoch can hold."""

    SLACK = 256  # decode steps a leaf can grow before the tree is laid out again

    def __init__(self, tree: TreeCache, device: torch.device, max_q_len: int, block_len: int, max_block_len: int) -> None:
        self.tree, self.device = tree, device
        self.cfg = (int(max_q_len), int(block_len), int(max_block_len))
        self.epoch = -1
        self.version = 0   # bumped by everything that changes the device copy (upload, journal replay, advance): a DecodeSession on a
                           # window plan re-plans when somebody else moved the copy between two of its steps
        self.stage = None  # pinned staging buffer of the upload image
        self.stage_event = None

    def _upload(self) -> None:
        t, (mq, bl, mbl) = self.tree, self.cfg
        sizes = np.zeros(5, dtype=np.int64)
        check(lib.deft_tree_layout(t._native, self.SLACK, _ptr(sizes)), "deft_tree_layout")
        n, nq, nqw, total_cap, epoch = (int(x) for x in sizes)
        self.n, self.nq, self.nqw = n, nq, nqw
        # upload image: [node_start | node_len | node_cap | leaf_node | slots] int32, then refs uint64
        n32 = 3 * n + nq + total_cap
        n32 += n32 & 1  # keep the uint64 part 8-byte aligned
        words = n32 + 2 * n * nqw
        if self.stage is None or self.stage.numel() < words:
            self.stage = torch.empty(max(2 * words, 1 << 12), dtype=torch.int32).pin_memory()
        elif self.stage_event is not None:
            self.stage_event.synchronize()  # the previous upload has left the staging buffer
        img = self.stage.numpy()[:words]
        v_start, v_len, v_cap = img[0:n], img[n : 2 * n], img[2 * n : 3 * n]
        v_leaf, v_slots = img[3 * n : 3 * n + nq], img[3 * n + nq : 3 * n + nq + total_cap]
        v_refs = img[n32:words].view(np.uint64)
        check(lib.deft_tree_layout_fetch(t._native, _ptr(v_start), _ptr(v_len), _ptr(v_cap), _ptr(v_refs), _ptr(v_leaf),
                                         _ptr(v_slots)), "deft_tree_layout_fetch")
        self.h_leaf = v_leaf.copy()  # query row -> DFS index of its leaf (fixed for the epoch; DecodeSession's window bookkeeping)
        self.h_refs = v_refs.copy()  # ... and the nodes' leaf sets
        self.version += 1
        dev = self.stage[:words].to(self.device, non_blocking=True)
        self.stage_event = torch.cuda.Event()
        self.stage_event.record(torch.cuda.current_stream(self.device))
        self.image = dev
        base = dev.data_ptr()
        self.p_start, self.p_len, self.p_cap = base, base + 4 * n, base + 8 * n
        self.p_leaf, self.p_slots, self.p_refs = base + 12 * n, base + 4 * (3 * n + nq), base + 4 * n32
        # output buffers for the largest tree of this epoch (every leaf SLACK tokens longer)
        cap = np.zeros(9, dtype=np.int64)
        # (a node's room is rounded up to a multiple of four slots, so a leaf can outgrow SLACK by three)
        # (upper bounds over EVERY growth up to that: block counts and query lists are not monotone in the leaves' lengths)
        check(lib.deft_tree_md_caps(t._native, mq, bl, mbl, self.SLACK + 4, _ptr(cap)), "deft_tree_md_caps")
        self.cap_lens = _lens_from_sizes(cap)
        self.nbp_cap = int(cap[8])
        self.out = torch.empty(sum(self.cap_lens.values()) + 1, dtype=torch.int64, device=self.device)
        sb = int(lib.deft_tree_dev_scratch_bytes(n, nqw, self.nbp_cap))
        self.scratch = torch.zeros(sb, dtype=torch.uint8, device=self.device)  # (dims[] start at zero)
        self.scratch_bytes = sb
        # (read AFTER the fetch: a fetch that found a non-empty journal ends the epoch for every other copy -- they never saw those
        #  changes -- and this copy, whose image holds them, adopts the new number)
        self.epoch = t._epoch()

    def sync(self) -> bool:
        """Bring the device copy to the tree's structural epoch; True iff an upload happened (the uploaded image holds
        every slot the native tree holds NOW; a copy that was already current holds what it held before).  Changes the epoch
        absorbed -- slots appended to a node with room, a node's slots dropped: the native tree's journal -- are replayed on a
        copy that stays."""
        if self.epoch != self.tree._epoch():
            self._upload()
            return True
        return self.apply_journal() < 0

    def apply_journal(self) -> int:
        """Hand the native tree's journal of absorbed changes to the device copy (eager: one small upload + one kernel); returns
        the words replayed, or -1 when the journal was too long for one replay and the whole tree was UPLOADED instead (the image
        then holds every slot the native tree holds now, this step's included: the caller must not advance it again).  A captured
        session does the same inside its step graph (deft_tree_dev_build_md_ops)."""
        cap = 64 + 8 * max(self.nq, 1) + 4 * self.n
        buf = np.zeros(cap + 1, dtype=np.int32)
        nw = int(lib.deft_tree_journal_take(self.tree._native, _ptr(buf[1:]), cap))
        if nw == -5:  # too long for one replay: the call started a new epoch
            self._upload()
            return -1
        if nw < 0:
            check(nw, "deft_tree_journal_take")
        if nw == 0:
            return 0
        buf[0] = nw
        ops = torch.from_numpy(buf[: nw + 1]).to(self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(lib.deft_tree_dev_apply_ops(*self._tree_args(), ops.data_ptr(), self.scratch.data_ptr(), stream), "deft_tree_dev_apply_ops")
        self.version += 1
        return nw

    def _tree_args(self):
        return (self.n, self.nq, self.nqw, self.p_start, self.p_len, self.p_cap, self.p_refs, self.p_leaf, self.p_slots)

    def advance(self, cache_loc: torch.Tensor) -> None:
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(lib.deft_tree_dev_advance(*self._tree_args(), cache_loc.data_ptr(), self.scratch.data_ptr(), stream),
              "deft_tree_dev_advance")
        self.version += 1

    def build(self) -> Dict[str, object]:
        """Launch the metadata kernels; returns device tensors (views of the epoch's buffers) shaped by sizes the
        host computes from node lengths alone."""
        self.sync()
        t, (mq, bl, mbl) = self.tree, self.cfg
        sizes = np.zeros(9, dtype=np.int64)
        check(lib.deft_tree_md_sizes(t._native, mq, bl, mbl, 0, _ptr(sizes)), "deft_tree_md_sizes")
        lens = _lens_from_sizes(sizes)
        views, ptrs, off = {}, [], 0
        for k in _FIELDS:
            if lens[k] > self.cap_lens[k]:
                raise RuntimeError(f"device metadata buffer {k} too small ({lens[k]} > {self.cap_lens[k]})")
            views[k] = self.out[off : off + lens[k]]
            ptrs.append(self.out.data_ptr() + 8 * off)
            off += self.cap_lens[k]
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(lib.deft_tree_dev_build_md(*self._tree_args(), mq, bl, mbl, self.nbp_cap, self.scratch.data_ptr(),
                                         self.scratch_bytes, *ptrs, None, stream), "deft_tree_dev_build_md")
        leaf_ids = np.empty(max(int(sizes[0]), 1), dtype=np.int64)
        nl = int(lib.deft_tree_leaf_ids(t._native, _ptr(leaf_ids), len(leaf_ids)))
        views.update(query_num=int(sizes[0]), node_num=int(sizes[1]), total_kv_len=int(sizes[2]),
                     leaf_to_q={int(l): i for i, l in enumerate(leaf_ids[:nl])})
        return views

    def dims(self) -> List[int]:
        """dims[] as the device wrote them (debug / tests; synchronises)."""
        return self.scratch[:64].view(torch.int32).tolist()


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
        device_build: Optional[bool] = None,
        copy: bool = False,
    ) -> "TreeMetadata":
        """`device_build` (GPU pools): None / True = the arrays are built on the GPU from the device copy of the tree;
        False = by the host builder and uploaded in one copy (the round-1 path, kept as the checker).

        ALIASING (device build): the twelve tensors are VIEWS of the tree's per-epoch output buffer -- the next
        `from_tree_cache` of the same tree (the next decode step) overwrites them; the reference returns fresh tensors every
        call (tree_cache.py:813-857).  A decode loop consumes a step's metadata before it builds the next one; a caller that
        keeps metadata ACROSS steps passes `copy=True` (twelve small device-to-device copies on the current stream)."""
        assert tree.root is not None
        block_len = BLOCK_CONFIG["BLOCK_LEN"]
        if max_block_len == -1:
            max_block_len = BLOCK_CONFIG["MAX_BLOCK_LEN"]
        dev = torch.device(device) if device is not None else tree.token_to_kv_pool.device
        if dev.type == "cuda" and dev.index is None:  # ("cuda" and "cuda:0" must name the same device copy of the tree)
            dev = torch.device("cuda", torch.cuda.current_device())
        if dev.type != "cpu" and (DEVICE_METADATA if device_build is None else device_build):
            if not tree._consistent():
                raise RuntimeError("tree.nodes / tree.leaves were edited behind TreeCache's back")
            dt = tree._device_tree
            if dt is None or dt.device != dev or dt.cfg != (int(max_q_len), int(block_len), int(max_block_len)):
                dt = tree._device_tree = _DeviceTree(tree, dev, max_q_len, block_len, max_block_len)
            views = dt.build()
            if copy:
                views = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in views.items()}
            md = cls(block_len=block_len, **views)
            md.max_q_len = int(max_q_len)  # (the operators check it: at most 32 queries per block, deft_attention.py)
            return md
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
        md = cls(
            query_num=host["query_num"], node_num=host["node_num"], total_kv_len=host["total_kv_len"],
            leaf_to_q=host["leaf_to_q"], block_len=block_len, **views,
        )
        md.max_q_len = int(max_q_len)
        return md


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
