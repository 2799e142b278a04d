"""A decode loop over ONE tree as captured hipGraphs: one (legacy) or two (window plan) graphs per structural epoch of the tree.

What a runner does per decode step around the attention path (DeFT/deft/tree_decoding/generation/tree_generate.py:93-131,
model_runner.py:162-231): `tree.alloc()`, `TreeMetadata.from_tree_cache(tree)`, then the model's forward, whose every
layer appends this step's K/V rows and calls the tree-attention operator.  With the tree's compact copy on the GPU
(deft_amd/csrc/tree_plan.h) every launch of that sequence has the SAME arguments on every step until the tree's structure
changes -- the arrays and the plan live in buffers sized for the epoch, the block count of the step is read on the device --
so the whole step is captured once and replayed:

    page-table write, device tree advance (this step's slots)      tree_advance_kernel
    TreeMetadata on the GPU                                          tree_md_scan / _blocks / _nodes
    per-step plan                                                    flatten_units / _records, qrows_hist / _fill
    layers x (fused paged append + stage 1, merge)                   stage1_np_kernel, merge_kernel

Per step the host picks the new slots (its allocator mirrors the pool), appends them to the native tree, copies
nq slot numbers and page-table coordinates from pinned memory, and replays: ~0.1 ms of host time, nothing else crosses
PCIe.  A branch / cut / merge (or a leaf outgrowing its room) starts a new epoch: upload the compact tree, capture again.

`incremental=True` (the default; round 6, deft_amd/csrc/window.h): the metadata + plan head -- what the reference rebuilds whole on
every step, tree_cache.py:619-881 -- runs only on REPLAN steps, once per WINDOW of steps; all other steps run ONE small kernel
that patches the plan: this step's tokens (and the slots a speculative-decoding step merges into a node, and the slots it
drops) go to overflow tiles at the end of the plan, with per-slot row masks.  The books (which overflow position belongs to which
node, when a window is full) are kept here on the host; two graphs per epoch, the host picks which one a step replays.  Outputs
of window steps are another PARTITION of the same keys than the eager path's: equal within the oracle's tolerance, not bit for bit.
`incremental=False` is the round-5 loop: bit-identical to the eager path step for step (tests/test_session.py).

DeFT-Flatten and DeFT-Node; head_dim 128, and 64 as head pairs.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from ._lib import check, lib
from .tree_cache import BLOCK_CONFIG, TreeCache, _DeviceTree, _FIELDS, _ptr

__all__ = ["DecodeSession", "FlattenDecodeSession"]

import threading
import time

_TILE = 128
_WIN_PASSES = 16  # (plan_records.h WIN_PASSES: the overflow table's passes per query chunk)

# What a session needs from the DRIVER -- pinned host memory for its staging ring, a stream to capture on -- is kept per process and
# handed from session to session: hipHostMalloc and a stream's first use cost 1-8 ms each on a quiet box and 20-60 ms when the
# node's driver is busy (profiles/r6_slow_run_hunt.txt), a decode step 0.06 ms of host time.
_POOL_LOCK = threading.Lock()
_PINNED_RINGS: Dict[int, list] = {}  # bytes -> [(pinned uint8 tensor, event behind its last reader | None)]
_CAPTURE_STREAMS: Dict[tuple, "torch.cuda.Stream"] = {}  # (device index, thread) -> the stream that thread's sessions capture on


def _ring_acquire(nbytes: int) -> torch.Tensor:
    with _POOL_LOCK:
        have = _PINNED_RINGS.get(nbytes)
        item = have.pop() if have else None
    if item is None:
        return torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
    ring, ev = item
    if ev is not None:
        ev.synchronize()  # (the last copy / kernel that read it has run)
    return ring.zero_()


def _ring_release(ring: Optional[torch.Tensor], stream: Optional["torch.cuda.Stream"]) -> None:
    if ring is None:
        return
    ev = None
    if stream is not None:
        ev = torch.cuda.Event()
        ev.record(stream)
    with _POOL_LOCK:
        have = _PINNED_RINGS.setdefault(ring.numel(), [])
        if len(have) < 64:  # (beyond that the memory goes back to the driver)
            have.append((ring, ev))


def _capture_stream(device: torch.device) -> "torch.cuda.Stream":
    key = (device.index, threading.get_ident())
    side = _CAPTURE_STREAMS.get(key)
    if side is None:
        # made AND first used here: a HIP stream is created lazily at its first use, 5.6 ms that would otherwise land in a captured step
        side = _CAPTURE_STREAMS[key] = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
    return side


class DecodeSession:
    RING = 8        # pinned staging slots: how many steps the host may run ahead of the GPU
    EVENT_EVERY = 4  # ... kept from lapping it by an event behind every fourth step (an event per step sits between two captured
                     # graphs and costs the queue idle time: tools/step_boundary.py)

    def __init__(self, tree: TreeCache, num_heads: int, num_kv_heads: int, head_dim: int, layers: int,
                 qkv: Callable[[int], tuple], max_q_len: int = 32, use_graph: bool = True, mode: str = "flatten",
                 capture_after="auto", incremental: bool = True, win_tiles: Optional[int] = None, staging: Optional[str] = None) -> None:
        """`qkv(layer)` -> (q [nq, Hq*D], k_new [nq, Hkv*D], v_new [nq, Hkv*D]) fp16 CUDA tensors at FIXED addresses (the
        model writes this step's projections there); outputs are in `self.out[layer]` ([nq, Hq*D]).

        head_dim 128, or 64 with an even number of KV heads (the pool's contiguous heads: two heads of 64 to a 256-byte row, the
        tile-parallel kernel's head-pair form) -- the geometries whose stage 1 reads a per-step PLAN, which is what lets one captured
        launch serve every step of an epoch; head_dim 32 / 16 take the eager operators.

        `capture_after`: an epoch's step is captured once the epoch has lasted that many steps (its first step always runs
        eagerly: it follows an upload).  1 = at the second step, the round-3 behaviour: a capture costs ~0.65 ms of host time and
        a replayed step ~0.07 ms against ~0.55 ms for an eager one, so it pays from the third step of an epoch on.  "auto": 1,
        unless the LAST epoch ended within three steps -- trees that change shape every step or two (a controller that prunes after
        every token) then stay eager instead of paying a capture per step, and go back to capturing as soon as an epoch lasts.

        `incremental`: window plans (module docstring).  `win_tiles`: overflow tiles per query chunk (None: enough for ~16 steps of
        nq tokens, between 2 and 8)."""
        pool = tree.token_to_kv_pool
        assert pool.device.type == "cuda" and mode in ("flatten", "node")
        assert head_dim == 128 or (head_dim == 64 and num_kv_heads % 2 == 0), "DecodeSession: head_dim 128, or 64 with an even number of KV heads"
        assert capture_after == "auto" or int(capture_after) >= 1
        # (the operators fold at most 32 queries per block / entry -- the reference's BLOCK_M = 32, tree_attention.py:98, :586 -- so
        #  the metadata must not chunk the queries any coarser; tools/fuzz_session.py runs 7 / 16 / 32)
        assert 1 <= int(max_q_len) <= 32, "DecodeSession: max_q_len between 1 and 32 (the attention operators' query tile)"
        self.capture_after = capture_after
        self._epoch_steps = 0       # steps of the current epoch so far (its eager first one included)
        self._last_epoch_steps = 1 << 30
        self.mode = mode
        self.incremental, self.win_tiles_arg = bool(incremental), win_tiles
        # how a step's host-written words reach the GPU.  "copy" (default): ONE async copy from the pinned ring slot in front of the
        # step.  "kernel": fetched by the step's first kernel itself (csrc/window.h StageFetch; nothing but graph launches in the
        # stream) -- built to get rid of the idle queue around the copy, measured EQUAL once the per-step events were out of the stream
        # (few-shot replay 946.8 / 947.0 us per step, advancing loop 1.015-1.018 / 1.018-1.020 x a frozen step), and in about one run in
        # twenty the whole loop ran 2.6 x slower with it (kernel reads over PCIe; never with the copy: profiles/r6_staging_kernel_vs_copy.txt).
        import os as _os

        self.staging = staging or _os.environ.get("DEFT_SESSION_STAGING", "copy")
        assert self.staging in ("kernel", "copy")
        # (the device WITH its index: torch.device("cuda") != torch.device("cuda:0"), and the page-table fold below compares devices --
        #  round 3: with the default "cuda" pool the fold never happened and every step carried an index_put)
        self.tree, self.pool, self.device = tree, pool, pool._storage.device
        self.Hq, self.Hkv, self.D, self.layers = num_heads, num_kv_heads, head_dim, layers
        self.qkv, self.max_q_len, self.use_graph = qkv, max_q_len, use_graph
        self.graphs: Dict[str, Optional[torch.cuda.CUDAGraph]] = {}
        self.graph_epoch = -1
        self.captures = 0
        self.step_kinds = {"upload": 0, "legacy": 0, "replan": 0, "patch": 0}  # what the steps so far ran (tests, tools/replay.py)
        # the stream captures run on: one per device and thread, shared by the sessions (captures are serial on a thread)
        self._side: Optional[torch.cuda.Stream] = _capture_stream(self.device)
        self._last_stream: Optional[torch.cuda.Stream] = None  # the stream the last step's launches went to
        self.out: List[torch.Tensor] = []
        # the host's words of a step reach the GPU through a RING of pinned slots (copied from in front of the step, or read by the
        # step's first kernel: `staging`); an event behind every fourth step keeps the host from lapping the GPU
        self._ring: Optional[torch.Tensor] = None
        self._ring_slot = 0
        self._ring_events: List[Optional[torch.cuda.Event]] = [None] * (2 * self.RING // self.EVENT_EVERY)
        self._stage_no = 0   # steps staged so far (== the device counter once the GPU has caught up)
        self._ctr = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.win = 0  # native books of the epoch's window plans (deft_window_create), 0 = none
        self.debug, self.last_staged = False, None

    def __del__(self) -> None:
        try:
            if self.win:
                lib.deft_window_free(self.win)
            self.win = 0
        except Exception:
            pass
        try:  # (the pinned ring goes to the next session, behind an event on the stream that read it last)
            _ring_release(self._ring, self._last_stream)
            self._ring = None
        except Exception:
            pass

    @property
    def graph(self):  # (round-5 name: the epoch's captured step, whichever form)
        return self.graphs.get("legacy") or self.graphs.get("patch") or self.graphs.get("replan")

    # ---- per epoch ----------------------------------------------------------------------------------------
    def _epoch_setup(self) -> bool:
        """Buffers of a structural epoch; True iff the device copy of the tree was uploaded just now (then it already
        holds this step's slots), False iff it was current before this step's `alloc_step` -- a device-built
        `TreeMetadata.from_tree_cache(tree)` or another session got there first -- and still needs the step's slots."""
        tree, dev = self.tree, self.device
        block_len = BLOCK_CONFIG["BLOCK_LEN"]
        dt = tree._device_tree
        cfg = (int(self.max_q_len), int(block_len), int(BLOCK_CONFIG["MAX_BLOCK_LEN"]))
        if dt is None or dt.device != dev or dt.cfg != cfg:
            dt = tree._device_tree = _DeviceTree(tree, dev, *cfg)
        uploaded = dt.sync()
        self.dt = dt
        self.nq = dt.nq
        Hq, Hkv, D = self.Hq, self.Hkv, self.D
        nqm = max(self.nq, 1)
        # ---- window plan: overflow tiles per query chunk, and the room they need in the metadata arrays ------------------------
        chunks = (nqm + self.max_q_len - 1) // self.max_q_len
        W = 0
        G = Hq // Hkv
        # (a REGION of the overflow = one (query chunk, 32-row pass) pair; it takes the tokens of the queries whose rows it holds)
        regions = sum((min(self.max_q_len, nqm - c * self.max_q_len) * G + 31) // 32 for c in range(chunks))
        per_region = min(nqm, self.max_q_len, (32 + G - 1) // G)  # queries (= tokens per step) of a full region
        if self.incremental and self.nq > 0 and block_len == _TILE and int(lib.deft_window_supported(self.nq, self.max_q_len, Hq, Hkv)):
            # default: room for ~16 steps, at most 4 tiles.  A replan costs what EVERY step used to cost (~58 us), but an overflow
            # run is ONE chunk -- a serial chain of up to W tiles inside every layer's launch: measured (tools/replay.py --win-tiles,
            # profiles/r6_window_tiles.txt, us per step against the legacy loop) few-shot 4k x 32 on Llama-2-7B -52 / -55 at 4 / 8
            # tiles, the same tree on Llama-3-8B (a region holds 8 queries) -34 at 1 tile, +24 at 4, +94 at 8 (its chunks are 4 tiles)
            W = int(self.win_tiles_arg) if self.win_tiles_arg else min(4, max(1, (16 * per_region + _TILE - 1) // _TILE))
            if W * _TILE < 2 * per_region:  # (a window must hold at least two steps' tokens)
                W = 0
        self.W, self.chunks = W, chunks
        caps = dict(dt.cap_lens)
        if W:
            if self.mode == "flatten":
                for k in ("block_lens", "block_q_cnts", "block_q_offset"):
                    caps[k] += chunks * W
                caps["block_q"] += W * self.nq
                for k in ("block_kv", "block_bitmasks"):
                    caps[k] += chunks * W * _TILE
            else:
                for k in ("node_q_len", "node_kv_len", "node_q_offset", "node_kv_offset"):
                    caps[k] += chunks
                caps["node_q"] += self.nq
                caps["node_kv"] += chunks * W * _TILE
        # (the session's own arrays: dt.out belongs to TreeMetadata.from_tree_cache's device builder)
        self.md_out = torch.empty(sum(caps.values()) + 1, dtype=torch.int64, device=dev)
        self.md_caps = caps
        self.NB, self.P = caps["block_lens"], caps["block_q"]
        off, self.md_ptrs = 0, {}
        for k in _FIELDS:
            self.md_ptrs[k] = self.md_out.data_ptr() + 8 * off
            off += caps[k]
        if self.mode == "flatten":
            self.plan_bytes = int(lib.deft_flatten_plan_bytes(self.NB, self.P, Hq, Hkv))
            self.ws_bytes = int(lib.deft_flatten_workspace_bytes(self.NB, self.P, self.nq, Hq, Hkv, D))
        else:  # Node arrays: capacities of node_q_len (entries), node_q (pairs), node_kv (slots, repeated per query chunk)
            self.NE, self.PN, self.TKV = caps["node_q_len"], caps["node_q"], caps["node_kv"]
            self.plan_bytes = int(lib.deft_node_plan_bytes(self.NE, self.PN, self.TKV, Hq, Hkv))
            self.ws_bytes = int(lib.deft_node_workspace_bytes(self.NE, self.PN, self.TKV, self.nq, Hq, Hkv, D))
        self.plan = torch.empty(max(self.plan_bytes, 1), dtype=torch.uint8, device=dev)
        self.ws = torch.empty(max(self.ws_bytes, 1), dtype=torch.uint8, device=dev)
        # what the host supplies per step, in ONE allocation (fetched from the pinned ring by the step's first kernel):
        # [cache_loc int32[nq] | page-table coordinates int64[2][nq] | journal of absorbed changes int32 {words, ...} |
        #  window plan: patch list int32 {entries, active tiles of 64 regions, {region << 20 | position, row mask, slot | new row} ...}]
        cb = (4 * nqm + 255) // 256 * 256
        self.ops_cap = 64 + 8 * nqm  # one EXTEND of up to nq slots and a RESET per leaf, with room to spare (a speculative-decoding step)
        ob = cb + 16 * nqm
        pb = (ob + 4 * (self.ops_cap + 1) + 255) // 256 * 256
        self.patch_cap = (4 * nqm + 64 + 2 * nqm * regions) if W else 0  # (a slot merged into the root is an entry in every region)
        self._small = torch.zeros((pb + 4 * (1 + 64 + 3 * self.patch_cap) + 15) // 16 * 16, dtype=torch.uint8, device=dev)  # (fetched in 16-byte chunks)
        self.cache_loc = self._small[: 4 * nqm].view(torch.int32)
        self.idx = self._small[cb:ob].view(torch.int64).view(2, nqm)
        self.ops = self._small[ob:pb].view(torch.int32)
        self.patch = self._small[pb:].view(torch.int32)
        self._ops_off, self._patch_off = ob, pb
        need = (16 + self._small.numel() + 15) // 16 * 16
        if self._ring is None or need > self._ring_slot:  # (grown only: between epochs, once every slot has been read)
            if self._ring is not None:
                torch.cuda.current_stream(dev).synchronize()
                _ring_release(self._ring, self._last_stream)
            self._ring_slot = max(4096, 1 << (need - 1).bit_length())  # (powers of two: rings are handed from session to session)
            self._ring = _ring_acquire(self.RING * self._ring_slot)
        self.win_tab = torch.zeros(max(chunks, 1) * _WIN_PASSES * 2, dtype=torch.int32, device=dev)
        if self.win:
            lib.deft_window_free(self.win)
        self.win = 0
        if W:
            self.win = int(lib.deft_window_create(dt.n, self.nq, dt.nqw, W, _ptr(dt.h_leaf), _ptr(dt.h_refs), self.max_q_len, Hq // Hkv,
                                                  self.patch_cap))
            if self.win < 0:
                check(self.win, "deft_window_create")
        self.out = [torch.empty((self.nq, Hq * D), dtype=torch.float16, device=dev) for _ in range(self.layers)]
        order = sorted(tree.leaves)
        self.leaf_handles = [tree.leaves[i] for i in order]
        self.leaf_reqs = np.asarray([tree.leaf_to_req[i] for i in order], dtype=np.int64)
        self._journal = np.zeros(self.ops_cap, dtype=np.int32)
        self._journal_p = _ptr(self._journal)
        self._slot_views = [None] * self.RING  # (per ring slot: numpy views of this epoch's layout, `_stage`)
        self._ring_p, self._small_p, self._small_n = self._ring.data_ptr(), self._small.data_ptr(), self._small.numel()
        self.graphs, self.graph_epoch = {}, dt.epoch
        return uploaded

    def _fetch(self, stream: int) -> None:
        if self.staging == "copy":  # (the words were copied in front of the step: nothing to fetch)
            return
        check(lib.deft_stage_fetch(self._ring.data_ptr(), self._ring_slot, self.RING, self._small.data_ptr(), self._ctr.data_ptr(), stream),
              "deft_stage_fetch")

    def _can_fold(self, advance: bool) -> bool:
        table = self.tree.req_to_token_pool.req_to_token
        return advance and table.dtype == torch.int32 and table.device == self.device and table.dim() == 2 and table.stride(1) == 1

    def _fold(self, advance: bool) -> bool:
        """Are the page-table entries of this step's tokens written by the step's first kernel?  (When the table is what the kernel
        expects -- int32, contiguous rows, on this device; by an index_put otherwise.)"""
        table = self.tree.req_to_token_pool.req_to_token
        fold = self._can_fold(advance)
        self.page_table_folded = fold  # (tests: the fold must really happen for the pools this package makes)
        if not fold:
            table[self.idx[0], self.idx[1]] = self.cache_loc
        return fold

    def _launch_layers(self) -> None:
        Hq, Hkv, D = self.Hq, self.Hkv, self.D
        kv0 = self.pool.kv_data[0]
        v_off = kv0.stride(1) * 2
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if self.mode == "flatten":
            mdl = [self.md_ptrs[k] for k in ("block_q", "block_q_cnts", "block_q_offset", "block_bitmasks", "block_kv", "block_lens")]
            fn, tail = lib.deft_flatten_decode_append_f16, (self.NB, self.P, self.nq, Hq, Hkv, D, 1.0 / (D ** 0.5))
        else:
            mdl = [self.md_ptrs[k] for k in ("node_kv", "node_kv_offset", "node_kv_len", "node_q", "node_q_offset", "node_q_len")]
            fn, tail = lib.deft_node_decode_append_f16, (self.NE, self.PN, self.TKV, self.nq, Hq, Hkv, D, 1.0 / (D ** 0.5))
        for l in range(self.layers):
            q, k, v = self.qkv(l)
            kptr = self.pool.kv_data[l].data_ptr()
            check(fn(q.data_ptr(), q.stride(0), D, kptr, kptr + v_off, kv0.stride(0), kv0.stride(2), self.out[l].data_ptr(), Hq * D, D,
                     *mdl, *tail, self.cache_loc.data_ptr(), k.data_ptr(), v.data_ptr(), k.stride(0), self.nq, self.plan.data_ptr(),
                     self.ws.data_ptr(), self.ws_bytes, stream), "decode layer")

    def _launch_step(self, advance: bool = True) -> None:
        """The device side of one LEGACY decode step (everything rebuilt); identical arguments on every step of the epoch.
        `advance=False`: the first step of an epoch when the tree was uploaded just now -- the image already holds this step's slots."""
        dt, dev = self.dt, self.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        table = self.tree.req_to_token_pool.req_to_token
        self._fetch(stream)
        fold = self._fold(advance)
        mq, bl, mbl = dt.cfg
        # (the append of this step's slots to the device tree rides in the first metadata kernel)
        # (only the six arrays this mode's operator reads are written: the kernel of the other group is not launched)
        wanted = _FIELDS[6:] if self.mode == "flatten" else _FIELDS[:6]
        # (first the journal of changes the epoch absorbed since the last step -- {0} when there are none --, then the advance)
        check(lib.deft_tree_dev_build_md_ops(*dt._tree_args(), mq, bl, mbl, dt.nbp_cap, dt.scratch.data_ptr(), dt.scratch_bytes,
                                             *[self.md_ptrs[k] if k in wanted else None for k in _FIELDS],
                                             self.cache_loc.data_ptr() if advance else None, self.ops.data_ptr(),
                                             table.data_ptr() if fold else None, table.stride(0) if fold else 0,
                                             self.idx[0].data_ptr() if fold else None, self.idx[1].data_ptr() if fold else None, stream),
              "deft_tree_dev_build_md_ops")
        Hq, Hkv, D = self.Hq, self.Hkv, self.D
        q0, k0, _ = self.qkv(0)
        kv0 = self.pool.kv_data[0]
        if self.mode == "flatten":
            mdl = [self.md_ptrs[k] for k in ("block_q", "block_q_cnts", "block_q_offset", "block_bitmasks", "block_kv", "block_lens")]
            check(lib.deft_flatten_build_plan_dims(*mdl, self.NB, self.P, dt.scratch.data_ptr(), Hq, Hkv, q0.stride(0), D,
                                                   kv0.stride(0), self.cache_loc.data_ptr(), self.nq, k0.stride(0),
                                                   self.plan.data_ptr(), self.plan_bytes, stream), "deft_flatten_build_plan_dims")
        else:
            mdl = [self.md_ptrs[k] for k in ("node_kv", "node_kv_offset", "node_kv_len", "node_q", "node_q_offset", "node_q_len")]
            check(lib.deft_node_build_plan_dims(*mdl, self.NE, self.PN, self.TKV, dt.scratch.data_ptr(), Hq, Hkv, q0.stride(0), D,
                                                kv0.stride(0), self.cache_loc.data_ptr(), self.nq, k0.stride(0),
                                                self.plan.data_ptr(), self.plan_bytes, stream), "deft_node_build_plan_dims")
        self._launch_layers()

    def _launch_window_step(self, replan: bool) -> None:
        """The device side of a window-plan step (csrc/window.h).  REPLAN: metadata and plan of the tree as it stands before this
        step's tokens (the scan kernel replays the journal, appends nothing), overflow tiles appended; then -- PATCH steps: only --
        one kernel that advances the device tree and applies the host's patch list; then the layers."""
        dt, dev = self.dt, self.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        table = self.tree.req_to_token_pool.req_to_token
        # a PATCH step whose page-table write rides in the patch kernel has that kernel fetch the step's words too (one launch
        # in front of the layers); everything else fetches with a launch of its own, in front of whatever reads them
        folded_fetch = not replan and self._can_fold(True) and self.staging == "kernel"
        if not folded_fetch:
            self._fetch(stream)
        fold = self._fold(True)
        mq, bl, mbl = dt.cfg
        Hq, Hkv, D = self.Hq, self.Hkv, self.D
        q0, k0, _ = self.qkv(0)
        kv0 = self.pool.kv_data[0]
        if replan:
            wanted = _FIELDS[6:] if self.mode == "flatten" else _FIELDS[:6]
            check(lib.deft_tree_dev_build_md_ops(*dt._tree_args(), mq, bl, mbl, dt.nbp_cap, dt.scratch.data_ptr(), dt.scratch_bytes,
                                                 *[self.md_ptrs[k] if k in wanted else None for k in _FIELDS],
                                                 None, self.ops.data_ptr(), None, 0, None, None, stream), "deft_tree_dev_build_md_ops")
            if self.mode == "flatten":
                mdl = [self.md_ptrs[k] for k in ("block_q", "block_q_cnts", "block_q_offset", "block_bitmasks", "block_kv", "block_lens")]
                check(lib.deft_flatten_build_plan_window(*mdl, self.NB, self.P, dt.scratch.data_ptr(), self.nq, mq, self.W,
                                                         self.win_tab.data_ptr(), Hq, Hkv, q0.stride(0), D, kv0.stride(0),
                                                         self.plan.data_ptr(), self.plan_bytes, stream), "deft_flatten_build_plan_window")
            else:
                mdl = [self.md_ptrs[k] for k in ("node_kv", "node_kv_offset", "node_kv_len", "node_q", "node_q_offset", "node_q_len")]
                check(lib.deft_node_build_plan_window(*mdl, self.NE, self.PN, self.TKV, dt.scratch.data_ptr(), self.nq, mq, self.W,
                                                      self.win_tab.data_ptr(), Hq, Hkv, q0.stride(0), D, kv0.stride(0),
                                                      self.plan.data_ptr(), self.plan_bytes, stream), "deft_node_build_plan_window")
        check(lib.deft_window_patch(*dt._tree_args(), None if replan else self.ops.data_ptr(), self.cache_loc.data_ptr(),
                                    table.data_ptr() if fold else None, table.stride(0) if fold else 0,
                                    self.idx[0].data_ptr() if fold else None, self.idx[1].data_ptr() if fold else None,
                                    self.patch.data_ptr(), self.win_tab.data_ptr(), self.plan.data_ptr(), mq, self.W, Hq, Hkv,
                                    kv0.stride(0), k0.stride(0), dt.scratch.data_ptr(),
                                    *((self._ring.data_ptr(), self._ring_slot, self.RING, self._small.data_ptr(), self._ctr.data_ptr())
                                      if folded_fetch else (None, 0, 0, None, None)), stream), "deft_window_patch")
        self._launch_layers()

    # ---- per step ------------------------------------------------------------------------------------------
    def step(self) -> List[torch.Tensor]:
        """One decode step: every live leaf has taken its token (`leaf.append_token`); returns the per-layer outputs."""
        tree = self.tree
        n = len(tree.leaves)
        loc = self.pool.alloc_host(n)
        assert loc is not None
        loc64 = loc.astype(np.int64)
        check(lib.deft_tree_alloc_step(tree._native, n, _ptr(loc64)), "deft_tree_alloc_step")
        # changes the epoch absorbed since the last step (merge_nodes into a node with room, reset_node_KV): the journal rides in
        # this step's upload and is replayed by the step's first kernel.  One that does not fit starts a new epoch instead.
        jn = 0
        if tree._epoch() == self.graph_epoch:
            jn = int(lib.deft_tree_journal_take(tree._native, self._journal_p, self.ops_cap))
            if jn < 0 and jn != -5:
                check(jn, "deft_tree_journal_take")
            jn = max(jn, 0)
        if tree._epoch() != self.graph_epoch:
            # a new structural epoch (branch / cut / merge since the last step, or a leaf outgrew its room): an upload made
            # now already contains this step's slots, so this step runs eagerly without the advance; a device copy that was
            # current BEFORE this step's alloc_step (and stayed in its epoch) appends them itself.  The next step captures.
            self._last_epoch_steps, self._epoch_steps = (self._epoch_steps if self.graph_epoch >= 0 else 1 << 30), 1
            uploaded = self._epoch_setup()
            jn = 0
            if not uploaded:  # (a device copy that stays may still owe the journal)
                jn = max(int(lib.deft_tree_journal_take(tree._native, self._journal_p, self.ops_cap)), 0)
                if tree._epoch() != self.graph_epoch:  # too long: that call started another epoch
                    uploaded, jn = self._epoch_setup(), 0
            self._stage(loc, jn, False)
            self._launch_step(advance=not uploaded)  # (a legacy step: the next one starts the epoch's first window)
            self._staged()
            self._moved()
            self.step_kinds["upload"] += 1
            return self.out
        # ---- which form this step takes: the books decide while the staging buffer is written --------------------------------
        kind = self._stage(loc, jn, bool(self.win))
        self._epoch_steps += 1
        self.step_kinds[kind] += 1
        launch = self._launch_step if kind == "legacy" else (lambda: self._launch_window_step(kind == "replan"))
        if self.graphs.get(kind) is None:
            wait = (1 if self._last_epoch_steps > 3 else 4) if self.capture_after == "auto" else int(self.capture_after)
            if not self.use_graph or self._epoch_steps <= wait:  # (this is step `_epoch_steps` of the epoch; `wait` of them run eagerly)
                launch()
                self._staged()
                self._moved()
                return self.out
            self._capture(kind, launch)
        self.graphs[kind].replay()
        self._staged()
        self._moved()
        return self.out

    def _staged(self) -> None:
        """The launches that read the ring slot written by the last `_stage` are in the stream; every EVENT_EVERY-th step leaves an event."""
        s = self._stage_no - 1  # the step just launched
        if s % self.EVENT_EVERY != self.EVENT_EVERY - 1:
            return
        k = (s // self.EVENT_EVERY) % len(self._ring_events)
        if self._ring_events[k] is None:
            self._ring_events[k] = torch.cuda.Event()  # (re-recorded: not one hipEventCreate per step)
        self._ring_events[k].record(torch.cuda.current_stream(self.device))

    def device_errors(self) -> int:
        """Error flags the step's kernels raised on the device since the epoch's upload (SYNCHRONISES; tests and fuzzers call it at
        the end of a run): bit 0 a leaf outgrew its room, bit 1 more blocks than the buffers hold, bit 2 a malformed journal, bit 3 a
        window plan's overflow tiles or patch entries out of range.  0 = none."""
        torch.cuda.current_stream(self.device).synchronize()
        return int(self.dt.scratch[:64].view(torch.int32)[9].item())

    def _moved(self) -> None:
        """This session's launches have advanced the device copy of the tree."""
        self.dt.version += 1
        self._dt_version = self.dt.version

    def _stage(self, loc: np.ndarray, journal_words: int, window: bool) -> str:
        """This step's slot numbers, page-table coordinates, journal and -- window plans -- patch list, written into a slot of the
        pinned ring; the step's first kernel copies it into the fixed device tensors the graph reads (`_fetch`, csrc/window.h).
        Returns the form the step takes: "patch" (the window goes on), "replan" (a new window starts) or "legacy"."""
        n = self.nq
        k = self._stage_no % self.RING
        need = self._stage_no - self.RING  # the step that read this slot last must have run: wait for the first event at or behind it
        self._stage_no += 1
        if need >= 0:
            e = need + (self.EVENT_EVERY - 1 - need) % self.EVENT_EVERY
            ev = self._ring_events[(e // self.EVENT_EVERY) % len(self._ring_events)]
            if ev is not None and not ev.query():
                # Polled, not hipEventSynchronize: on some boxes that call returns 10 or 20 ms late (the wake-up is missed and a timer
                # finds the finished event, profiles/r6_slow_run_hunt.txt) while the query reads the signal itself.  Spinning for
                # 0.2 ms, then between short sleeps: the host is 4-8 steps ahead here, a late look costs the GPU nothing.
                t_spin = time.perf_counter() + 2e-4
                while not ev.query():
                    if time.perf_counter() > t_spin:
                        time.sleep(5e-5)
        sv = self._slot_views[k]
        if sv is None:  # (numpy views of ring slot k and their addresses: made once per epoch, not per step)
            slot = self._ring.numpy()[k * self._ring_slot : (k + 1) * self._ring_slot]
            h = slot[16:]
            nqm = max(n, 1)
            loc32 = h[: 4 * n].view(np.int32)
            ph = h[self._patch_off : self._small.numel()].view(np.int32)  # (what the device-side patch area holds)
            sv = self._slot_views[k] = (slot[:4].view(np.uint32), h, loc32, h[self._ops_off - 16 * nqm : self._ops_off].view(np.int64).reshape(2, nqm),
                                        h[self._ops_off : self._patch_off].view(np.int32), ph, _ptr(loc32), _ptr(ph), ph.size)
            sv[3][0, :n] = self.leaf_reqs  # (the leaves' request rows do not change within an epoch)
        hdr, h, loc32, idx_h, ops_h, ph, loc_p, ph_p, ph_n = sv
        loc32[:] = loc
        idx_h[1, :n] = [lf.positions[-1] for lf in self.leaf_handles]
        ops_h[0] = journal_words
        if journal_words:
            ops_h[1 : 1 + journal_words] = self._journal[:journal_words]
        kind, used = "legacy", self._patch_off
        if window:
            words = -1
            # (nobody else has moved the device copy since this session's last step?  Otherwise the plan's static part is stale: replan)
            if self.dt.version == self._dt_version:
                words = int(lib.deft_window_step(self.win, 0, self._journal_p, journal_words, loc_p, ph_p, ph_n))
                kind = "patch"
            if words < 0:
                words = int(lib.deft_window_step(self.win, 1, self._journal_p, journal_words, loc_p, ph_p, ph_n))
                kind = "replan"
            if words < -1:
                check(words, "deft_window_step")
            if words < 0:
                kind = "legacy"
            else:
                used += 4 * words
        if self.debug:  # (tools/fuzz_session.py: what this step staged, for a failure report)
            self.last_staged = {"kind": kind, "journal": self._journal[:journal_words].copy(), "loc": loc32.copy(),
                                "patch": h[self._patch_off : used].view(np.int32).copy() if kind != "legacy" else None}
        hdr[0] = used  # (only what this step wrote crosses PCIe: the journal and patch areas are sized for the worst step)
        cur = self._last_stream = torch.cuda.current_stream(self.device)  # (where this step's copy and launches go)
        if self.staging == "copy":
            check(lib.deft_stage_copy(self._ring_p, self._ring_slot, k, self._small_p, self._small_n, cur.cuda_stream), "deft_stage_copy")
        return kind

    def _capture(self, kind: str, launch) -> None:
        dev = self.device
        graph = torch.cuda.CUDAGraph()
        # ONE capture stream per session: a new stream's first use costs 5.6 ms (measured round 3: the HIP stream is created
        # lazily, at the wait below) -- per structural epoch, more than everything else a capture does (0.8 ms)
        if self._side is None:
            self._side = _capture_stream(dev)
        side = self._side
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            # capture_begin / capture_end directly: `with torch.cuda.graph(...)` also synchronises the device, runs the Python garbage
            # collector and empties the caching allocator on entry -- milliseconds per structural epoch, and every buffer of the
            # next epoch then comes from hipMalloc again
            graph.capture_begin()
            try:
                launch()
            finally:
                graph.capture_end()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graphs[kind] = graph
        self.captures += 1


class FlattenDecodeSession(DecodeSession):
    """DeFT-Flatten (the north-star mode)."""

    def __init__(self, *args, **kw) -> None:
        kw["mode"] = "flatten"
        super().__init__(*args, **kw)
