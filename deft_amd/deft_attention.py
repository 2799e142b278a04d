"""DeFTAttention: the nn.Module the Llama layers call (llama2.py:111), for the two
DeFT decode modes.

Mirrors DeFT/deft/layers/attention/deft_attention.py:
  deft_node_forward     :72-108
  deft_flatten_forward  :110-151
  prefill_forward_triton  :50-70    (causal attention over the prompt, then store_kv_cache)
  radix_attention_forward :153-188  (sequential per-leaf attention: the comparator, `--mode seq`)
  forward               :349-388   (dispatch on input_metadata.forward_mode)
  store_kv_cache        :390-403

The reference brackets both phases with `torch.cuda.synchronize()` timers
(:117,126,135,149) and does a device-to-host `.item()` per layer for IO accounting
(:88-91); none of that is on the device path here — a decode step is
launch-only, so 32 layers can be replayed from one hipGraph.
"""
from __future__ import annotations

import torch
from torch import nn

from ._lib import DeftLibraryError, check, lib, tensor_version
from .forward_mode import ForwardMode, InputMetadata
from .context_attention import context_attention_fwd
from .token_attention import seq_append_attention, token_attention_fwd
from .tree_attention import (flatten_append_attention, node_append_attention, tree_attention_fwd,
                             tree_attention_subtree_fwd)
from .tree_cache import get_global_tree_metadata


ROPE_FUSE_MAX_KV_BYTES = 24 << 20  # DeFTAttention.forward(fuse_rope=None): fuse the rotary embedding below this much KV per layer
FUSED_APPEND = True  # False: store_kv_cache (its own launch), then the operator -- the reference's two-call form, for A/B


def _fused_append_enabled() -> bool:
    return FUSED_APPEND


class _DecodeStep:
    """Everything about one decode step that is the same for all layers, resolved once: metadata pointers, the
    per-step plan, the pool geometry, the step's cache_loc.  The 32 attention modules of a step share one instance
    (parked on the TreeMetadata object the runner registers, tree_cache.py:1021-1037), so a layer's call costs a
    handful of pointer computations and one ctypes call -- not the argument checks, views and cache lookups of the
    general operator (deft_amd.tree_attention.*_append_attention: ~36 us of host time per call, this path ~12 us;
    tools/host_overhead.py).  Same launches, same results."""

    def __init__(self, mode: ForwardMode, md, pool, cache_loc: torch.Tensor, q: torch.Tensor, k: torch.Tensor,
                 Hq: int, Hkv: int, D: int) -> None:
        from .tree_attention import _flatten_plan, _node_plan

        kv0 = pool.kv_data[0]
        if not (q.is_cuda and kv0.is_cuda and q.dtype == torch.float16 and kv0.dtype == torch.float16):
            raise TypeError("decode step needs fp16 CUDA tensors")
        self.mode, self.md, self.pool, self.cache_loc = mode, md, pool, cache_loc
        # (an inference tensor has no version counter, -1: the step is then identified by the objects alone -- a
        #  TreeMetadata and a cache_loc tensor are made anew every decode step)
        self.cache_loc_version = tensor_version(cache_loc)
        self.Hq, self.Hkv, self.D, self.nq = Hq, Hkv, D, q.shape[0]
        self.q_shape, self.q_stride, self.k_stride = tuple(q.shape), q.stride(0), k.stride(0)
        self.device = q.device
        self.dev_index = q.device.index if q.device.index is not None else torch.cuda.current_device()
        if self.dev_index != torch.cuda.current_device():  # (the library's per-device state follows HIP's current device)
            raise DeftLibraryError(f"tensors on cuda:{self.dev_index} but the current device is cuda:{torch.cuda.current_device()}: "
                                   "call inside `with torch.cuda.device(tensor.device):`")
        self.kv_ss, self.kv_sh = kv0.stride(0), kv0.stride(2)
        self.v_off_bytes = kv0.stride(1) * 2
        self.layer_ptrs = [t.data_ptr() for t in pool.kv_data]
        self.scale = 1.0 / (D ** 0.5)
        stream = torch._C._cuda_getCurrentRawStream(self.dev_index)
        if cache_loc.dtype != torch.int32 or not cache_loc.is_cuda:
            raise TypeError("cache_loc must be an int32 CUDA tensor")
        self.n_new = cache_loc.shape[0]
        if mode == ForwardMode.TREE_DECODE_FLATTEN:
            mdl = [md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv, md.block_lens]
            self.NB, self.P = md.block_q_cnts.shape[0], md.block_q.shape[0]
            self.plan = _flatten_plan(mdl, self.NB, self.P, Hq, Hkv, (self.q_stride, D), self.kv_ss, stream,
                                      cache_loc=cache_loc, new_stride=self.k_stride)
            self.ws_bytes = lib.deft_flatten_workspace_bytes(self.NB, self.P, self.nq, Hq, Hkv, D)
            self.tail = (self.NB, self.P, self.nq, Hq, Hkv, D, self.scale)
            self.fn = lib.deft_flatten_decode_append_f16
            self.fn_rope = lib.deft_flatten_decode_rope_append_f16
        else:
            mdl = [md.node_kv, md.node_kv_offset, md.node_kv_len, md.node_q, md.node_q_offset, md.node_q_len]
            NE, P, total = md.node_kv_offset.shape[0], md.node_q.shape[0], md.node_kv.shape[0]
            self.plan = _node_plan(mdl, NE, P, total, Hq, Hkv, (self.q_stride, D), self.kv_ss, stream,
                                   cache_loc=cache_loc, new_stride=self.k_stride)
            self.ws_bytes = lib.deft_node_workspace_bytes(NE, P, total, self.nq, Hq, Hkv, D)
            self.tail = (NE, P, total, self.nq, Hq, Hkv, D, self.scale)
            self.fn = lib.deft_node_decode_append_f16
            self.fn_rope = lib.deft_node_decode_rope_append_f16
        for t in mdl:
            if t.dtype != torch.int64 or not t.is_cuda or not t.is_contiguous():
                raise TypeError("TreeMetadata arrays must be contiguous int64 CUDA tensors")
        self.md_ptrs = tuple(t.data_ptr() for t in mdl)
        self.md_keep = mdl
        self.cache_loc_ptr = cache_loc.data_ptr()
        self.plan_ptr = self.plan.data_ptr()
        self.ws = {}  # raw stream -> workspace (calls on one stream are ordered; streams must not share scratch)
        self._rope_key, self._rope_rows = None, None
        nl = len(self.layer_ptrs)
        self._o_all = torch.empty((nl, self.nq, Hq * D), dtype=torch.float16, device=self.device) if nl * self.nq * Hq * D <= (64 << 20) else None
        self._o_used = [False] * nl  # (a layer called twice within a step gets a fresh tensor the second time)

    def matches(self, mode, md, pool, cache_loc, q, k) -> bool:
        return (mode is self.mode and md is self.md and pool is self.pool and cache_loc is self.cache_loc
                and tensor_version(cache_loc) == self.cache_loc_version and tuple(q.shape) == self.q_shape
                and q.stride(0) == self.q_stride and k.stride(0) == self.k_stride and q.dtype == torch.float16)

    def run(self, layer_id: int, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, rope=None) -> torch.Tensor:
        """`rope` = (positions int64 [nq], cos_sin_cache fp32 [max_pos][rotary_dim], rotary_dim, is_neox_style): q and k are
        the UNROTATED rows; the rotation happens inside the stage-1 launch (deft_*_decode_rope_append_f16)."""
        Hq, D = self.Hq, self.D
        if q.stride(1) != 1 or k.stride(1) != 1 or v.stride(1) != 1 or v.stride(0) != self.k_stride or k.dtype != torch.float16:
            raise ValueError("q / k / v rows must be contiguous fp16, k and v with the same row stride")
        # outputs of the step's layers are rows of ONE allocation made with the step (a step object lives for one decode
        # step, so nothing is overwritten): saves an allocator round trip per layer call
        if self._o_all is None or layer_id >= self._o_all.shape[0] or self._o_used[layer_id]:
            o = torch.empty((self.nq, Hq * D), dtype=torch.float16, device=self.device)
        else:
            o = self._o_all[layer_id]
            self._o_used[layer_id] = True
        stream = torch._C._cuda_getCurrentRawStream(self.dev_index)
        ws = self.ws.get(stream)
        if ws is None:
            ws = self.ws[stream] = torch.empty(max(self.ws_bytes, 1), dtype=torch.uint8, device=self.device)
        kptr = self.layer_ptrs[layer_id]
        head = (q.data_ptr(), self.q_stride, D, kptr, kptr + self.v_off_bytes, self.kv_ss, self.kv_sh,
                o.data_ptr(), Hq * D, D, *self.md_ptrs, *self.tail,
                self.cache_loc_ptr, k.data_ptr(), v.data_ptr(), self.k_stride, self.n_new)
        if rope is None:
            rc = self.fn(*head, self.plan_ptr, ws.data_ptr(), self.ws_bytes, stream)
        else:
            positions, cache, rotary_dim, neox = rope
            # cos|sin rows of this step's positions, gathered ONCE per step (the 32 layers share them): the kernel reads
            # row j for query row j -- no positions -> cache indirection in front of a workgroup's first MFMA
            if (self._rope_key is None or self._rope_key[0] is not positions
                    or self._rope_key[1] != tensor_version(positions) or self._rope_key[2] is not cache):
                rows = torch.empty((self.nq, rotary_dim), dtype=torch.float32, device=self.device)
                check(lib.deft_rope_gather_rows(positions.data_ptr(), cache.data_ptr(), cache.stride(0), self.nq, rotary_dim,
                                                rows.data_ptr(), stream), "deft_rope_gather_rows")
                self._rope_rows = rows
                self._rope_key = (positions, tensor_version(positions), cache)
            rc = self.fn_rope(*head, self._rope_rows.data_ptr(), rotary_dim, 1 if neox else 0,
                              self.plan_ptr, ws.data_ptr(), self.ws_bytes, stream)
        check(rc, "decode step")
        return o


def rope_fusable(rotary_emb, head_dim: int) -> bool:
    """The configurations deft_*_decode_rope_append_f16 take (the reference's Llama: NeoX pairing over the whole head)."""
    cache = getattr(rotary_emb, "cos_sin_cache", None)
    return (head_dim == 128 and getattr(rotary_emb, "rotary_dim", None) == head_dim
            and getattr(rotary_emb, "head_size", None) == head_dim and bool(getattr(rotary_emb, "is_neox_style", False))
            # the kernels read the cache as rows of fp32 cos | sin (rotary_embedding.py; the reference's get_rope builds it so)
            and isinstance(cache, torch.Tensor) and cache.dtype == torch.float32 and cache.dim() == 2 and cache.stride(1) == 1)


def _decode_step(mode, md, input_metadata, q, k, Hq, Hkv, D):
    """The step object for this (metadata, pool, cache_loc), or None when the fused path does not apply."""
    updater = input_metadata.kv_updater
    pool = input_metadata.token_to_kv_pool
    if (updater is None or updater.cache_loc is None or updater.token_to_kv_pool is not pool or not _fused_append_enabled()
            or q.dim() != 2 or k.dim() != 2 or not q.is_cuda):
        return None
    step = md.__dict__.get("_deft_step")
    if step is None or not step.matches(mode, md, pool, updater.cache_loc, q, k):
        cl = updater.cache_loc
        if cl.dtype != torch.int32 or not cl.is_cuda:
            return None
        step = md.__dict__["_deft_step"] = _DecodeStep(mode, md, pool, cl, q, k, Hq, Hkv, D)
    return step


def _check_query_tile(md) -> None:
    """The operators fold at most 32 queries per block / entry -- the reference's BLOCK_M = 32 (tree_attention.py:98, :586).  Metadata
    built with a coarser `max_q_len` is what the reference's builder would emit, but no kernel of either implementation reads it
    right: refused loudly (found by tools/fuzz_session.py, round 6)."""
    if getattr(md, "max_q_len", 32) > 32:
        from ._lib import DeftLibraryError

        raise DeftLibraryError(f"TreeMetadata was built with max_q_len={md.max_q_len}: the attention operators take at most 32 queries "
                               "per block (the reference's BLOCK_M)")


class DeFTAttention(nn.Module):
    def __init__(self, num_heads: int, head_dim: int, scaling: float, num_kv_heads: int, layer_id: int) -> None:
        super().__init__()
        self.tp_q_head_num = num_heads
        self.tp_k_head_num = num_kv_heads
        self.tp_v_head_num = num_kv_heads
        self.scaling = scaling
        self.head_dim = head_dim
        self.layer_id = layer_id
        if abs(scaling - head_dim ** -0.5) > 1e-6 * scaling:
            # the reference kernels ignore `scaling` and always use 1/sqrt(head_dim) (tree_attention.py:102, :601)
            raise ValueError("scaling must be head_dim ** -0.5, as the reference kernels assume")

    def deft_node_forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                          input_metadata: InputMetadata) -> torch.Tensor:
        md = get_global_tree_metadata()
        assert md is not None
        _check_query_tile(md)
        assert input_metadata.token_to_kv_pool is not None
        step = _decode_step(ForwardMode.TREE_DECODE_NODE, md, input_metadata, q, k, self.tp_q_head_num,
                            self.tp_k_head_num, self.head_dim)
        if step is not None:
            return step.run(self.layer_id, q, k, v)
        k = k.view(-1, self.tp_k_head_num, self.head_dim)
        v = v.view(-1, self.tp_v_head_num, self.head_dim)
        o = torch.empty((q.shape[0], self.tp_q_head_num * self.head_dim), dtype=q.dtype, device=q.device)
        pool = input_metadata.token_to_kv_pool
        updater = input_metadata.kv_updater
        if (updater is not None and updater.cache_loc is not None and updater.token_to_kv_pool is pool
                and _fused_append_enabled()):
            # store_kv_cache (:83) and the operator (:94-105) in one fused call
            node_append_attention(
                q.view(-1, self.tp_q_head_num, self.head_dim), pool.kv_data[self.layer_id],
                o.view(-1, self.tp_q_head_num, self.head_dim), updater.cache_loc, k, v,
                md.node_kv, md.node_kv_offset, md.node_kv_len, md.node_q, md.node_q_offset, md.node_q_len,
            )
            return o
        self.store_kv_cache(k, v, input_metadata)
        tree_attention_fwd(
            q.view(-1, self.tp_q_head_num, self.head_dim),
            pool.get_key_buffer(self.layer_id),
            pool.get_value_buffer(self.layer_id),
            o.view(-1, self.tp_q_head_num, self.head_dim),
            md.node_kv, md.node_kv_offset, md.node_kv_len,
            md.node_q, md.node_q_offset, md.node_q_len,
        )
        return o

    def deft_flatten_forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                             input_metadata: InputMetadata) -> torch.Tensor:
        md = get_global_tree_metadata()
        assert md is not None
        _check_query_tile(md)
        assert input_metadata.token_to_kv_pool is not None
        step = _decode_step(ForwardMode.TREE_DECODE_FLATTEN, md, input_metadata, q, k, self.tp_q_head_num,
                            self.tp_k_head_num, self.head_dim)
        if step is not None:
            return step.run(self.layer_id, q, k, v)
        k = k.view(-1, self.tp_k_head_num, self.head_dim)
        v = v.view(-1, self.tp_v_head_num, self.head_dim)
        o = torch.empty((q.shape[0], self.tp_q_head_num * self.head_dim), dtype=q.dtype, device=q.device)
        pool = input_metadata.token_to_kv_pool
        updater = input_metadata.kv_updater
        if (updater is not None and updater.cache_loc is not None and updater.token_to_kv_pool is pool
                and _fused_append_enabled()):
            # store_kv_cache (:121) and the operator (:136-148) in one fused call
            flatten_append_attention(
                q.view(-1, self.tp_q_head_num, self.head_dim), pool.kv_data[self.layer_id],
                o.view(-1, self.tp_q_head_num, self.head_dim), updater.cache_loc, k, v,
                md.block_len, md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv, md.block_lens,
            )
            return o
        self.store_kv_cache(k, v, input_metadata)
        tree_attention_subtree_fwd(
            q.view(-1, self.tp_q_head_num, self.head_dim),
            pool.get_key_buffer(self.layer_id),
            pool.get_value_buffer(self.layer_id),
            o.view(-1, self.tp_q_head_num, self.head_dim),
            md.block_len, md.block_q, md.block_q_cnts, md.block_q_offset,
            md.block_bitmasks, md.block_kv, md.block_lens,
        )
        return o

    def prefill_forward_triton(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                               input_metadata: InputMetadata) -> torch.Tensor:
        """Causal attention over the prompt, then the prompt's K/V go to the pool (deft_attention.py:50-70; the name
        is the reference's, the kernel here is HIP)."""
        o = torch.empty_like(q)
        k3 = k.view(-1, self.tp_k_head_num, self.head_dim)
        v3 = v.view(-1, self.tp_v_head_num, self.head_dim)
        context_attention_fwd(q.view(-1, self.tp_q_head_num, self.head_dim), k3, v3,
                              o.view(-1, self.tp_q_head_num, self.head_dim), input_metadata.start_loc,
                              input_metadata.seq_lens, input_metadata.max_seq_len)
        self.store_kv_cache(k3, v3, input_metadata)
        return o

    def radix_attention_forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                                input_metadata: InputMetadata) -> torch.Tensor:
        """Sequential per-request attention through the page table (deft_attention.py:153-188): the comparator."""
        k = k.view(-1, self.tp_k_head_num, self.head_dim)
        v = v.view(-1, self.tp_v_head_num, self.head_dim)
        o = torch.empty((q.shape[0], self.tp_q_head_num * self.head_dim), dtype=q.dtype, device=q.device)
        assert input_metadata.token_to_kv_pool is not None
        assert input_metadata.req_to_token_pool is not None
        pool = input_metadata.token_to_kv_pool
        updater = input_metadata.kv_updater
        table = input_metadata.req_to_token_pool.req_to_token
        if (updater is not None and updater.cache_loc is not None and updater.token_to_kv_pool is pool
                and _fused_append_enabled()):
            seq_append_attention(
                q.view(-1, self.tp_q_head_num, self.head_dim), pool.kv_data[self.layer_id],
                o.view(-1, self.tp_q_head_num, self.head_dim), updater.cache_loc, k, v, table,
                input_metadata.req_pool_indices, input_metadata.start_loc, input_metadata.seq_lens,
                input_metadata.total_num_tokens,
            )
            return o
        self.store_kv_cache(k, v, input_metadata)
        token_attention_fwd(
            q.view(-1, self.tp_q_head_num, self.head_dim),
            pool.get_key_buffer(self.layer_id),
            pool.get_value_buffer(self.layer_id),
            o.view(-1, self.tp_q_head_num, self.head_dim),
            table,
            input_metadata.req_pool_indices,
            input_metadata.start_loc,
            input_metadata.seq_lens,
            input_metadata.max_seq_len,
            input_metadata.other_kv_index,
            input_metadata.total_num_tokens,
        )
        return o

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, input_metadata: InputMetadata,
                rotary_emb=None, positions: torch.Tensor = None, fuse_rope: bool = None) -> torch.Tensor:
        """`rotary_emb` + `positions` (optional, not in the reference's signature): hand the rotary embedding that
        LlamaAttention.forward applies in front of this call (llama2.py:108-110) to the attention launch itself.  In the
        two DeFT decode modes with Llama's rotary configuration the rotation is fused into stage 1 (no launch of its own, q
        and k stay unrotated); in every other case `rotary_emb(positions, q, k)` runs first, in place, as the reference does.
        `fuse_rope`: True / False forces the choice; None fuses where it measured faster than the rotation's own launch --
        launches of less than ~24 MB of KV per layer (tools/rope_fused_ab.py on MI355X: Medusa-64 19.2 vs 19.8 us per layer
        fused vs rope + attention, ToT-50 27.2 vs 26.0, the north-star tree 43.7 vs 42.6: on the HBM-bound trees the
        cos|sin fetch and the LDS pass in front of every workgroup's first MFMA cost more than the 3 us launch they save)."""
        mode = input_metadata.forward_mode
        if rotary_emb is not None:
            if positions is None:
                raise ValueError("rotary_emb needs positions")
            if (fuse_rope is not False and mode in (ForwardMode.TREE_DECODE_FLATTEN, ForwardMode.TREE_DECODE_NODE)
                    and rope_fusable(rotary_emb, self.head_dim)):
                md = get_global_tree_metadata()
                if fuse_rope is None and md is not None:
                    kv_bytes = 4 * int(getattr(md, "total_kv_len", 1 << 40)) * self.tp_k_head_num * self.head_dim
                    if kv_bytes >= ROPE_FUSE_MAX_KV_BYTES:
                        md = None  # (falls through to the rotation's own launch)
                step = None if md is None else _decode_step(mode, md, input_metadata, q, k, self.tp_q_head_num,
                                                            self.tp_k_head_num, self.head_dim)
                if step is not None:
                    pos = positions.flatten()
                    if pos.dtype != torch.int64 or not pos.is_cuda or pos.shape[0] != step.nq:
                        raise TypeError("positions must be an int64 CUDA tensor with one entry per query row")
                    if rotary_emb.cos_sin_cache.device != q.device:
                        rotary_emb.cos_sin_cache = rotary_emb.cos_sin_cache.to(q.device)
                    return step.run(self.layer_id, q, k, v,
                                    rope=(pos, rotary_emb.cos_sin_cache, rotary_emb.rotary_dim, rotary_emb.is_neox_style))
            # (llama2.py:108-110 rebinds: a rotary module may rotate in place or return new tensors)
            rotated = rotary_emb(positions, q, k)
            if rotated is not None:
                q, k = rotated
        if mode == ForwardMode.DECODE:
            return self.radix_attention_forward(q, k, v, input_metadata)
        if mode == ForwardMode.PREFILL:
            return self.prefill_forward_triton(q, k, v, input_metadata)
        if mode == ForwardMode.TREE_DECODE_FLATTEN:
            return self.deft_flatten_forward(q, k, v, input_metadata)
        if mode == ForwardMode.TREE_DECODE_NODE:
            return self.deft_node_forward(q, k, v, input_metadata)
        raise NotImplementedError(
            f"Unsupported forward mode: {mode} (deft_amd covers TREE_DECODE_FLATTEN, TREE_DECODE_NODE, DECODE and PREFILL with paged KV)"
        )

    def store_kv_cache(self, cache_k: torch.Tensor, cache_v: torch.Tensor, input_metadata: InputMetadata) -> None:
        input_metadata.kv_updater.update(self.layer_id, cache_k, cache_v)
