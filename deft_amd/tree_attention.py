"""The two replaceable operators of the reference, backed by libdeft_amd.so.

Same names, positional signatures, in-place output and argument meaning as
DeFT/deft/layers/attention/tree_attention.py:

  tree_attention_subtree_fwd (:551-667)   DeFT-Flatten
  tree_attention_fwd         (:14-68)     DeFT-Node

so `DeFTAttention.deft_flatten_forward / deft_node_forward`
(DeFT/deft/layers/attention/deft_attention.py:136-148, :94-105) can import them
from here unchanged.  Launches are asynchronous on torch's current stream, the
functions keep no state, and temporaries come from torch's caching allocator —
like the reference (:588-597, :307-312).  `output` is overwritten; the reference
needs it pre-zeroed (:546), which is tolerated.

There is no PyTorch or CPU fallback: tensors must be fp16 CUDA tensors and the
HIP library must be loadable.
"""
from __future__ import annotations

import torch

from ._lib import DeftLibraryError, check, lib, tensor_version

__all__ = ["tree_attention_subtree_fwd", "tree_attention_fwd", "flatten_append_attention", "kv_append",
           "flatten_stage1_partials"]


def _stream_ptr(t: torch.Tensor) -> int:
    """torch's current stream on the tensor's device.  The library keeps per-device facts (CU count, raised LDS limits) under
    HIP's CURRENT device, so the tensors must live on it -- one process per GPU with torch.cuda.set_device(local_rank), or
    `with torch.cuda.device(t.device):` around the call."""
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        raise DeftLibraryError(f"tensors on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                               "call inside `with torch.cuda.device(tensor.device):`")
    return torch.cuda.current_stream(t.device).cuda_stream


def _check_qkv(query_states, key_buffer, value_buffer, output):
    for name, t in (("query_states", query_states), ("key_buffer", key_buffer), ("value_buffer", value_buffer),
                    ("output", output)):
        if not t.is_cuda:
            raise DeftLibraryError(f"{name} must be a CUDA (ROCm) tensor; deft_amd has no CPU path")
        if t.dtype != torch.float16:
            raise TypeError(f"{name} must be float16, got {t.dtype}")
        if t.dim() != 3 or t.stride(2) != 1:
            raise ValueError(f"{name} must be [n, heads, head_dim] with a contiguous head_dim")
    nq, Hq, D = query_states.shape
    Hkv = key_buffer.shape[1]
    assert D in {16, 32, 64, 128}  # tree_attention.py:100, :582
    if key_buffer.shape != value_buffer.shape or key_buffer.stride() != value_buffer.stride():
        raise ValueError("key_buffer and value_buffer must have identical shape and strides")
    if output.shape != query_states.shape:
        raise ValueError("output must have the shape of query_states")
    return nq, Hq, Hkv, D


def _flatten_plan(md, NB: int, P: int, Hq: int, Hkv: int, q_strides, kv_stride_slot: int, stream: int,
                  cache_loc=None, new_stride: int = 0):
    """Device-side repack of the Flatten metadata, built once per decode step.

    The reference builds TreeMetadata once per step and all layers read the same tensor
    objects (tree_cache.py:1021-1037), so the plan is cached ON the block_q tensor, keyed by
    the identity and in-place version of all six arrays, the head counts and the q / pool
    strides.  Fresh
    tensors (or an in-place edit) simply rebuild it; results never depend on the cache."""
    block_q = md[0]
    versions = [tensor_version(t) for t in md] + ([tensor_version(cache_loc)] if cache_loc is not None else [])
    cacheable = min(versions) >= 0  # inference tensors keep no version counter: their plans are never reused
    key = (lib.deft_plan_variant(), kv_stride_slot, NB, P, Hq, Hkv, tuple(q_strides)) + tuple(t.data_ptr() for t in md) + tuple(versions)
    if cache_loc is not None:  # fused-append plans mark this step's new slots
        key += (cache_loc.data_ptr(), cache_loc.shape[0], new_stride)
    cached = getattr(block_q, "_deft_plan", None) if cacheable else None
    if cached is not None and cached[0] == key:
        return cached[1]
    nbytes = lib.deft_flatten_plan_bytes(NB, P, Hq, Hkv)
    plan = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=block_q.device)
    check(lib.deft_flatten_build_plan(*[t.data_ptr() for t in md], NB, P, Hq, Hkv, q_strides[0], q_strides[1],
                                      kv_stride_slot, cache_loc.data_ptr() if cache_loc is not None else None,
                                      cache_loc.shape[0] if cache_loc is not None else 0, new_stride,
                                      plan.data_ptr(), nbytes, stream), "deft_flatten_build_plan")
    if cacheable:
        try:
            block_q._deft_plan = (key, plan)
        except Exception:  # tensors that refuse attributes just do not cache
            pass
    return plan


def _node_plan(md, NE: int, P: int, total_kv: int, Hq: int, Hkv: int, q_strides, kv_stride_slot: int, stream: int,
               cache_loc=None, new_stride: int = 0):
    """Node-mode counterpart of `_flatten_plan`; cached on the KVMapQ_List (node_q) tensor."""
    node_q = md[3]
    versions = [tensor_version(t) for t in md] + ([tensor_version(cache_loc)] if cache_loc is not None else [])
    cacheable = min(versions) >= 0
    key = (lib.deft_plan_variant(), kv_stride_slot, NE, P, total_kv, Hq, Hkv, tuple(q_strides)) + tuple(t.data_ptr() for t in md) + tuple(versions)
    if cache_loc is not None:  # fused-append plans mark this step's new slots
        key += (cache_loc.data_ptr(), cache_loc.shape[0], new_stride)
    cached = getattr(node_q, "_deft_plan", None) if cacheable else None
    if cached is not None and cached[0] == key:
        return cached[1]
    nbytes = lib.deft_node_plan_bytes(NE, P, total_kv, Hq, Hkv)
    plan = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=node_q.device)
    check(lib.deft_node_build_plan(*[t.data_ptr() for t in md], NE, P, total_kv, Hq, Hkv, q_strides[0], q_strides[1],
                                   kv_stride_slot, cache_loc.data_ptr() if cache_loc is not None else None,
                                   cache_loc.shape[0] if cache_loc is not None else 0, new_stride,
                                   plan.data_ptr(), nbytes, stream), "deft_node_build_plan")
    if cacheable:
        try:
            node_q._deft_plan = (key, plan)
        except Exception:
            pass
    return plan


def _i64(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.int64 or not t.is_cuda:
        raise TypeError(f"{name} must be an int64 CUDA tensor (TreeMetadata contract)")
    return t if t.is_contiguous() else t.contiguous()


@torch.inference_mode()
def tree_attention_subtree_fwd(
    query_states: torch.Tensor,  # (query_num, num_heads, head_dim)
    key_buffer: torch.Tensor,  # (-1, num_kv_heads, head_dim)
    value_buffer: torch.Tensor,  # (-1, num_kv_heads, head_dim)
    output: torch.Tensor,  # (query_num, num_heads, head_dim)
    block_len: int,
    block_q: torch.Tensor,  # (partial_q_num)
    block_q_cnts: torch.Tensor,  # (block_num)
    block_q_offset: torch.Tensor,  # (block_num)
    block_bitmasks: torch.Tensor,  # (kv_len)
    block_kv: torch.Tensor,  # (kv_len)
    block_lens: torch.Tensor,  # (block_num)
) -> None:
    nq, Hq, Hkv, D = _check_qkv(query_states, key_buffer, value_buffer, output)
    if block_len != 128:
        # the reference kernel hard-codes BLOCK_N=128 and ignores this argument (tree_attention.py:655, :922)
        raise ValueError(f"block_len must be 128 (got {block_len})")
    NB = block_q_cnts.shape[0]
    P = block_q.shape[0]
    md = [_i64(t, n) for t, n in ((block_q, "block_q"), (block_q_cnts, "block_q_cnts"),
                                  (block_q_offset, "block_q_offset"), (block_bitmasks, "block_bitmasks"),
                                  (block_kv, "block_kv"), (block_lens, "block_lens"))]
    if block_kv.shape[0] != NB * 128 or block_bitmasks.shape[0] != NB * 128:
        raise ValueError("block_kv / block_bitmasks must hold 128 entries per block")
    ws_bytes = lib.deft_flatten_workspace_bytes(NB, P, nq, Hq, Hkv, D)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=query_states.device)
    scale = 1.0 / (D ** 0.5)  # tree_attention.py:601
    stream = _stream_ptr(query_states)
    plan = _flatten_plan(md, NB, P, Hq, Hkv, (query_states.stride(0), query_states.stride(1)), key_buffer.stride(0), stream)
    rc = lib.deft_flatten_decode_f16(
        query_states.data_ptr(), query_states.stride(0), query_states.stride(1),
        key_buffer.data_ptr(), value_buffer.data_ptr(), key_buffer.stride(0), key_buffer.stride(1),
        output.data_ptr(), output.stride(0), output.stride(1),
        *[t.data_ptr() for t in md],
        NB, P, nq, Hq, Hkv, D, scale, plan.data_ptr(), ws.data_ptr(), ws_bytes, stream,
    )
    check(rc, "deft_flatten_decode_f16")


@torch.inference_mode()
def flatten_append_attention(query_states, kv_layer, output, cache_loc, cache_k, cache_v, block_len, block_q, block_q_cnts,
                             block_q_offset, block_bitmasks, block_kv, block_lens) -> None:
    """`store_kv_cache` + `tree_attention_subtree_fwd` in ONE launch sequence
    (DeFTAttention.deft_flatten_forward, deft_attention.py:110-151): kv_layer[cache_loc, 0/1] = cache_k / cache_v
    and output = Flatten attention that already sees those rows.  kv_layer is one layer of the pool,
    [size, 2, Hkv, D] fp16 (memory_pool.py:61-66)."""
    key_buffer, value_buffer = kv_layer[:, 0], kv_layer[:, 1]
    nq, Hq, Hkv, D = _check_qkv(query_states, key_buffer, value_buffer, output)
    if block_len != 128:
        raise ValueError(f"block_len must be 128 (got {block_len})")
    NB, P = block_q_cnts.shape[0], block_q.shape[0]
    md = [_i64(t, n) for t, n in ((block_q, "block_q"), (block_q_cnts, "block_q_cnts"),
                                  (block_q_offset, "block_q_offset"), (block_bitmasks, "block_bitmasks"),
                                  (block_kv, "block_kv"), (block_lens, "block_lens"))]
    n = cache_loc.shape[0]
    k = cache_k.reshape(n, Hkv, D)
    v = cache_v.reshape(n, Hkv, D)
    if k.stride(2) != 1 or k.stride(1) != D:
        k = k.contiguous()
    if v.stride() != k.stride():
        k, v = k.contiguous(), v.contiguous()
    if cache_loc.dtype != torch.int32 or not cache_loc.is_cuda:
        cache_loc = cache_loc.to(device=query_states.device, dtype=torch.int32)
    ws_bytes = lib.deft_flatten_workspace_bytes(NB, P, nq, Hq, Hkv, D)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=query_states.device)
    stream = _stream_ptr(query_states)
    plan = _flatten_plan(md, NB, P, Hq, Hkv, (query_states.stride(0), query_states.stride(1)), key_buffer.stride(0), stream,
                         cache_loc=cache_loc, new_stride=k.stride(0))
    rc = lib.deft_flatten_decode_append_f16(
        query_states.data_ptr(), query_states.stride(0), query_states.stride(1),
        key_buffer.data_ptr(), value_buffer.data_ptr(), key_buffer.stride(0), key_buffer.stride(1),
        output.data_ptr(), output.stride(0), output.stride(1),
        *[t.data_ptr() for t in md],
        NB, P, nq, Hq, Hkv, D, 1.0 / (D ** 0.5),
        cache_loc.data_ptr(), k.data_ptr(), v.data_ptr(), k.stride(0), n,
        plan.data_ptr(), ws.data_ptr(), ws_bytes, stream,
    )
    check(rc, "deft_flatten_decode_append_f16")


@torch.inference_mode()
def tree_attention_fwd(
    query_states: torch.Tensor,  # (query_num, num_heads, head_dim)
    key_buffer: torch.Tensor,  # (-1, num_heads, head_dim)
    value_buffer: torch.Tensor,  # (-1, num_heads, head_dim)
    output: torch.Tensor,  # (query_num, num_heads, head_dim)
    KV_indices: torch.Tensor,  # (total_len)
    KV_indices_offset: torch.Tensor,  # (KV_num)
    KV_len: torch.Tensor,  # (KV_num)
    KVMapQ_List: torch.Tensor,  # (parital_num)
    KVMapQ_List_Offset: torch.Tensor,  # (kv_num)
    KVMapQ_List_Len: torch.Tensor,  # (kv_num)
) -> None:
    nq, Hq, Hkv, D = _check_qkv(query_states, key_buffer, value_buffer, output)
    NE = KV_indices_offset.shape[0]
    P = KVMapQ_List.shape[0]
    total_kv = KV_indices.shape[0]
    md = [_i64(t, n) for t, n in ((KV_indices, "KV_indices"), (KV_indices_offset, "KV_indices_offset"),
                                  (KV_len, "KV_len"), (KVMapQ_List, "KVMapQ_List"),
                                  (KVMapQ_List_Offset, "KVMapQ_List_Offset"), (KVMapQ_List_Len, "KVMapQ_List_Len"))]
    ws_bytes = lib.deft_node_workspace_bytes(NE, P, total_kv, nq, Hq, Hkv, D)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=query_states.device)
    scale = 1.0 / (D ** 0.5)  # tree_attention.py:102
    stream = _stream_ptr(query_states)
    plan = _node_plan(md, NE, P, total_kv, Hq, Hkv, (query_states.stride(0), query_states.stride(1)),
                      key_buffer.stride(0), stream)
    rc = lib.deft_node_decode_f16(
        query_states.data_ptr(), query_states.stride(0), query_states.stride(1),
        key_buffer.data_ptr(), value_buffer.data_ptr(), key_buffer.stride(0), key_buffer.stride(1),
        output.data_ptr(), output.stride(0), output.stride(1),
        *[t.data_ptr() for t in md],
        NE, P, total_kv, nq, Hq, Hkv, D, scale, plan.data_ptr(), ws.data_ptr(), ws_bytes, stream,
    )
    check(rc, "deft_node_decode_f16")


@torch.inference_mode()
def node_append_attention(query_states, kv_layer, output, cache_loc, cache_k, cache_v, KV_indices, KV_indices_offset,
                          KV_len, KVMapQ_List, KVMapQ_List_Offset, KVMapQ_List_Len) -> None:
    """`store_kv_cache` + `tree_attention_fwd` in ONE launch sequence (DeFTAttention.deft_node_forward,
    deft_attention.py:72-108): kv_layer[cache_loc, 0/1] = cache_k / cache_v and output = Node attention that
    already sees those rows."""
    key_buffer, value_buffer = kv_layer[:, 0], kv_layer[:, 1]
    nq, Hq, Hkv, D = _check_qkv(query_states, key_buffer, value_buffer, output)
    NE = KV_indices_offset.shape[0]
    P = KVMapQ_List.shape[0]
    total_kv = KV_indices.shape[0]
    md = [_i64(t, n) for t, n in ((KV_indices, "KV_indices"), (KV_indices_offset, "KV_indices_offset"),
                                  (KV_len, "KV_len"), (KVMapQ_List, "KVMapQ_List"),
                                  (KVMapQ_List_Offset, "KVMapQ_List_Offset"), (KVMapQ_List_Len, "KVMapQ_List_Len"))]
    n = cache_loc.shape[0]
    k = cache_k.reshape(n, Hkv, D)
    v = cache_v.reshape(n, Hkv, D)
    if k.stride(2) != 1 or k.stride(1) != D:
        k = k.contiguous()
    if v.stride() != k.stride():
        k, v = k.contiguous(), v.contiguous()
    if cache_loc.dtype != torch.int32 or not cache_loc.is_cuda:
        cache_loc = cache_loc.to(device=query_states.device, dtype=torch.int32)
    ws_bytes = lib.deft_node_workspace_bytes(NE, P, total_kv, nq, Hq, Hkv, D)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=query_states.device)
    stream = _stream_ptr(query_states)
    plan = _node_plan(md, NE, P, total_kv, Hq, Hkv, (query_states.stride(0), query_states.stride(1)),
                      key_buffer.stride(0), stream, cache_loc=cache_loc, new_stride=k.stride(0))
    rc = lib.deft_node_decode_append_f16(
        query_states.data_ptr(), query_states.stride(0), query_states.stride(1),
        key_buffer.data_ptr(), value_buffer.data_ptr(), key_buffer.stride(0), key_buffer.stride(1),
        output.data_ptr(), output.stride(0), output.stride(1),
        *[t.data_ptr() for t in md],
        NE, P, total_kv, nq, Hq, Hkv, D, 1.0 / (D ** 0.5),
        cache_loc.data_ptr(), k.data_ptr(), v.data_ptr(), k.stride(0), n,
        plan.data_ptr(), ws.data_ptr(), ws_bytes, stream,
    )
    check(rc, "deft_node_decode_append_f16")


@torch.inference_mode()
def kv_append(kv_layer: torch.Tensor, cache_loc: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor) -> None:
    """kv_layer[cache_loc, 0] = cache_k; kv_layer[cache_loc, 1] = cache_v in one launch.

    kv_layer is one layer of the pool, [size, 2, Hkv, D] fp16 (memory_pool.py:61-66);
    semantics of KVCacheUpdater.update, paged branch (tree_cache.py:70-76)."""
    if not kv_layer.is_cuda:
        raise DeftLibraryError("kv_append needs a CUDA (ROCm) pool; deft_amd has no CPU path")
    size, two, Hkv, D = kv_layer.shape
    assert two == 2 and kv_layer.dtype == torch.float16 and kv_layer.stride(3) == 1
    n = cache_loc.shape[0]
    k = cache_k.reshape(n, Hkv, D)
    v = cache_v.reshape(n, Hkv, D)
    if k.stride(2) != 1 or k.stride(1) != D:
        k = k.contiguous()
    if v.stride() != k.stride():
        v = v.contiguous()
        k = k.contiguous()
    if cache_loc.dtype != torch.int32:
        cache_loc = cache_loc.to(torch.int32)
    cache_loc = cache_loc.to(kv_layer.device)
    rc = lib.deft_kv_append_f16(
        kv_layer[:, 0].data_ptr(), kv_layer[:, 1].data_ptr(), kv_layer.stride(0), kv_layer.stride(2),
        cache_loc.data_ptr(), k.data_ptr(), v.data_ptr(), k.stride(0), n, Hkv, D, _stream_ptr(kv_layer),
    )
    check(rc, "deft_kv_append_f16")


@torch.inference_mode()
def flatten_stage1_partials(query_states, key_buffer, value_buffer, block_q, block_q_cnts, block_q_offset,
                            block_bitmasks, block_kv, block_lens):
    """Stage 1 only; returns (partial_o [Hq,P,D] f32, partial_lse [Hq,P] f32) like the
    reference's temporaries (tree_attention.py:588-597).  For tests and profiling."""
    nq, Hq, D = query_states.shape
    Hkv = key_buffer.shape[1]
    NB, P = block_q_cnts.shape[0], block_q.shape[0]
    ws_bytes = lib.deft_flatten_workspace_bytes(NB, P, nq, Hq, Hkv, D)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=query_states.device)
    st = _stream_ptr(query_states)
    rc = lib.deft_flatten_stage1_f16(
        query_states.data_ptr(), query_states.stride(0), query_states.stride(1),
        key_buffer.data_ptr(), value_buffer.data_ptr(), key_buffer.stride(0), key_buffer.stride(1),
        block_q.data_ptr(), block_q_cnts.data_ptr(), block_q_offset.data_ptr(), block_bitmasks.data_ptr(),
        block_kv.data_ptr(), block_lens.data_ptr(), NB, P, nq, Hq, Hkv, D, 1.0 / (D ** 0.5),
        None, ws.data_ptr(), ws_bytes, st,
    )
    check(rc, "deft_flatten_stage1_f16")
    po = torch.empty((Hq, P, D), dtype=torch.float32, device=query_states.device)
    pl = torch.empty((Hq, P), dtype=torch.float32, device=query_states.device)
    check(lib.deft_flatten_read_partials(ws.data_ptr(), ws_bytes, NB, P, nq, Hq, Hkv, D, po.data_ptr(), pl.data_ptr(), st),
          "deft_flatten_read_partials")
    return po, pl
