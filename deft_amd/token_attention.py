"""Sequential (per-request) paged attention — the comparator DeFT is measured against.

Same name, positional signature and in-place output as
DeFT/deft/layers/attention/token_attention.py:297-335 (`token_attention_fwd`), the
operator behind `DeFTAttention.radix_attention_forward` (deft_attention.py:153-188,
`--mode seq --mem paged`): request i attends to
`req_to_token[b_req_idx[i], :b_seq_len[i]]`, so a prefix shared by k leaves is read k
times.  Backed by libdeft_amd.so (deft_seq_*), which runs the requests through the
Node-mode kernels as one-query entries; the reference's `att_m` logit matrix
(:312-314) is never materialised, the argument is accepted and ignored.

There is no PyTorch or CPU fallback.
"""
from __future__ import annotations

import torch

from ._lib import DeftLibraryError, check, lib, tensor_version
from .tree_attention import _check_qkv, _stream_ptr

__all__ = ["token_attention_fwd", "seq_append_attention"]


def _i32(t: torch.Tensor, name: str, device) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a tensor")
    if not t.is_cuda:
        raise DeftLibraryError(f"{name} must be a CUDA (ROCm) tensor; deft_amd has no CPU path")
    if t.dtype != torch.int32:  # the reference builds seq_lens as positions + 1 (int64), the rest as int32
        t = t.to(torch.int32)
    return t.contiguous()


def _seq_plan(req_to_token, b_req_idx, b_start_loc, b_seq_len, total_num_tokens: int, Hq: int, Hkv: int, q_strides,
              kv_stride_slot: int, stream: int, cache_loc=None, new_stride: int = 0):
    """Per-step plan (page table -> slot lists -> tile records), cached on b_start_loc like the Flatten / Node plans:
    all layers of a decode step pass the same metadata tensors (model_runner.py:162-231).  Takes the caller's
    tensors as they are (the reference builds seq_lens as positions + 1, i.e. int64); int32 copies are made only
    when a plan is actually built."""
    nq = b_req_idx.shape[0]
    versions = tuple((t.data_ptr(), tensor_version(t)) for t in (req_to_token, b_req_idx, b_start_loc, b_seq_len))
    key = (lib.deft_plan_variant(), nq, int(total_num_tokens), Hq, Hkv, tuple(q_strides), kv_stride_slot) + versions
    if cache_loc is not None:
        key += (cache_loc.data_ptr(), tensor_version(cache_loc), cache_loc.shape[0], new_stride)
    cacheable = all(v >= 0 for _, v in versions)
    cached = getattr(b_start_loc, "_deft_plan", None)
    if cacheable and cached is not None and cached[0] == key:
        return cached[1]
    dev = req_to_token.device
    req_idx, start_loc, seq_len = (_i32(t, n, dev) for t, n in ((b_req_idx, "b_req_idx"), (b_start_loc, "b_start_loc"),
                                                                (b_seq_len, "b_seq_len")))
    nbytes = lib.deft_seq_plan_bytes(nq, int(total_num_tokens), Hq, Hkv)
    plan = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    check(lib.deft_seq_build_plan(req_to_token.data_ptr(), req_to_token.stride(0), req_idx.data_ptr(),
                                  start_loc.data_ptr(), seq_len.data_ptr(), nq, int(total_num_tokens), Hq, Hkv,
                                  q_strides[0], q_strides[1], kv_stride_slot,
                                  cache_loc.data_ptr() if cache_loc is not None else None,
                                  cache_loc.shape[0] if cache_loc is not None else 0, new_stride,
                                  plan.data_ptr(), nbytes, stream), "deft_seq_build_plan")
    if cacheable:
        try:
            b_start_loc._deft_plan = (key, plan)
        except Exception:
            pass
    return plan


def _prep(q, k_buffer, v_buffer, o, req_to_token, b_req_idx, b_start_loc, b_seq_len, total_num_tokens):
    nq, Hq, Hkv, D = _check_qkv(q, k_buffer, v_buffer, o)
    if req_to_token.dtype != torch.int32 or not req_to_token.is_cuda or req_to_token.stride(1) != 1:
        raise TypeError("req_to_token must be an int32 CUDA tensor [max_requests, max_context] (memory_pool.py:13-16)")
    for t, n in ((b_req_idx, "b_req_idx"), (b_start_loc, "b_start_loc"), (b_seq_len, "b_seq_len")):
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise DeftLibraryError(f"{n} must be a CUDA (ROCm) tensor; deft_amd has no CPU path")
    if not (b_req_idx.shape[0] == b_start_loc.shape[0] == b_seq_len.shape[0] == nq):
        raise ValueError("one request per query row: b_req_idx / b_start_loc / b_seq_len must have query_num entries")
    return nq, Hq, Hkv, D, b_req_idx, b_start_loc, b_seq_len, int(total_num_tokens)


@torch.inference_mode()
def token_attention_fwd(q, k_buffer, v_buffer, o, req_to_token, b_req_idx, b_start_loc, b_seq_len, max_len_in_batch,
                        other_kv_index, total_num_tokens, att_m=None) -> None:
    nq, Hq, Hkv, D, b_req_idx, b_start_loc, b_seq_len, total = _prep(q, k_buffer, v_buffer, o, req_to_token, b_req_idx,
                                                                     b_start_loc, b_seq_len, total_num_tokens)
    stream = _stream_ptr(q)
    plan = _seq_plan(req_to_token, b_req_idx, b_start_loc, b_seq_len, total, Hq, Hkv, (q.stride(0), q.stride(1)),
                     k_buffer.stride(0), stream)
    ws_bytes = lib.deft_seq_workspace_bytes(nq, total, Hq, Hkv, D)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=q.device)
    check(lib.deft_seq_decode_f16(q.data_ptr(), q.stride(0), q.stride(1), k_buffer.data_ptr(), v_buffer.data_ptr(),
                                  k_buffer.stride(0), k_buffer.stride(1), o.data_ptr(), o.stride(0), o.stride(1),
                                  plan.data_ptr(), nq, total, Hq, Hkv, D, 1.0 / (D ** 0.5), ws.data_ptr(), ws_bytes, stream),
          "deft_seq_decode_f16")


@torch.inference_mode()
def seq_append_attention(q, kv_layer, o, cache_loc, cache_k, cache_v, req_to_token, b_req_idx, b_start_loc, b_seq_len,
                         total_num_tokens) -> None:
    """`store_kv_cache` + `token_attention_fwd` in one launch sequence (radix_attention_forward,
    deft_attention.py:153-188); kv_layer is one layer of the pool, [size, 2, Hkv, D] fp16."""
    k_buffer, v_buffer = kv_layer[:, 0], kv_layer[:, 1]
    nq, Hq, Hkv, D, b_req_idx, b_start_loc, b_seq_len, total = _prep(q, k_buffer, v_buffer, o, req_to_token, b_req_idx,
                                                                     b_start_loc, b_seq_len, total_num_tokens)
    n = cache_loc.shape[0]
    k = cache_k.reshape(n, Hkv, D)
    v = cache_v.reshape(n, Hkv, D)
    if k.stride(2) != 1 or k.stride(1) != D:
        k = k.contiguous()
    if v.stride() != k.stride():
        k, v = k.contiguous(), v.contiguous()
    if cache_loc.dtype != torch.int32 or not cache_loc.is_cuda:
        cache_loc = cache_loc.to(device=q.device, dtype=torch.int32)
    stream = _stream_ptr(q)
    plan = _seq_plan(req_to_token, b_req_idx, b_start_loc, b_seq_len, total, Hq, Hkv, (q.stride(0), q.stride(1)),
                     k_buffer.stride(0), stream, cache_loc=cache_loc, new_stride=k.stride(0))
    ws_bytes = lib.deft_seq_workspace_bytes(nq, total, Hq, Hkv, D)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=q.device)
    check(lib.deft_seq_decode_append_f16(q.data_ptr(), q.stride(0), q.stride(1), k_buffer.data_ptr(), v_buffer.data_ptr(),
                                         k_buffer.stride(0), k_buffer.stride(1), o.data_ptr(), o.stride(0), o.stride(1),
                                         plan.data_ptr(), nq, total, Hq, Hkv, D, 1.0 / (D ** 0.5),
                                         cache_loc.data_ptr(), k.data_ptr(), v.data_ptr(), k.stride(0), n,
                                         ws.data_ptr(), ws_bytes, stream), "deft_seq_decode_append_f16")
