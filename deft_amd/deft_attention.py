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

import os

import torch
from torch import nn

from .forward_mode import ForwardMode, InputMetadata
from .context_attention import context_attention_fwd
from .token_attention import seq_append_attention, token_attention_fwd
from .tree_attention import (flatten_append_attention, node_append_attention, tree_attention_fwd,
                             tree_attention_subtree_fwd)
from .tree_cache import get_global_tree_metadata


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
        k = k.view(-1, self.tp_k_head_num, self.head_dim)
        v = v.view(-1, self.tp_v_head_num, self.head_dim)
        o = torch.empty((q.shape[0], self.tp_q_head_num * self.head_dim), dtype=q.dtype, device=q.device)
        md = get_global_tree_metadata()
        assert md is not None
        assert input_metadata.token_to_kv_pool is not None
        pool = input_metadata.token_to_kv_pool
        updater = input_metadata.kv_updater
        if (updater is not None and updater.cache_loc is not None and updater.token_to_kv_pool is pool
                and not os.environ.get("DEFT_NO_FUSED_APPEND")):
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
        k = k.view(-1, self.tp_k_head_num, self.head_dim)
        v = v.view(-1, self.tp_v_head_num, self.head_dim)
        o = torch.empty((q.shape[0], self.tp_q_head_num * self.head_dim), dtype=q.dtype, device=q.device)
        md = get_global_tree_metadata()
        assert md is not None
        assert input_metadata.token_to_kv_pool is not None
        pool = input_metadata.token_to_kv_pool
        updater = input_metadata.kv_updater
        if (updater is not None and updater.cache_loc is not None and updater.token_to_kv_pool is pool
                and not os.environ.get("DEFT_NO_FUSED_APPEND")):
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
                and not os.environ.get("DEFT_NO_FUSED_APPEND")):
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

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, input_metadata: InputMetadata) -> torch.Tensor:
        mode = input_metadata.forward_mode
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
