"""ForwardMode names and the slice of InputMetadata the attention module reads.

Mirrors DeFT/deft/model_runner.py:31-42 (ForwardMode) and the three fields of
InputMetadata (:73-231) that DeFTAttention touches: `forward_mode`,
`kv_updater` and `token_to_kv_pool` (plus the page-table fields of the sequential comparator).  The CLI spelling map follows
DeFT/examples/run_DeFT_llama_paged.py:123-152; BASELINE.json's
`deft_flatten` / `deft_node` (notebook labels) are accepted as aliases.
"""
from __future__ import annotations

from dataclasses import dataclass
from enum import Enum, auto
from typing import Optional

import torch

from .memory_pool import ReqToTokenPool, TokenToKVPool
from .tree_cache import BLOCK_CONFIG, KVCacheUpdater


class ForwardMode(Enum):
    PREFILL = auto()
    EXTEND = auto()
    DECODE = auto()
    TREE_DECODE = auto()
    TREE_DECODE_NODE = auto()
    TREE_DECODE_FLATTEN = auto()
    TREE_DECODE_INDEX_NODE = auto()
    UNPAGED_FD = auto()
    UNPAGED_MEDUSA = auto()
    UNPAGED_DEFT_NODE = auto()
    UNPAGED_DEFT_FLATTEN = auto()


@dataclass
class InputMetadata:
    forward_mode: ForwardMode
    kv_updater: Optional[KVCacheUpdater] = None
    token_to_kv_pool: Optional[TokenToKVPool] = None
    # sequential (per-request) decode, `--mode seq`: the fields radix_attention_forward reads
    # (deft_attention.py:153-188; built by InputMetadata.from_tree, model_runner.py:162-231)
    req_to_token_pool: Optional[ReqToTokenPool] = None
    req_pool_indices: Optional[torch.Tensor] = None  # int32 [batch]: page-table row of each leaf, leaves by id
    start_loc: Optional[torch.Tensor] = None          # int32 [batch]: exclusive prefix sum of seq_lens
    seq_lens: Optional[torch.Tensor] = None           # [batch]: tokens on each leaf's path, this step's token included
    max_seq_len: int = 0
    total_num_tokens: int = 0
    other_kv_index: Optional[int] = None
    return_logprob: bool = False

    @classmethod
    def from_tree(cls, tree, req_to_token_pool, token_to_kv_pool, forward_mode: ForwardMode, positions: torch.Tensor,
                  kv_updater: KVCacheUpdater, return_logprob: bool = False) -> "InputMetadata":
        """model_runner.py:162-231 without its host syncs on the tree modes: seq_lens = positions + 1, leaves in id
        order.  `other_kv_index` (a flashinfer-era leftover, read by nobody on this path) stays None."""
        seq_lens = positions + 1
        batch = positions.shape[0]
        start_loc = torch.zeros((batch,), dtype=torch.int32, device=positions.device)
        start_loc[1:] = torch.cumsum(seq_lens[:-1], dim=0)
        reqs = [v for _, v in sorted(tree.leaf_to_req.items(), key=lambda x: x[0])]
        lens = [leaf_len for leaf_len in seq_lens.tolist()] if forward_mode == ForwardMode.DECODE else []
        return cls(forward_mode=forward_mode, kv_updater=kv_updater, token_to_kv_pool=token_to_kv_pool,
                   req_to_token_pool=req_to_token_pool,
                   req_pool_indices=torch.tensor(reqs, dtype=torch.int32, device=positions.device),
                   start_loc=start_loc, seq_lens=seq_lens, max_seq_len=max(lens) if lens else 0,
                   total_num_tokens=sum(lens) if lens else 0, return_logprob=return_logprob)


def forward_mode_from_cli(mode: str, mem: str = "paged") -> ForwardMode:
    """--mode/--mem -> ForwardMode for the paths this package covers."""
    if mem != "paged":
        raise NotImplementedError(f"--mem {mem}: deft_amd covers the paged KV cache only")
    mode = {"deft_flatten": "flatten", "deft_node": "node", "deft_node_chunk": "node_chunk"}.get(mode, mode)
    if mode == "seq":  # sequential per-leaf attention through the page table (the comparator)
        return ForwardMode.DECODE
    if mode == "flatten":
        return ForwardMode.TREE_DECODE_FLATTEN
    if mode == "node":
        BLOCK_CONFIG["MAX_BLOCK_LEN"] = -1
        return ForwardMode.TREE_DECODE_NODE
    if mode == "node_chunk":
        BLOCK_CONFIG["MAX_BLOCK_LEN"] = 128  # examples/run_DeFT_llama_paged.py:147
        return ForwardMode.TREE_DECODE_NODE
    raise NotImplementedError(f"--mode {mode}: out of scope (covered: flatten, node, node_chunk, seq)")
