"""ForwardMode names and the slice of InputMetadata the attention module reads.

Mirrors DeFT/deft/model_runner.py:31-42 (ForwardMode) and the three fields of
InputMetadata (:73-231) that DeFTAttention touches: `forward_mode`,
`kv_updater` and `token_to_kv_pool`.  The CLI spelling map follows
DeFT/examples/run_DeFT_llama_paged.py:123-152; BASELINE.json's
`deft_flatten` / `deft_node` (notebook labels) are accepted as aliases.
"""
from __future__ import annotations

from dataclasses import dataclass
from enum import Enum, auto
from typing import Optional

from .memory_pool import TokenToKVPool
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


def forward_mode_from_cli(mode: str, mem: str = "paged") -> ForwardMode:
    """--mode/--mem -> ForwardMode for the paths this package covers."""
    if mem != "paged":
        raise NotImplementedError(f"--mem {mem}: deft_amd covers the paged KV cache only")
    mode = {"deft_flatten": "flatten", "deft_node": "node", "deft_node_chunk": "node_chunk"}.get(mode, mode)
    if mode == "flatten":
        return ForwardMode.TREE_DECODE_FLATTEN
    if mode == "node":
        BLOCK_CONFIG["MAX_BLOCK_LEN"] = -1
        return ForwardMode.TREE_DECODE_NODE
    if mode == "node_chunk":
        BLOCK_CONFIG["MAX_BLOCK_LEN"] = 128  # examples/run_DeFT_llama_paged.py:147
        return ForwardMode.TREE_DECODE_NODE
    raise NotImplementedError(f"--mode {mode}: out of scope (covered: flatten, node, node_chunk)")
