"""Replay scenarios on the PRODUCT tree (deft_amd.TreeCache) — no oracle imports here."""
from __future__ import annotations

import numpy as np
import torch

import deft_amd
from scenarios import SCENARIOS


def product_tree(name: str, device: str = "cpu", heads=(1, 8), layers: int = 1):
    sc = SCENARIOS[name]
    Hkv, D = heads
    req = deft_amd.ReqToTokenPool(128, sc.pool_size + 8, device=device)
    pool = deft_amd.TokenToKVPool(sc.pool_size, torch.float16, Hkv, D, layers, device=device)
    tree = deft_amd.TreeCache(torch.float16, Hkv, D, layers, req, pool, None, True, False)
    sc.script(tree, lambda n: torch.arange(1, n + 1, dtype=torch.int32))
    return tree


def product_metadata(name: str, tree=None, device=None):
    sc = SCENARIOS[name]
    tree = tree if tree is not None else product_tree(name)
    saved = dict(deft_amd.BLOCK_CONFIG)
    deft_amd.BLOCK_CONFIG["BLOCK_LEN"] = sc.block_len
    deft_amd.BLOCK_CONFIG["MAX_BLOCK_LEN"] = -1
    try:
        return deft_amd.TreeMetadata.from_tree_cache(tree, max_q_len=sc.max_q_len, max_block_len=sc.max_block_len,
                                                     device=device)
    finally:
        deft_amd.BLOCK_CONFIG.update(saved)


MD_FIELDS = ("node_q", "node_kv", "node_q_len", "node_kv_len", "node_q_offset", "node_kv_offset",
             "block_q", "block_q_cnts", "block_q_offset", "block_bitmasks", "block_kv", "block_lens")


def md_numpy(md):
    return {k: getattr(md, k).cpu().numpy() for k in MD_FIELDS}
