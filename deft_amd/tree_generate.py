"""The decode loop around the path, under the reference's name and signature
(DeFT/deft/tree_decoding/generation/tree_generate.py:20-275): prefill, the branch controller's first call, then per iteration
`leaf_to_q` (leaves by id), `tree.alloc()`, `TreeMetadata.from_tree_cache(tree)` for the DeFT modes, the model's
`forward_tree_decode`, softmax, the branch controller -- until it says stop.

`model` is anything shaped like the reference's ModelRunner where this loop touches it: `.tree` (a `deft_amd.TreeCache`),
`.use_paged_memory`, `.use_tree_index`, `forward_prefill(input_ids, req_pool_indices, seq_lens, prefix_lens,
position_ids_offsets, kv_updater, return_logprob) -> (logits, _)` and `forward_tree_decode(forward_mode, token_ids, positions,
kv_updater, return_logprob, tree_metadata) -> ((logits, ...), seconds)`.  The Llama model itself is out of scope;
tests/test_replay_golden.py drives this loop with a stub model against the reference's own runs.  What the reference's version
does besides -- a `torch.cuda.synchronize()` and a dozen GlobalTimer sections per step -- is left out: the loop is launch-only,
`perf_metrics` (optional, duck-typed: `update(...)`, `print_latency(...)`) receives the iteration's wall clock and the model's
forward time.
"""
from __future__ import annotations

import time
from typing import Any, Optional

import torch

from .forward_mode import ForwardMode
from .tree_cache import TreeMetadata

__all__ = ["tree_generate"]

_DEFT_MODES = (ForwardMode.TREE_DECODE_FLATTEN, ForwardMode.TREE_DECODE_NODE, ForwardMode.TREE_DECODE_INDEX_NODE,
               ForwardMode.UNPAGED_DEFT_FLATTEN, ForwardMode.UNPAGED_DEFT_NODE)


def tree_generate(model, mode: ForwardMode, tokenizer, prompt_ids: torch.Tensor, max_seq_len: int, width: int, depth: int,
                  branch_controller, tree_template, output_file: Optional[str] = None, perf_metrics: Optional[Any] = None) -> None:
    prompt_len = prompt_ids.shape[1]
    input_ids = prompt_ids[0]  # (:40: the first prompt)
    max_gen_len = max_seq_len - prompt_len
    tree = model.tree
    dev = tree.token_to_kv_pool.device
    # ---- init_tree_data (:44-72) + prefill (:74-87) -----------------------------------------------------------
    seq_lens = torch.tensor([prompt_len], dtype=torch.int32, device=dev)
    prefix_lens = torch.tensor([0], dtype=torch.int32, device=dev)
    position_ids_offsets = torch.tensor([0], dtype=torch.int32, device=dev)
    kv_updater = tree.init_prompt(input_ids)
    req = tree.leaf_to_req[tree.root.id] if getattr(model, "use_paged_memory", True) else 0
    req_pool_indices = torch.tensor([req], dtype=torch.int32, device=dev)
    branch_controller.set_execution_graph(tree_templates=tree_template)  # (:184)
    t_start = t_prefill = time.time()
    logits, _ = model.forward_prefill(input_ids, req_pool_indices, seq_lens, prefix_lens, position_ids_offsets, kv_updater, False)
    prob = torch.softmax(logits, dim=-1)
    stop = branch_controller.apply_branching(model=model, iter=0, max_gen_len=max_gen_len, width=width, depth=depth, logits=prob,
                                             execution_graph=branch_controller.tree_templates)  # (:189-197)
    ttft = (time.time() - t_prefill) * 1000
    # ---- decode (:92-169, :200-255) ------------------------------------------------------------------------------
    it = 1
    while stop is False and it < max_gen_len:
        t_step = time.time()
        leaves = sorted(tree.leaves.values(), key=lambda x: x.id)
        leaf_to_q = {leaf.id: i for i, leaf in enumerate(leaves)}
        kv_updater = tree.alloc()
        token_ids = [leaf.token_ids[-1] for leaf in leaves]
        positions = [leaf.positions[-1] for leaf in leaves]
        tree_metadata = None
        if mode in _DEFT_MODES:
            if getattr(model, "use_tree_index", False):
                raise NotImplementedError("tree_index mode is WIP upstream and out of scope here")
            tree_metadata = TreeMetadata.from_tree_cache(tree)
        result, t_forward = model.forward_tree_decode(mode, torch.tensor(token_ids, device=dev).reshape(-1),
                                                      torch.tensor(positions, device=dev).reshape(-1), kv_updater, True, tree_metadata)
        prob = torch.softmax(result[0].float(), dim=-1) + 1e-6  # (:150)
        tree.leaf_to_q = leaf_to_q  # (:219)
        stop = branch_controller.apply_branching(model=model, iter=it, max_gen_len=max_gen_len, width=width, depth=depth, logits=prob,
                                                 execution_graph=branch_controller.tree_templates)
        if perf_metrics is not None:
            zero = dict.fromkeys(("prepare", "branch", "attn_mem", "attn_comp", "traversal", "alloc", "positions", "tree_metadata",
                                  "input_metadata"), 0.0)
            perf_metrics.update(iter_time=(time.time() - t_step) * 1000, forward=t_forward * 1000, **zero)
        if stop:
            break
        it += 1
    if perf_metrics is not None:
        if hasattr(type(perf_metrics), "update_e2e_latency"):
            type(perf_metrics).update_e2e_latency((time.time() - t_start) * 1000)
        perf_metrics.print_latency(prompt_len=prompt_len, generated_len=tree.get_tree_token_number() - prompt_len, ttft=ttft)
    tree.free()  # (:274)
