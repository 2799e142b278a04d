"""The reference's branch functions under their own names and signatures
(DeFT/deft/tree_decoding/generation/branch_func_example.py:12-62, :293-372, :374-442), for code written against
`deft.tree_decoding.generation.branch_func_example` + `deft.tree_decoding.branch_controller.Branch_Controller`
(examples/run_DeFT_llama_paged.py:166-184): `model` is anything with a `.tree` (a `deft_amd.TreeCache`), `logits` the
next-token probabilities of the step (`tree_generate.py:150`), `execution_graph` a template (`deft_amd.data_loader.ExecuteTree`).
The work is `deft_amd.replay`'s -- the functions tests/test_replay_golden.py holds to the reference's behaviour step for step --
these are its reference-shaped front ends.
"""
from __future__ import annotations

from typing import Any, Callable, Optional

from .replay import branch_few_shot, branch_from_tree_template, branch_speculative_decoding
from .templates import TreeTemplate, synthetic_few_shot_template

__all__ = ["example_branch_Func1_SimpleTree", "example_branch_Func3_FromTreeTemplate", "example_branch_Func4_SpeculativeDecoding",
           "Branch_Controller"]


def example_branch_Func1_SimpleTree(model, iter: int, max_gen_len: int, width: int, depth: int, logits, **kwargs) -> bool:
    """:12-62 -- branch into `width` leaves after the prefill, greedy afterwards; the last iteration outputs the branches."""
    return branch_few_shot(model.tree, iter, max_gen_len, logits, synthetic_few_shot_template(width))


def example_branch_Func3_FromTreeTemplate(model, iter: int, max_gen_len: int, width: int, depth: int, logits,
                                          execution_graph: Optional[TreeTemplate] = None) -> bool:
    """:293-372 -- branch / prune as the template's records say, greedy otherwise."""
    assert execution_graph is not None
    return branch_from_tree_template(model.tree, iter, max_gen_len, logits, execution_graph)


def example_branch_Func4_SpeculativeDecoding(model, iter: int, max_gen_len: int, width: int, depth: int, logits,
                                             execution_graph: Optional[TreeTemplate] = None) -> bool:
    """:374-442 -- the speculative-decoding mock: `node_num` one-token leaves, the accepted ones squeezed into the root."""
    assert execution_graph is not None and execution_graph.accepted_len_list is not None
    return branch_speculative_decoding(model.tree, iter, max_gen_len, logits, execution_graph)


class Branch_Controller:
    """DeFT/deft/tree_decoding/branch_controller.py:10-31."""

    def __init__(self, branching_function: Callable) -> None:
        self.branching_function = branching_function
        self.tree_templates: Optional[TreeTemplate] = None

    def apply_branching(self, *args, **kwargs) -> Any:
        if self.branching_function is not None:
            return self.branching_function(*args, **kwargs)
        raise ValueError("Branching function is not set.")

    def set_execution_graph(self, tree_templates: Optional[TreeTemplate]) -> None:
        self.tree_templates = tree_templates
