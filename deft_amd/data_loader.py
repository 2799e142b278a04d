"""The reference's loader names on top of `deft_amd.templates` (DeFT/deft/data_loader.py:134-235), so that a script
written against `deft.data_loader` -- examples/run_DeFT_llama_paged.py:236-263 -- finds what it imports:

    load_dataset(path)                          the parsed .json / .pkl file                        (:134-143)
    load_trees(path)                            reasoning templates, incompleted records skipped     (:146-149)
    load_prompts(path)                          speculative-decoding / few-shot records              (:181-197)
    generate_accepted_len_list(max_gen_len, tree)   fit a record's accepted lengths to a budget, drawing from the module-level
                                                `random` exactly as the reference does (the example script seeds it, :32)  (:200-235)
    ExecuteTree                                 = TreeTemplate: `branch_record`, `prune_record`, `max_depth`, `max_width`,
                                                `width_per_depth`, `node_num`, `prompt`, `accepted_len_list`, `root`, `nodes`

tests/test_surface.py holds these to the reference's signatures; tests/test_replay_golden.py to its behaviour.
"""
from __future__ import annotations

from typing import Any, List

from .templates import TreeTemplate, _read_json_or_pickle, fit_accept_lengths, read_reasoning_file, read_speculative_file

__all__ = ["ExecuteTree", "load_dataset", "load_trees", "load_prompts", "generate_accepted_len_list"]

ExecuteTree = TreeTemplate


def load_dataset(path: str) -> Any:
    return _read_json_or_pickle(path)


def load_trees(path: str) -> List[TreeTemplate]:
    return read_reasoning_file(path)


def load_prompts(path: str) -> List[TreeTemplate]:
    return read_speculative_file(path)


def generate_accepted_len_list(max_gen_len: int, tree: TreeTemplate) -> None:
    fit_accept_lengths(tree, max_gen_len)  # (rng=None: the `random` module's global generator, as data_loader.py:224 uses)
