#!/usr/bin/env python3
"""Build-container only: deft_amd/templates.py against the reference's own loader (DeFT/deft/data_loader.py, imported from
/root/reference, never copied) on EVERY tree of every shipped template file -- branch / prune records, level statistics, accepted
lengths and their fitting to a generation length.  The committed pin of the same property is tests/golden/templates.json
(first complete tree of each Reasoning file); this is the exhaustive form.  Usage: python tools/check_templates_vs_reference.py"""
import os
import random
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/DeFT")
sys.path.insert(0, ROOT)
import deft.data_loader as ref  # noqa: E402

from deft_amd import templates as T  # noqa: E402

BASE = "/root/reference/dataset/generation"
total = 0
for f in sorted(os.listdir(os.path.join(BASE, "Reasoning"))):
    if not f.endswith(".json"):
        continue
    path = os.path.join(BASE, "Reasoning", f)
    theirs, mine = ref.build_trees(ref.load_dataset(path)), T.read_reasoning_file(path)
    assert len(theirs) == len(mine), f
    for a, b in zip(theirs, mine):
        assert a.branch_record == b.branch_record and a.prune_record == b.prune_record, f
        assert (a.max_depth, a.max_width, a.width_per_depth, a.node_num) == (b.max_depth, b.max_width, b.width_per_depth, b.node_num)
        assert all(nd.depth == b.level[nd.id] and nd.width == b.rank[nd.id] for nd in a.nodes if b._reached[nd.id])
        total += 1
print("reasoning templates identical:", total)
for f in sorted(os.listdir(os.path.join(BASE, "Speculative_Decoding"))):
    if not f.endswith(".json"):
        continue
    path = os.path.join(BASE, "Speculative_Decoding", f)
    theirs, mine = ref.load_prompts(path), T.read_speculative_file(path)
    assert len(theirs) == len(mine)
    for a, b in zip(theirs, mine):
        assert a.accepted_len_list == b.accept_lengths and a.node_num == b.node_num
        assert a.prune_record == b.prune_record and a.branch_record == b.branch_record
    for a, b in list(zip(theirs, mine))[:20]:
        random.seed(5)
        ref.generate_accepted_len_list(300, a)
        T.fit_accept_lengths(b, 300, random.Random(5))
        assert a.accepted_len_list == b.accept_lengths
    print(f, "records identical:", len(theirs))
