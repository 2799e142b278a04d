#!/usr/bin/env python3
"""Golden vectors for deft_amd.replay's template model: tests/golden/templates.json, made by importing the REFERENCE's
data_loader (DeFT/deft/data_loader.py) in the build container and recording, for the first complete tree of each
dataset/generation/Reasoning/*.json file, the node table it was built from (ids, token counts, start / end iterations,
children -- the tree's shape, no text) and the branch_record / prune_record / depth / width the reference derives.
For one Speculative_Decoding file: Token_Tree_size, the first record's Accept_length and Tree_Structure (the 63 node paths
of the depth-4 width-10 Medusa tree).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_templates.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/DeFT")
from deft import data_loader as ref  # noqa: E402

BASE = "/root/reference/dataset/generation"
out = {"reasoning": {}, "speculative": {}}
for name in ("docmergeToT", "keywordToT", "set128ToT", "sorting128ToT"):
    dataset = ref.load_dataset(os.path.join(BASE, "Reasoning", name + ".json"))
    item = next(it for it in dataset if not it.get("incompleted"))
    tree = ref.build_trees([item])[0]
    out["reasoning"][name] = {
        "data": {k: {f: v[f] for f in ("id", "value", "start", "end", "children")} for k, v in item["data"].items()},
        "branch_record": {str(k): {str(p): c for p, c in v.items()} for k, v in tree.branch_record.items()},
        "prune_record": {str(k): v for k, v in tree.prune_record.items()},
        "max_depth": tree.max_depth, "max_width": tree.max_width, "node_num": tree.node_num,
        "width_per_depth": {str(k): v for k, v in tree.width_per_depth.items()},
    }
sd = ref.load_dataset(os.path.join(BASE, "Speculative_Decoding", "tree_size64.json"))
trees = ref.load_prompts(os.path.join(BASE, "Speculative_Decoding", "tree_size64.json"))
out["speculative"]["tree_size64"] = {"Token_Tree_size": sd["Token_Tree_size"], "records": len(sd["Records"]),
                                     "Accept_length_0": sd["Records"][0]["Accept_length"],
                                     "node_num": trees[0].node_num}
# the token tree's TOPOLOGY (Medusa choices: one path of top-k indices per node) -- BASELINE configs[2] read literally
out["speculative"]["tree_size64"]["Tree_Structure"] = sd["Tree_Structure"]
path = os.path.join(ROOT, "tests", "golden", "templates.json")
json.dump(out, open(path, "w"), separators=(",", ":"))
print(path, os.path.getsize(path), "bytes;", {k: v["node_num"] for k, v in out["reasoning"].items()})
