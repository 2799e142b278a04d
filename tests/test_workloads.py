"""deft_amd.utils.workloads: the synthetic trees bench.py measures are the shapes they claim to be."""
import json
import os

import torch

import deft_amd
from deft_amd.utils import workloads as wl

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "templates.json")))


def test_medusa_topology_is_the_reference_files():
    """BASELINE configs[2] read literally: the node paths bench.py builds `medusa64_tree_*` from are `Tree_Structure` of the
    reference's dataset/generation/Speculative_Decoding/tree_size64.json (recorded by tools/gen_golden_templates.py)."""
    ref = [tuple(p) for p in GOLD["speculative"]["tree_size64"]["Tree_Structure"]]
    assert list(wl.MEDUSA_TREE_SIZE64) == ref
    assert len(ref) == 63 and max(len(p) for p in ref) == 4 and max(max(p) for p in ref) == 9  # depth 4, width 10
    assert all(p[:-1] in ref for p in ref if len(p) > 1)  # every node's parent is a node


def test_medusa_tree_workload_shape():
    w = wl.WORKLOADS["medusa64_tree_node"]
    tree, pool = wl.build_tree(w, 0, "cpu")
    assert len(tree.nodes) == 64 and len(tree.leaves) == 42 == w.width
    assert len(tree.root.kv_indices) == 1016
    assert all(len(n.kv_indices) == 1 and len(n.token_ids) == 1 for i, n in tree.nodes.items() if i != 0)
    assert wl.tree_tokens(w) == 1016 + 63 == int((pool.mem_state != 0).sum())
    md = deft_amd.TreeMetadata.from_tree_cache(tree, device="cpu")
    assert md.query_num == 42 and md.total_kv_len == 1079
    # every leaf's path = prompt + its chain of one-token nodes; position of its token = 1016 + depth - 1
    depth = {n.id: 0 for n in tree.nodes.values()}
    for n in sorted(tree.nodes.values(), key=lambda x: x.id):
        if n.parent is not None:
            depth[n.id] = depth[n.parent.id] + 1
    for lf in tree.leaves.values():
        assert lf.positions == [1016 + depth[lf.id] - 1]
        assert len(tree.leaf_path_slots(lf)) == 1016 + depth[lf.id]
        row = tree.req_to_token_pool.req_to_token[tree.leaf_to_req[lf.id], : 1016 + depth[lf.id]]
        assert row.tolist() == tree.leaf_path_slots(lf)  # the page-table row is the path (what --mode seq reads)
