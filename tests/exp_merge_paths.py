"""Child process of tests/test_gpu_parity.py::test_small_list_merge_equals_the_general_path: runs against the EXPERIMENTS build
(DEFT_AMD_LIB=deft_amd/lib/libdeft_amd_exp.so), whose DEFT_MERGE_SMALL=0 sends lists of up to eight rows through the general merge
(merge_accumulate + merge_finish) instead of merge_small_wave128.  Both paths on the same trees, Flatten and Node: the outputs must
be equal BIT FOR BIT (the short path restates the general one's arithmetic in the same order, deft_amd/csrc/merge.h).

usage: python tests/exp_merge_paths.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import deft_amd  # noqa: E402
from deft_amd.memory_pool import ReqToTokenPool, TokenToKVPool  # noqa: E402
from deft_amd.tree_cache import TreeCache  # noqa: E402

D = 128
checked = 0
# (Hq, Hkv, prefix, widths per level, steps per level): lists of 1 ... 8 rows and longer ones (which take the general path either way)
for Hq, Hkv, prefix, widths, steps in [(32, 32, 4096, (32,), (200,)), (32, 8, 1500, (7, 6), (40, 30)), (8, 2, 130, (20,), (3,)),
                                       (4, 4, 5, (40,), (1,)), (32, 32, 1016, (64,), (1,)), (8, 8, 9000, (24,), (300,))]:
    size = prefix + sum(s * 64 for s in steps) * 8 + 4096
    req = ReqToTokenPool(600, size + 8, device="cuda")
    pool = TokenToKVPool(size, torch.float16, Hkv, D, 1, device="cuda")
    tree = TreeCache(torch.float16, Hkv, D, 1, req, pool, None, True, False)
    tree.init_prompt(torch.arange(1, prefix + 1, dtype=torch.int32))
    for wd, st in zip(widths, steps):
        for leaf in sorted(tree.leaves.values(), key=lambda n: n.id):
            tree.branch(leaf, wd)
        for _ in range(st):
            for leaf in list(tree.leaves.values()):
                leaf.append_token(7)
            tree.alloc()
    md = deft_amd.TreeMetadata.from_tree_cache(tree)
    nq = md.query_num
    g = torch.Generator(device="cuda").manual_seed(5)
    pool._storage.normal_(generator=g)
    q = torch.randn((nq, Hq, D), dtype=torch.float16, device="cuda", generator=g)
    kb, vb = pool.get_key_buffer(0), pool.get_value_buffer(0)
    outs = {}
    for small in ("1", "0"):
        os.environ["DEFT_MERGE_SMALL"] = small
        of = torch.full_like(q, float("nan"))
        deft_amd.tree_attention_subtree_fwd(q, kb, vb, of, md.block_len, md.block_q, md.block_q_cnts, md.block_q_offset,
                                            md.block_bitmasks, md.block_kv, md.block_lens)
        on = torch.full_like(q, float("nan"))
        deft_amd.tree_attention_fwd(q, kb, vb, on, md.node_kv, md.node_kv_offset, md.node_kv_len, md.node_q, md.node_q_offset,
                                    md.node_q_len)
        torch.cuda.synchronize()
        outs[small] = (of.clone(), on.clone())
    for a, b, what in zip(outs["1"], outs["0"], ("flatten", "node")):
        assert not torch.isnan(a).any(), (what, "nan")
        assert torch.equal(a, b), (what, Hq, Hkv, prefix, widths, float((a.float() - b.float()).abs().max()))
        checked += 1
print(f"merge paths identical: {checked} outputs bit-equal")
