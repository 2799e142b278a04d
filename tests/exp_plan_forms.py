"""Child process of tests/test_gpu_parity.py::test_plan_build_forms_give_identical_plans: runs against the EXPERIMENTS build
(DEFT_AMD_LIB=deft_amd/lib/libdeft_amd_exp.so), the only one that has `deft_debug_plan_form`.

usage: python tests/exp_plan_forms.py Hq Hkv prefix width steps
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import deft_amd  # noqa: E402


def _flatten_args(md):
    return (md.block_len, md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv, md.block_lens)


def run(shape):
    """The Flatten plan is written either by all waves of the unit kernel from a table of runs (default), or by one
    lane as it walks the blocks (tables beyond the LDS), or by one lane after the run table overflowed
    (`deft_debug_plan_form(serial, runcap)` forces either): the three must produce the same plan, so the outputs are
    bit-identical (the partial rows are a function of the plan) and the plan bytes the kernels read are equal."""
    from deft_amd._lib import check, lib
    from deft_amd.memory_pool import ReqToTokenPool, TokenToKVPool
    from deft_amd.tree_cache import TreeCache

    Hq, Hkv, prefix, width, steps = shape
    D = 128
    size = prefix + (steps + 1) * width + 256
    req = ReqToTokenPool(width + 8, size + 8, device="cuda")
    pool = TokenToKVPool(size, torch.float16, Hkv, D, 1, device="cuda")
    tree = TreeCache(torch.float16, Hkv, D, 1, req, pool, None, True, False)
    tree.init_prompt(torch.arange(1, prefix + 1, dtype=torch.int32))
    tree.branch(tree.root, width)
    for _ in range(steps):
        for leaf in list(tree.leaves.values()):
            leaf.append_token(7)
        tree.alloc()
    md = deft_amd.TreeMetadata.from_tree_cache(tree)
    g = torch.Generator(device="cuda").manual_seed(11)
    pool._storage.normal_(generator=g)
    q = torch.randn((width, Hq, D), dtype=torch.float16, device="cuda", generator=g)
    kb, vb = pool.get_key_buffer(0), pool.get_value_buffer(0)
    mdl = [md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv, md.block_lens]
    NB, P = md.block_q_cnts.shape[0], md.block_q.shape[0]
    nbytes = lib.deft_flatten_plan_bytes(NB, P, Hq, Hkv)
    cap = NB * (Hq // Hkv)
    outs, plans = [], []
    for form in ((0, 0), (1, 0), (0, 2)):
        try:
            lib.deft_debug_plan_form(*form)
            o = torch.full_like(q, float("nan"))
            deft_amd.tree_attention_subtree_fwd(q, kb, vb, o, *_flatten_args(md))  # (the plan cache is keyed by these knobs)
            plan = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            check(lib.deft_flatten_build_plan(*[t.data_ptr() for t in mdl], NB, P, Hq, Hkv, q.stride(0), q.stride(1), kb.stride(0),
                                              None, 0, 0, plan.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream),
                  "deft_flatten_build_plan")
            torch.cuda.synchronize()
            outs.append(o)
            n_units = int(plan[:4].view(torch.int32).item())
            assert 0 < n_units <= cap
            # header words 0..1 (units, chunk leaders) and every record the kernels may read (unit slots + sentinel)
            plans.append((plan[:8].clone(), plan[4096:4096 + 2048 * (n_units + 1)].clone()))
        finally:
            lib.deft_debug_plan_form(0, 0)
    assert torch.isfinite(outs[0].float()).all()
    for o, (hd, rec) in zip(outs[1:], plans[1:]):
        assert torch.equal(o, outs[0])
        assert torch.equal(hd, plans[0][0])
        assert torch.equal(rec, plans[0][1])
    print("flatten units", int(plans[0][0][:4].view(torch.int32).item()), "leaders", int(plans[0][0][4:8].view(torch.int32).item()))
    # the item loop of a capped grid (a workgroup takes several chunk leaders), in plain (b, b + W, ...) and in mirrored order
    # (b, 2W - 1 - b, 2W + b, ...: GQA launches of one tree): forced to many rounds by a tiny grid, every item exactly once --
    # the same bits as one workgroup per item
    for grid, mirror in ((37, 1), (37, 0), (5, 1), (1, 1)):
        os.environ["DEFT_NP_GRID"], os.environ["DEFT_NP_MIRROR"] = str(grid), str(mirror)
        try:
            o = torch.full_like(q, float("nan"))
            deft_amd.tree_attention_subtree_fwd(q, kb, vb, o, *_flatten_args(md))
            torch.cuda.synchronize()
        finally:
            del os.environ["DEFT_NP_GRID"], os.environ["DEFT_NP_MIRROR"]
        assert torch.equal(o, outs[0]), (grid, mirror)

    def temporal_leaders(rec):  # chunk leaders (desc[4] > 0) whose rows are asked for with the temporal cache policy (desc[6])
        d = rec.view(-1, 2048)[:, 1536:1568].contiguous().view(torch.int32).view(-1, 8)
        return int(((d[:, 4] > 0) & (d[:, 6] != 0)).sum().item())

    print("flatten temporal leaders", temporal_leaders(plans[0][1]))
    # the Node plan (entries cut into tiles, small entries packed) has the same three forms
    nd = [md.node_kv, md.node_kv_offset, md.node_kv_len, md.node_q, md.node_q_offset, md.node_q_len]
    NE, Pn, total_kv = md.node_kv_offset.shape[0], md.node_q.shape[0], md.node_kv.shape[0]
    nbytes = lib.deft_node_plan_bytes(NE, Pn, total_kv, Hq, Hkv)
    outs, plans = [], []
    for form in ((0, 0), (1, 0), (0, 2)):
        try:
            lib.deft_debug_plan_form(*form)
            o = torch.full_like(q, float("nan"))
            deft_amd.tree_attention_fwd(q, kb, vb, o, *nd)
            plan = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            check(lib.deft_node_build_plan(*[t.data_ptr() for t in nd], NE, Pn, total_kv, Hq, Hkv, q.stride(0), q.stride(1),
                                           kb.stride(0), None, 0, 0, plan.data_ptr(), nbytes,
                                           torch.cuda.current_stream().cuda_stream), "deft_node_build_plan")
            torch.cuda.synchronize()
            outs.append(o)
            n_units = int(plan[:4].view(torch.int32).item())
            assert n_units > 0 and 4096 + 2048 * (n_units + 1) <= nbytes
            plans.append((plan[:8].clone(), plan[4096:4096 + 2048 * (n_units + 1)].clone()))
        finally:
            lib.deft_debug_plan_form(0, 0)
    assert torch.isfinite(outs[0].float()).all()
    for o, (hd, rec) in zip(outs[1:], plans[1:]):
        assert torch.equal(o, outs[0])
        assert torch.equal(hd, plans[0][0])
        assert torch.equal(rec, plans[0][1])
    print("node temporal leaders", temporal_leaders(plans[0][1]))


if __name__ == "__main__":
    run(tuple(int(x) for x in sys.argv[1:6]))
    print("forms identical")
