"""GPU parity: the HIP path (through the C ABI) against the reference's golden
outputs, the CPU oracle and fp64 sequential ground truth.

Tolerance: 1e-3 absolute on fp16 outputs (BASELINE.json north_star) against the
reference's own outputs; 5e-4 against the exact-merge oracle and fp64 truth.
Integer work (slots, metadata) is bit-exact and covered by tests/test_host_logic.py.
"""
import numpy as np
import pytest
import torch

import deft_amd
from deft_amd.tree_attention import flatten_stage1_partials
from helpers import leaf_paths, max_abs, oracle_metadata, oracle_tree, seeded_inputs
from oracle import attention as oa
from product_helpers import md_numpy, product_metadata, product_tree
from scenarios import FULL_GEOMETRY, GQA_GEOMETRY, SCENARIOS, SMALL_GEOMETRIES

pytestmark = pytest.mark.gpu
TOL = 1e-3
TOL_EXACT = 5e-4


def _run(name, geom, mode):
    Hq, Hkv, D = geom
    tree = product_tree(name, device="cuda", heads=(Hkv, D))
    md = product_metadata(name, tree)
    q_np, kv_np = seeded_inputs(name, geom, md.query_num)
    tree.token_to_kv_pool.kv_data[0].copy_(torch.from_numpy(kv_np))
    q = torch.from_numpy(q_np).cuda()
    pool = tree.token_to_kv_pool
    o = torch.full((md.query_num, Hq, D), float("nan"), dtype=torch.float16, device="cuda")
    if mode == "flatten":
        deft_amd.tree_attention_subtree_fwd(q, pool.get_key_buffer(0), pool.get_value_buffer(0), o, md.block_len,
                                            md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks,
                                            md.block_kv, md.block_lens)
    else:
        deft_amd.tree_attention_fwd(q, pool.get_key_buffer(0), pool.get_value_buffer(0), o, md.node_kv,
                                    md.node_kv_offset, md.node_kv_len, md.node_q, md.node_q_offset, md.node_q_len)
    torch.cuda.synchronize()
    return o.cpu().numpy(), q_np, kv_np, md


def _cases():
    for name, sc in SCENARIOS.items():
        if sc.kernels:
            for geom in SMALL_GEOMETRIES:
                yield name, geom
    for name, geom in {**FULL_GEOMETRY, **GQA_GEOMETRY}.items():
        yield name, geom


@pytest.mark.parametrize("mode", ["flatten", "node"])
@pytest.mark.parametrize("name,geom", list(_cases()))
def test_matches_reference_golden_and_oracle(name, geom, mode, golden):
    out, q_np, kv_np, md = _run(name, geom, mode)
    assert np.isfinite(out.astype(np.float32)).all()
    ref = golden(name)["o_%s_%d_%d_%d" % ((mode,) + tuple(geom))]
    assert max_abs(out, ref) < TOL
    otree = oracle_tree(name)
    omd = oracle_metadata(name, otree)
    # the metadata that fed the kernel is the oracle's, bit for bit
    got = md_numpy(md)
    for k, v in got.items():
        assert np.array_equal(v, omd[k]), k
    if geom[0] * md.query_num <= 512:  # full oracle / truth only where they finish in seconds
        fwd = oa.flatten_forward if mode == "flatten" else oa.node_forward
        assert max_abs(out, fwd(q_np, kv_np, omd)) < TOL_EXACT
        assert max_abs(out, oa.sequential_truth(q_np, kv_np, leaf_paths(otree))) < TOL_EXACT
    else:
        paths = leaf_paths(otree)
        rows = [0, len(paths) // 2, len(paths) - 1]
        truth = oa.sequential_truth(q_np[rows], kv_np, [paths[r] for r in rows])
        assert max_abs(out[rows], truth) < TOL_EXACT


@pytest.mark.parametrize("name,geom", [("multilevel", (8, 2, 128)), ("wide40", (4, 4, 64)), ("spec_mock", (4, 4, 128))])
def test_flatten_stage1_partials_match_oracle(name, geom):
    Hq, Hkv, D = geom
    tree = product_tree(name, device="cuda", heads=(Hkv, D))
    md = product_metadata(name, tree)
    q_np, kv_np = seeded_inputs(name, geom, md.query_num)
    tree.token_to_kv_pool.kv_data[0].copy_(torch.from_numpy(kv_np))
    pool = tree.token_to_kv_pool
    po, pl = flatten_stage1_partials(torch.from_numpy(q_np).cuda(), pool.get_key_buffer(0), pool.get_value_buffer(0),
                                     md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv,
                                     md.block_lens)
    torch.cuda.synchronize()
    opo, opl = oa.flatten_stage1(q_np, kv_np, oracle_metadata(name))
    assert max_abs(pl.cpu().numpy(), opl) < 2e-3  # lse of fp16-rounded probabilities
    assert max_abs(po.cpu().numpy(), opo) < 2e-3


def test_strided_query_view_of_fused_qkv():
    """The model hands q as a row-strided view of the fused qkv projection
    (row stride (Hq+2Hkv)*D)."""
    name, geom = "multilevel", (8, 2, 128)
    Hq, Hkv, D = geom
    out_ref, q_np, kv_np, md = _run(name, geom, "flatten")
    tree = product_tree(name, device="cuda", heads=(Hkv, D))
    tree.token_to_kv_pool.kv_data[0].copy_(torch.from_numpy(kv_np))
    qkv = torch.zeros((md.query_num, (Hq + 2 * Hkv) * D), dtype=torch.float16, device="cuda")
    qkv[:, : Hq * D] = torch.from_numpy(q_np).cuda().view(md.query_num, -1)
    q = qkv[:, : Hq * D].view(-1, Hq, D)
    assert q.stride(0) == (Hq + 2 * Hkv) * D
    pool = tree.token_to_kv_pool
    o = torch.zeros((md.query_num, Hq, D), dtype=torch.float16, device="cuda")
    deft_amd.tree_attention_subtree_fwd(q, pool.get_key_buffer(0), pool.get_value_buffer(0), o, md.block_len, md.block_q,
                                        md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv, md.block_lens)
    torch.cuda.synchronize()
    assert np.array_equal(o.cpu().numpy(), out_ref)


def test_deterministic_run_to_run():
    a = _run("wide40", (8, 2, 128), "flatten")[0]
    b = _run("wide40", (8, 2, 128), "flatten")[0]
    assert np.array_equal(a, b)  # no atomics: bit-identical (the reference is order-nondeterministic)


def test_padding_never_reads_uninitialised_pool_memory():
    """Padded block positions (-1 slots) and unused pool slots may hold NaN/Inf."""
    name, geom = "edge128", (4, 4, 128)
    Hq, Hkv, D = geom
    tree = product_tree(name, device="cuda", heads=(Hkv, D))
    md = product_metadata(name, tree)
    q_np, kv_np = seeded_inputs(name, geom, md.query_num)
    used = sorted(s for n in tree.nodes.values() for s in n.kv_indices)
    poisoned = np.full_like(kv_np, np.nan)
    poisoned[used] = kv_np[used]
    tree.token_to_kv_pool.kv_data[0].copy_(torch.from_numpy(poisoned))
    pool = tree.token_to_kv_pool
    o = torch.zeros((md.query_num, Hq, D), dtype=torch.float16, device="cuda")
    deft_amd.tree_attention_subtree_fwd(torch.from_numpy(q_np).cuda(), pool.get_key_buffer(0), pool.get_value_buffer(0), o,
                                        md.block_len, md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks,
                                        md.block_kv, md.block_lens)
    torch.cuda.synchronize()
    out = o.cpu().numpy()
    assert np.isfinite(out.astype(np.float32)).all()
    assert max_abs(out, oa.flatten_forward(q_np, kv_np, oracle_metadata(name))) < TOL_EXACT


def test_very_negative_scores_stay_finite():
    """The reference returns 0 / NaN when every partial LSE is very negative (row max
    initialised to 0, fp16 atomics).  The HIP merge uses the true
    max, so it must match fp64 truth there."""
    name, geom = "cfgA_256x2", (4, 4, 128)
    Hq, Hkv, D = geom
    tree = product_tree(name, device="cuda", heads=(Hkv, D))
    md = product_metadata(name, tree)
    q_np, kv_np = seeded_inputs(name, geom, md.query_num)
    q_np = (q_np.astype(np.float32) * 0 + 3.0).astype(np.float16)
    kv_np = kv_np.copy()
    kv_np[:, 0] = np.float16(-3.0)  # every logit = -3*3*128/sqrt(128) = -101.8
    tree.token_to_kv_pool.kv_data[0].copy_(torch.from_numpy(kv_np))
    pool = tree.token_to_kv_pool
    o = torch.zeros((md.query_num, Hq, D), dtype=torch.float16, device="cuda")
    deft_amd.tree_attention_subtree_fwd(torch.from_numpy(q_np).cuda(), pool.get_key_buffer(0), pool.get_value_buffer(0), o,
                                        md.block_len, md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks,
                                        md.block_kv, md.block_lens)
    torch.cuda.synchronize()
    truth = oa.sequential_truth(q_np, kv_np, leaf_paths(oracle_tree(name)))
    assert max_abs(o.cpu().numpy(), truth) < TOL


def test_kv_append_and_module_forward():
    """DeFTAttention.forward = store_kv_cache (paged append) then attention, both modes
    (deft_attention.py:110-151, :72-108, :390-403)."""
    name, geom = "multilevel", (8, 2, 128)
    Hq, Hkv, D = geom
    tree = product_tree(name, device="cuda", heads=(Hkv, D))
    otree = oracle_tree(name)
    for leaf in list(tree.leaves.values()):
        leaf.append_token(9)
    for leaf in list(otree.leaves.values()):
        leaf.append_token(9)
    updater = tree.alloc()
    oloc = otree.alloc()
    assert updater.cache_loc.cpu().tolist() == oloc.tolist()
    md = deft_amd.TreeMetadata.from_tree_cache(tree)
    omd = oracle_metadata(name, otree)
    nq = md.query_num
    q_np, kv_np = seeded_inputs(name, geom, nq)
    kv_np = kv_np.copy()
    tree.token_to_kv_pool.kv_data[0].copy_(torch.from_numpy(kv_np))
    from deft_amd.utils.synthetic import dyadic_normal

    k_new = dyadic_normal((nq, Hkv * D), 77)
    v_new = dyadic_normal((nq, Hkv * D), 78)
    oa.kv_append(kv_np, oloc, k_new.reshape(nq, Hkv, D), v_new.reshape(nq, Hkv, D))
    truth = oa.sequential_truth(q_np, kv_np, leaf_paths(otree))
    deft_amd.register_tree_metadata(md)
    try:
        for mode in (deft_amd.ForwardMode.TREE_DECODE_FLATTEN, deft_amd.ForwardMode.TREE_DECODE_NODE):
            attn = deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, layer_id=0)
            meta = deft_amd.InputMetadata(mode, updater, tree.token_to_kv_pool)
            o = attn(torch.from_numpy(q_np).cuda().view(nq, -1), torch.from_numpy(k_new).cuda(),
                     torch.from_numpy(v_new).cuda(), meta)
            torch.cuda.synchronize()
            assert np.array_equal(tree.token_to_kv_pool.kv_data[0].cpu().numpy()[oloc], kv_np[oloc])
            assert max_abs(o.view(nq, Hq, D).cpu().numpy(), truth) < TOL_EXACT
    finally:
        deft_amd.unregister_tree_metadata()


def test_unsupported_head_dim_fails_loudly():
    q = torch.zeros(1, 4, 32, dtype=torch.float16, device="cuda")
    kv = torch.zeros(8, 4, 32, dtype=torch.float16, device="cuda")
    i64 = torch.zeros(128, dtype=torch.int64, device="cuda")
    with pytest.raises(deft_amd.DeftLibraryError, match="DEFT_EUNSUPPORTED"):
        deft_amd.tree_attention_subtree_fwd(q, kv, kv, q.clone(), 128, i64[:1], i64[:1] + 1, i64[:1], i64, i64, i64[:1] + 1)
