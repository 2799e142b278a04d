"""GPU parity: the HIP path (through the C ABI) against the reference's golden
outputs, the CPU oracle and fp64 sequential ground truth.

Tolerance: 1e-3 absolute on fp16 outputs (BASELINE.json north_star) against the
reference's own outputs; 5e-4 against the exact-merge oracle and fp64 truth.
Integer work (slots, metadata) is bit-exact and covered by tests/test_host_logic.py.
"""
import os

import numpy as np
import pytest
import torch

import deft_amd
from deft_amd.tree_attention import flatten_stage1_partials
from helpers import leaf_paths, max_abs, oracle_metadata, oracle_tree, seeded_inputs
from oracle import attention as oa
from product_helpers import md_numpy, product_metadata, product_tree
from scenarios import SCENARIOS, SMALL_GEOMETRIES, big_cases, small_d_cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    yield from big_cases()  # BASELINE's own configurations at full geometry
    yield from small_d_cases()  # head_dim 32 and 16


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
    """The reference asserts head_dim in {16, 32, 64, 128} (tree_attention.py:100, :582); so does the shim, and the C ABI
    reports DEFT_EUNSUPPORTED for anything else."""
    q = torch.zeros(1, 4, 48, dtype=torch.float16, device="cuda")
    kv = torch.zeros(8, 4, 48, dtype=torch.float16, device="cuda")
    i64 = torch.zeros(128, dtype=torch.int64, device="cuda")
    with pytest.raises(AssertionError):
        deft_amd.tree_attention_subtree_fwd(q, kv, kv, q.clone(), 128, i64[:1], i64[:1] + 1, i64[:1], i64, i64, i64[:1] + 1)
    assert deft_amd.lib.deft_supported(4, 4, 48) == 0 and deft_amd.lib.deft_supported(4, 4, 16) == 1
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    rc = deft_amd.lib.deft_flatten_decode_f16(q.data_ptr(), 4 * 48, 48, kv.data_ptr(), kv.data_ptr(), 4 * 48, 48, q.data_ptr(), 4 * 48, 48,
                                              *[i64.data_ptr()] * 6, 1, 1, 1, 4, 4, 48, 0.1, None, ws.data_ptr(), ws.numel(), None)
    assert rc == -2  # DEFT_EUNSUPPORTED


@pytest.mark.parametrize("geom", [(8, 2, 128), (4, 4, 64)])
def test_any_scale_through_the_c_abi(geom):
    """The C ABI takes `scale` as a plain float.  The kernels apply |scale| inside the exp2 argument and flip the sign of the Q
    fragments for a negative one: attention(q, scale = -s) must equal attention(-q, scale = +s) bit for bit (negating fp16 is exact),
    and scale = 0 must give the uniform average of the values each query sees -- not NaN from -inf x 0."""
    Hq, Hkv, D = geom
    name = "multilevel"
    tree = product_tree(name, device="cuda", heads=(Hkv, D))
    md = product_metadata(name, tree)
    q_np, kv_np = seeded_inputs(name, geom, md.query_num)
    tree.token_to_kv_pool.kv_data[0].copy_(torch.from_numpy(kv_np))
    pool = tree.token_to_kv_pool
    kb, vb = pool.get_key_buffer(0), pool.get_value_buffer(0)
    q = torch.from_numpy(q_np).cuda()
    nq = md.query_num
    NB, P = md.block_q_cnts.shape[0], md.block_q.shape[0]
    ws = torch.empty(deft_amd.lib.deft_flatten_workspace_bytes(NB, P, nq, Hq, Hkv, D), dtype=torch.uint8, device="cuda")

    def run(qt, scale):
        o = torch.full((nq, Hq, D), float("nan"), dtype=torch.float16, device="cuda")
        rc = deft_amd.lib.deft_flatten_decode_f16(
            qt.data_ptr(), qt.stride(0), qt.stride(1), kb.data_ptr(), vb.data_ptr(), kb.stride(0), kb.stride(1), o.data_ptr(), o.stride(0),
            o.stride(1), md.block_q.data_ptr(), md.block_q_cnts.data_ptr(), md.block_q_offset.data_ptr(), md.block_bitmasks.data_ptr(),
            md.block_kv.data_ptr(), md.block_lens.data_ptr(), NB, P, nq, Hq, Hkv, D, scale, None, ws.data_ptr(), ws.numel(),
            torch.cuda.current_stream().cuda_stream)
        assert rc == 0, deft_amd.lib.deft_last_error()
        torch.cuda.synchronize()
        return o

    s = D ** -0.5
    assert torch.equal(run(q, -s), run(-q, s))
    o0 = run(q, 0.0)
    assert torch.isfinite(o0).all()
    paths = leaf_paths(oracle_tree(name))
    kv = torch.from_numpy(kv_np).float()
    for qi in (0, nq - 1):
        want = kv[paths[qi], 1].mean(dim=0)  # [Hkv, D]: every key of the path weighs the same
        got = o0[qi].float().cpu().view(Hkv, Hq // Hkv, D)
        assert (got - want[:, None, :]).abs().max() < 2e-3


def _flatten_args(md):
    return (md.block_len, md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv, md.block_lens)


@pytest.mark.parametrize("mode", ["flatten", "node"])
@pytest.mark.parametrize("name", ["multilevel", "spec_mock"])
@pytest.mark.parametrize("geom", [(4, 4, 128), (8, 2, 128), (4, 4, 64)])
def test_fused_append_equals_append_then_attention(geom, name, mode):
    """deft_{flatten,node}_decode_append_f16 == deft_kv_append_f16 followed by deft_{flatten,node}_decode_f16, bit
    for bit, and the pool holds the new rows afterwards (deft_attention.py:72-151 in one launch sequence)."""
    from deft_amd.tree_attention import flatten_append_attention, node_append_attention
    from deft_amd.utils.synthetic import dyadic_normal

    Hq, Hkv, D = geom
    outs, pools = [], []
    for fused in (False, True):
        tree = product_tree(name, device="cuda", heads=(Hkv, D))
        for leaf in list(tree.leaves.values()):
            leaf.append_token(9)
        updater = tree.alloc()
        md = deft_amd.TreeMetadata.from_tree_cache(tree)
        nq = md.query_num
        q_np, kv_np = seeded_inputs(name, geom, nq)
        pool = tree.token_to_kv_pool
        pool.kv_data[0].copy_(torch.from_numpy(kv_np))
        k_new = torch.from_numpy(dyadic_normal((nq, Hkv, D), 5)).cuda()
        v_new = torch.from_numpy(dyadic_normal((nq, Hkv, D), 6)).cuda()
        q = torch.from_numpy(q_np).cuda()
        o = torch.zeros((nq, Hq, D), dtype=torch.float16, device="cuda")
        node_args = (md.node_kv, md.node_kv_offset, md.node_kv_len, md.node_q, md.node_q_offset, md.node_q_len)
        if fused and mode == "flatten":
            flatten_append_attention(q, pool.kv_data[0], o, updater.cache_loc, k_new, v_new, *_flatten_args(md))
        elif fused:
            node_append_attention(q, pool.kv_data[0], o, updater.cache_loc, k_new, v_new, *node_args)
        else:
            deft_amd.kv_append(pool.kv_data[0], updater.cache_loc, k_new, v_new)
            if mode == "flatten":
                deft_amd.tree_attention_subtree_fwd(q, pool.get_key_buffer(0), pool.get_value_buffer(0), o,
                                                    *_flatten_args(md))
            else:
                deft_amd.tree_attention_fwd(q, pool.get_key_buffer(0), pool.get_value_buffer(0), o, *node_args)
        torch.cuda.synchronize()
        outs.append(o.cpu().numpy())
        pools.append(pool.kv_data[0].cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(pools[0], pools[1])


def test_plan_is_reused_across_layers_and_rebuilt_on_change():
    """The per-step plan is cached on the metadata tensor: 3 'layers' with different pools share it;
    an in-place edit of the metadata invalidates it; results never depend on the cache."""
    name, geom = "multilevel", (4, 4, 128)
    Hq, Hkv, D = geom
    tree = product_tree(name, device="cuda", heads=(Hkv, D), layers=3)
    md = product_metadata(name, tree)
    q_np, kv_np = seeded_inputs(name, geom, md.query_num)
    q = torch.from_numpy(q_np).cuda()
    pool = tree.token_to_kv_pool
    ref = None
    plans = []
    for layer in range(3):
        pool.kv_data[layer].copy_(torch.from_numpy(kv_np))
        o = torch.zeros((md.query_num, Hq, D), dtype=torch.float16, device="cuda")
        deft_amd.tree_attention_subtree_fwd(q, pool.get_key_buffer(layer), pool.get_value_buffer(layer), o, *_flatten_args(md))
        torch.cuda.synchronize()
        plans.append(md.block_q._deft_plan[1].data_ptr())
        ref = o.cpu().numpy() if ref is None else ref
        assert np.array_equal(o.cpu().numpy(), ref)
    assert len(set(plans)) == 1  # one plan for all layers
    key0 = md.block_q._deft_plan[0]
    md.block_lens.add_(0)  # in-place touch bumps the version -> rebuild
    o = torch.zeros((md.query_num, Hq, D), dtype=torch.float16, device="cuda")
    deft_amd.tree_attention_subtree_fwd(q, pool.get_key_buffer(0), pool.get_value_buffer(0), o, *_flatten_args(md))
    torch.cuda.synchronize()
    # (the new plan may sit at the old address -- the allocator reuses it -- so it is the CACHE KEY that must have changed)
    assert md.block_q._deft_plan[0] != key0
    assert np.array_equal(o.cpu().numpy(), ref)


def test_full_size_properties_northstar_tree():
    """BASELINE.json's full size (Llama-2-7B, 4096-token prefix x 32 branches, 3 tokens each): properties that
    need no oracle at this size — Flatten == Node within rounding, permutation of pool slots leaves the output
    unchanged (the gather is by slot list), deterministic, finite."""
    from deft_amd.utils.workloads import Workload, build_tree

    w = Workload("t", "llama2-7b", "flatten", "few_shot", 4096, 32, 3)
    tree, pool = build_tree(w, 1, "cuda")
    md = deft_amd.TreeMetadata.from_tree_cache(tree)
    g = torch.Generator(device="cuda").manual_seed(3)
    pool._storage.normal_(generator=g)
    q = torch.randn((32, 32, 128), dtype=torch.float16, device="cuda", generator=g)
    kb, vb = pool.get_key_buffer(0), pool.get_value_buffer(0)
    o1 = torch.zeros_like(q)
    deft_amd.tree_attention_subtree_fwd(q, kb, vb, o1, *_flatten_args(md))
    o2 = torch.zeros_like(q)
    deft_amd.tree_attention_fwd(q, kb, vb, o2, md.node_kv, md.node_kv_offset, md.node_kv_len, md.node_q, md.node_q_offset,
                                md.node_q_len)
    o3 = torch.zeros_like(q)
    deft_amd.tree_attention_subtree_fwd(q, kb, vb, o3, *_flatten_args(md))
    torch.cuda.synchronize()
    assert torch.isfinite(o1.float()).all()
    assert torch.equal(o1, o3)
    assert (o1.float() - o2.float()).abs().max().item() < TOL_EXACT
    # four leaves against torch fp32 sequential attention on the GPU (test_DeFT_kernel.py:212-276 recipe)
    leaves = sorted(tree.leaves.values(), key=lambda n: n.id)
    for r in (0, 11, 31):
        slots = torch.tensor(tree.leaf_path_slots(leaves[r]), device="cuda")
        k = kb[slots].float().transpose(0, 1)  # [H,S,D]
        v = vb[slots].float().transpose(0, 1)
        s = torch.einsum("hd,hsd->hs", q[r].float(), k) / 128 ** 0.5
        ref = torch.einsum("hs,hsd->hd", torch.softmax(s, dim=-1), v)
        assert (o1[r].float() - ref).abs().max().item() < TOL


def test_full_size_fold_structure_does_not_change_the_result():
    """BASELINE's north-star tree (Llama-2-7B, 4096 x 32 x 200 tokens) through every decomposition the library has:
    Flatten blocks (chunks + union groups), Node entries cut into tiles, and the sequential comparator over the page
    table (every leaf its own full path).  Folding changes the order of fp32 additions, nothing else: all agree within
    the exact-merge tolerance, each is bit-deterministic; three leaves against torch fp32 attention."""
    from deft_amd.utils.workloads import Workload, build_tree

    w = Workload("t", "llama2-7b", "flatten", "few_shot", 4096, 32, 200)
    tree, pool = build_tree(w, 1, "cuda")
    md = deft_amd.TreeMetadata.from_tree_cache(tree)
    g = torch.Generator(device="cuda").manual_seed(7)
    pool._storage.normal_(generator=g)
    q = torch.randn((32, 32, 128), dtype=torch.float16, device="cuda", generator=g)
    kb, vb = pool.get_key_buffer(0), pool.get_value_buffer(0)
    o_ref = torch.zeros_like(q)
    deft_amd.tree_attention_subtree_fwd(q, kb, vb, o_ref, *_flatten_args(md))  # default: single launch
    torch.cuda.synchronize()
    outs = []
    for _ in range(2):
        o = torch.full_like(q, float("nan"))
        deft_amd.tree_attention_subtree_fwd(q, kb, vb, o, *_flatten_args(md))
        outs.append(o)
    o_node = torch.full_like(q, float("nan"))
    deft_amd.tree_attention_fwd(q, kb, vb, o_node, md.node_kv, md.node_kv_offset, md.node_kv_len, md.node_q,
                                md.node_q_offset, md.node_q_len)
    meta, lens = _seq_metadata(tree)
    o_seq = torch.full_like(q, float("nan"))
    deft_amd.token_attention_fwd(q, kb, vb, o_seq, tree.req_to_token_pool.req_to_token, meta.req_pool_indices,
                                 meta.start_loc, meta.seq_lens, meta.max_seq_len, None, meta.total_num_tokens)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], o_ref)
    assert torch.equal(outs[0], outs[1])
    for o in (outs[0], o_node, o_seq):
        assert torch.isfinite(o.float()).all()
        assert (o.float() - o_ref.float()).abs().max().item() < TOL_EXACT
    leaves = sorted(tree.leaves.values(), key=lambda n: n.id)
    for r in (0, 17, 31):
        slots = torch.tensor(tree.leaf_path_slots(leaves[r]), device="cuda")
        k = kb[slots].float().transpose(0, 1)
        v = vb[slots].float().transpose(0, 1)
        sc = torch.einsum("hd,hsd->hs", q[r].float(), k) / 128 ** 0.5
        ref = torch.einsum("hs,hsd->hd", torch.softmax(sc, dim=-1), v)
        assert (outs[0][r].float() - ref).abs().max().item() < TOL_EXACT


@pytest.mark.parametrize("mode", ["flatten", "node"])
@pytest.mark.parametrize("geom", [(8, 2, 128), (4, 4, 128)])
def test_forest_batch_equals_its_trees_one_by_one(geom, mode):
    """Several trees in one pool, ONE operator call over the concatenated metadata (deft_amd.Forest) against
    fp64 sequential attention per leaf (oracle) and against each tree attended on its own."""
    Hq, Hkv, D = geom
    names = ["multilevel", "wide40", "chain_300", "edge_fill", "spec_mock"]
    req = deft_amd.ReqToTokenPool(256, 4096, device="cuda")
    pool = deft_amd.TokenToKVPool(8192, torch.float16, Hkv, D, 1, device="cuda")
    trees = []
    for n in names:
        t = deft_amd.TreeCache(torch.float16, Hkv, D, 1, req, pool, None, True, False)
        SCENARIOS[n].script(t, lambda k: torch.arange(1, k + 1, dtype=torch.int32))
        trees.append(t)
    forest = deft_amd.Forest(trees)
    md = forest.metadata()
    g = torch.Generator(device="cuda").manual_seed(11)
    pool._storage.normal_(generator=g)
    q = torch.randn((md.query_num, Hq, D), dtype=torch.float16, device="cuda", generator=g)
    kb, vb = pool.get_key_buffer(0), pool.get_value_buffer(0)

    def attend(m, qq):
        o = torch.full_like(qq, float("nan"))
        if mode == "flatten":
            deft_amd.tree_attention_subtree_fwd(qq, kb, vb, o, *_flatten_args(m))
        else:
            deft_amd.tree_attention_fwd(qq, kb, vb, o, m.node_kv, m.node_kv_offset, m.node_kv_len, m.node_q,
                                        m.node_q_offset, m.node_q_len)
        return o

    o = attend(md, q)
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all()
    kv_np = pool.kv_data[0].cpu().numpy()
    truth = oa.sequential_truth(q.cpu().numpy(), kv_np, forest.leaf_paths())
    assert max_abs(o.cpu().numpy(), truth) < TOL_EXACT
    for t, tree in enumerate(trees):
        m1 = deft_amd.TreeMetadata.from_tree_cache(tree)
        lo = md.q_bases[t]
        o1 = attend(m1, q[lo : lo + m1.query_num].contiguous())
        torch.cuda.synchronize()
        assert (o1.float() - o[lo : lo + m1.query_num].float()).abs().max().item() < TOL_EXACT


# ---------------------------------------------------------------------------
# sequential comparator (token_attention_fwd / radix_attention_forward, `--mode seq`)
# ---------------------------------------------------------------------------
SEQ_GOLDEN = {"cfgA_256x2": [(4, 4, 128), (8, 2, 128)], "multilevel": [(4, 4, 128), (8, 2, 128), (4, 4, 64)],
              "wide40": [(8, 2, 128)], "chain_300": [(4, 4, 128)], "medusa64_tree": [(8, 2, 128)]}
TOL_SEQ_REF = 2.5e-3  # the reference keeps its logits in fp16 (token_attention.py:312-314): 1.3e-3 from truth by itself


def _seq_metadata(tree, device="cuda"):
    leaves = sorted(tree.leaves.values(), key=lambda n: n.id)
    lens = [len(tree.leaf_path_slots(lf)) for lf in leaves]
    positions = torch.tensor(lens, dtype=torch.int64, device=device) - 1
    return deft_amd.InputMetadata.from_tree(tree, tree.req_to_token_pool, tree.token_to_kv_pool,
                                            deft_amd.ForwardMode.DECODE, positions, None), lens


@pytest.mark.parametrize("name,geom", [(n, g) for n in sorted(SEQ_GOLDEN) for g in SEQ_GOLDEN[n]])
def test_sequential_attention_matches_reference_and_truth(name, geom):
    import os
    Hq, Hkv, D = geom
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "seq_" + name + ".npz"))
    tree = product_tree(name, device="cuda", heads=(Hkv, D))
    meta, lens = _seq_metadata(tree)
    assert lens == g["b_seq_len"].tolist() and meta.start_loc.tolist() == g["b_start_loc"].tolist()
    table = tree.req_to_token_pool.req_to_token
    for i, r in enumerate(meta.req_pool_indices.tolist()):  # the product page table = the reference's
        assert table[r, : lens[i]].tolist() == g["req_rows"][i][: lens[i]].tolist()
    q_np, kv_np = seeded_inputs(name, geom, len(lens))
    tree.token_to_kv_pool.kv_data[0].copy_(torch.from_numpy(kv_np))
    pool = tree.token_to_kv_pool
    q = torch.from_numpy(q_np).cuda()
    o = torch.full((len(lens), Hq, D), float("nan"), dtype=torch.float16, device="cuda")
    deft_amd.token_attention_fwd(q, pool.get_key_buffer(0), pool.get_value_buffer(0), o, table, meta.req_pool_indices,
                                 meta.start_loc, meta.seq_lens, meta.max_seq_len, meta.other_kv_index,
                                 meta.total_num_tokens)
    torch.cuda.synchronize()
    out = o.cpu().numpy()
    assert np.isfinite(out.astype(np.float32)).all()
    assert max_abs(out, g["o_seq_%d_%d_%d" % geom]) < TOL_SEQ_REF
    otree = oracle_tree(name)
    assert max_abs(out, oa.sequential_truth(q_np, kv_np, leaf_paths(otree))) < TOL_EXACT
    # and it is the same function of the inputs as the tree operators
    tree_out = _run(name, geom, "flatten")[0]
    assert max_abs(out, tree_out) < TOL_EXACT


def test_sequential_module_path_with_fused_append():
    """DeFTAttention.forward in ForwardMode.DECODE: store_kv_cache + token attention (deft_attention.py:153-188)."""
    name, geom = "multilevel", (8, 2, 128)
    Hq, Hkv, D = geom
    tree = product_tree(name, device="cuda", heads=(Hkv, D))
    q_np, kv_np = seeded_inputs(name, geom, len(tree.leaves))
    pool = tree.token_to_kv_pool
    pool.kv_data[0].copy_(torch.from_numpy(kv_np))
    leaves = sorted(tree.leaves.values(), key=lambda n: n.id)
    loc = torch.tensor([lf.kv_indices[-1] for lf in leaves], dtype=torch.int32, device="cuda")
    k_new = pool.kv_data[0][loc.long(), 0].clone()
    v_new = pool.kv_data[0][loc.long(), 1].clone()
    pool.kv_data[0][loc.long()] = 0  # the step's own rows are not in the pool yet
    updater = deft_amd.KVCacheUpdater(True, pool, loc, None, False)
    meta, lens = _seq_metadata(tree)
    meta.kv_updater = updater
    attn = deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, 0)
    q = torch.from_numpy(q_np).cuda().view(len(lens), Hq * D)
    out = attn(q, k_new.view(len(lens), -1), v_new.view(len(lens), -1), meta)
    torch.cuda.synchronize()
    assert torch.equal(pool.kv_data[0].cpu(), torch.from_numpy(kv_np))  # rows appended
    truth = oa.sequential_truth(q_np, kv_np, leaf_paths(oracle_tree(name)))
    assert max_abs(out.view(len(lens), Hq, D).cpu().numpy(), truth) < TOL_EXACT


@pytest.mark.parametrize("mode", ["flatten", "node"])
@pytest.mark.parametrize("shape", [(4, 2, 100_000, 48), (2, 1, 450_000, 3)])
def test_many_partial_rows_take_several_merge_passes(mode, shape):
    """A 100k-token prefix under 48 branches leaves ~37k (Flatten) / ~25k (Node) partial rows: more than one pass
    of the merge kernel's LDS row list (15872 rows per pass); a 450k-token prefix is 3516 Flatten blocks, past the
    unit kernel's 64 KB LDS tables (it takes the CU's whole LDS then).  Checked against fp64 attention per leaf."""
    from deft_amd.memory_pool import ReqToTokenPool, TokenToKVPool
    from deft_amd.tree_cache import TreeCache

    Hq, Hkv, prefix, width = shape
    D = 128
    size = prefix + 4 * width + 256
    req = ReqToTokenPool(width + 8, size + 8, device="cuda")
    pool = TokenToKVPool(size, torch.float16, Hkv, D, 1, device="cuda")
    tree = TreeCache(torch.float16, Hkv, D, 1, req, pool, None, True, False)
    tree.init_prompt(torch.arange(1, prefix + 1, dtype=torch.int32))
    tree.branch(tree.root, width)
    for _ in range(3):
        for leaf in list(tree.leaves.values()):
            leaf.append_token(7)
        tree.alloc()
    md = deft_amd.TreeMetadata.from_tree_cache(tree)
    g = torch.Generator(device="cuda").manual_seed(5)
    pool._storage.normal_(generator=g)
    q = torch.randn((width, Hq, D), dtype=torch.float16, device="cuda", generator=g)
    kb, vb = pool.get_key_buffer(0), pool.get_value_buffer(0)
    o = torch.full_like(q, float("nan"))
    if mode == "flatten":
        deft_amd.tree_attention_subtree_fwd(q, kb, vb, o, *_flatten_args(md))
    else:
        deft_amd.tree_attention_fwd(q, kb, vb, o, md.node_kv, md.node_kv_offset, md.node_kv_len, md.node_q,
                                    md.node_q_offset, md.node_q_len)
    torch.cuda.synchronize()
    leaves = sorted(tree.leaves.values(), key=lambda n: n.id)
    for r in sorted({0, width // 3, width - 1}):
        slots = torch.tensor(tree.leaf_path_slots(leaves[r]), device="cuda")
        k = kb[slots].double().repeat_interleave(Hq // Hkv, dim=1).transpose(0, 1)
        v = vb[slots].double().repeat_interleave(Hq // Hkv, dim=1).transpose(0, 1)
        s = torch.einsum("hd,hsd->hs", q[r].double(), k) / D ** 0.5
        ref = torch.einsum("hs,hsd->hd", torch.softmax(s, dim=-1), v)
        assert (o[r].double() - ref).abs().max().item() < TOL_EXACT


@pytest.mark.parametrize("mode", ["flatten", "node"])
@pytest.mark.parametrize("shape", [(4, 4, 70), (8, 2, 100), (4, 4, 33), (12, 4, 40), (6, 6, 17), (10, 2, 9), (24, 8, 35), (7, 1, 5)])
def test_nodes_with_more_than_32_queries_fold_as_interleaved_runs(mode, shape):
    """A node shared by 33 / 70 / 100 leaves is cut by the reference's builder into alternating blocks of at most 32
    queries (period 2, 3, 4): the plan folds each query chunk's blocks as one run.  Against fp64 attention per leaf.
    The later shapes: head counts and group sizes that are not powers of two (3, 5 and 7 query heads per KV head, six MHA
    heads) and query counts that do not fill the merge's four-query workgroups -- its head-to-workgroup permutation
    (a head is merged on the XCD that wrote its rows) must stay a bijection for all of them."""
    from deft_amd.memory_pool import ReqToTokenPool, TokenToKVPool
    from deft_amd.tree_cache import TreeCache

    Hq, Hkv, width = shape
    D, prefix = 128, 1500
    size = prefix + 4 * width + 256
    req = ReqToTokenPool(width + 8, size + 8, device="cuda")
    pool = TokenToKVPool(size, torch.float16, Hkv, D, 1, device="cuda")
    tree = TreeCache(torch.float16, Hkv, D, 1, req, pool, None, True, False)
    tree.init_prompt(torch.arange(1, prefix + 1, dtype=torch.int32))
    tree.branch(tree.root, width)
    for _ in range(3):
        for leaf in list(tree.leaves.values()):
            leaf.append_token(7)
        tree.alloc()
    md = deft_amd.TreeMetadata.from_tree_cache(tree)
    g = torch.Generator(device="cuda").manual_seed(9)
    pool._storage.normal_(generator=g)
    q = torch.randn((width, Hq, D), dtype=torch.float16, device="cuda", generator=g)
    kb, vb = pool.get_key_buffer(0), pool.get_value_buffer(0)
    o = torch.full_like(q, float("nan"))
    if mode == "flatten":
        deft_amd.tree_attention_subtree_fwd(q, kb, vb, o, *_flatten_args(md))
    else:
        deft_amd.tree_attention_fwd(q, kb, vb, o, md.node_kv, md.node_kv_offset, md.node_kv_len, md.node_q,
                                    md.node_q_offset, md.node_q_len)
    torch.cuda.synchronize()
    leaves = sorted(tree.leaves.values(), key=lambda n: n.id)
    for r in range(width):
        slots = torch.tensor(tree.leaf_path_slots(leaves[r]), device="cuda")
        k = kb[slots].double().repeat_interleave(Hq // Hkv, dim=1).transpose(0, 1)
        v = vb[slots].double().repeat_interleave(Hq // Hkv, dim=1).transpose(0, 1)
        s = torch.einsum("hd,hsd->hs", q[r].double(), k) / D ** 0.5
        ref = torch.einsum("hs,hsd->hd", torch.softmax(s, dim=-1), v)
        assert (o[r].double() - ref).abs().max().item() < TOL_EXACT, r


@pytest.mark.parametrize("shape", [(32, 32, 1024, 32, 200), (8, 2, 1500, 70, 3), (4, 4, 300, 5, 40), (4, 4, 5, 40, 1),
                                   (32, 8, 8192, 8, 40)])
def test_plan_build_forms_give_identical_plans(shape):
    """The plan kernels' three forms (tests/exp_plan_forms.py) produce identical plan bytes and output bits.  The hook that
    forces the fallback forms exists in the EXPERIMENTS build only (`deft_debug_plan_form`, libdeft_amd_exp.so: the shipped
    library exports what include/deft_amd.h declares and nothing else), so the check runs in a child process bound to it."""
    import subprocess
    import sys

    exp = os.path.join(ROOT, "deft_amd", "lib", "libdeft_amd_exp.so")
    assert os.path.exists(exp), "build the experiments library: make -C deft_amd/csrc exp (__graft_entry__.build() does)"
    env = dict(os.environ, DEFT_AMD_LIB=exp, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "exp_plan_forms.py"), *[str(x) for x in shape]], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "forms identical" in r.stdout
    if shape == (32, 8, 8192, 8, 40):
        # one 8192-token prefix under 8 branches, Llama-3-8B heads: with 4-tile chunks the launch would leave CUs empty (16 + 3
        # leaders x 8 KV heads = 152 workgroups), so the 64-tile run is cut into 22 chunks of 3 (round 4's rule); the leaves'
        # 320 tokens share 3 blocks: 25 leaders, in the serial and the parallel form of the rule alike
        assert "flatten units 67 leaders 25" in r.stdout, r.stdout[-500:]
        # 8 queries x 4 heads of a group = one 32-row pass over every tile: every chunk reads its rows non-temporally
        assert "flatten temporal leaders 0\n" in r.stdout and "node temporal leaders 0\n" in r.stdout, r.stdout[-500:]
    if shape == (8, 2, 1500, 70, 3):
        # 70 queries x 4 heads of a group over the 1500-token prefix = 4 + 4 + 1 passes (three query chunks of <= 32) over each of
        # its tiles: more than five, so the chunks over the prefix ask for their rows with the temporal policy (desc[6]) and the
        # later passes find them in L2; the leaves' tiles (one pass) do not.  Flatten: 12 prefix tiles x 9 passes in 4-tile chunks.
        import re
        fl = int(re.search(r"flatten temporal leaders (\d+)", r.stdout).group(1))
        nd = int(re.search(r"node temporal leaders (\d+)", r.stdout).group(1))
        leaders = int(re.search(r"flatten units \d+ leaders (\d+)", r.stdout).group(1))
        assert 0 < fl < leaders and nd > 0, r.stdout[-500:]


def test_decode_step_inside_inference_mode():
    """The reference decorates its operators with @torch.inference_mode() and runners wrap the whole loop in it: tensors
    made there keep no version counter (`t._version` raises).  alloc() + from_tree_cache + every layer's forward, Flatten and
    Node, inside inference mode, against fp64 truth."""
    name, geom = "multilevel", (8, 2, 128)
    Hq, Hkv, D = geom
    otree = oracle_tree(name)
    with torch.inference_mode():
        tree = product_tree(name, device="cuda", heads=(Hkv, D), layers=2)
        q_np, kv_np = seeded_inputs(name, geom, len(tree.leaves))
        for leaf in list(tree.leaves.values()):
            leaf.append_token(7)
        for leaf in list(otree.leaves.values()):
            leaf.append_token(7)
        upd = tree.alloc()
        otree.alloc()
        md = deft_amd.TreeMetadata.from_tree_cache(tree)
        deft_amd.register_tree_metadata(md)
        pool = tree.token_to_kv_pool
        for l in range(2):
            pool.kv_data[l].copy_(torch.from_numpy(kv_np))
        nq = md.query_num
        q = torch.from_numpy(q_np).cuda().view(nq, Hq * D)
        k_new = torch.from_numpy(dyadic_rows(nq, Hkv * D, 3)).cuda()
        v_new = torch.from_numpy(dyadic_rows(nq, Hkv * D, 4)).cuda()
        outs = {}
        for mode in ("flatten", "node"):
            meta = deft_amd.InputMetadata(deft_amd.forward_mode_from_cli(mode), upd, pool)
            attn = [deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, l) for l in range(2)]
            outs[mode] = [attn[l](q, k_new, v_new, meta) for l in range(2)]
        torch.cuda.synchronize()
    kv_ref = kv_np.copy()
    loc = upd.cache_loc.cpu().numpy()
    kv_ref[loc, 0] = k_new.cpu().numpy().reshape(nq, Hkv, D)
    kv_ref[loc, 1] = v_new.cpu().numpy().reshape(nq, Hkv, D)
    truth = oa.sequential_truth(q_np, kv_ref, leaf_paths(otree))
    for mode, os_ in outs.items():
        for o in os_:
            assert max_abs(o.view(nq, Hq, D).cpu().numpy(), truth) < TOL_EXACT, mode


def dyadic_rows(n, width, seed):
    from deft_amd.utils.synthetic import dyadic_normal

    return dyadic_normal((n, width), seed)


@pytest.mark.parametrize("mode", ["flatten", "node"])
def test_cfg5_forest_at_full_size_matches_the_single_tree_golden(mode, golden):
    """BASELINE configs[4], one GPU's share at FULL size: eight 8192-token-prefix trees (8 branches x 64 tokens, Llama-3-8B
    geometry 32/8/128) in ONE pool, attended by ONE operator call over the concatenated metadata (deft_amd.Forest).  Every
    tree holds the KV of the single-tree golden scenario, so each tree's eight output rows must equal the reference's
    output for that scenario (tests/golden/forest_tree_8kx8.npz, generated by the reference itself)."""
    name, geom, n_trees = "forest_tree_8kx8", (32, 8, 128), 8
    Hq, Hkv, D = geom
    sc = SCENARIOS[name]
    per_tree = 8192 + 8 * 64
    req = deft_amd.ReqToTokenPool(n_trees * 16, per_tree + 16, device="cuda")
    pool = deft_amd.TokenToKVPool(n_trees * per_tree + 64, torch.float16, Hkv, D, 1, device="cuda")
    trees = []
    for t in range(n_trees):
        tree = deft_amd.TreeCache(torch.float16, Hkv, D, 1, req, pool, None, True, False)
        sc.script(tree, lambda n: torch.arange(1, n + 1, dtype=torch.int32))
        # built one after the other on a first-free allocator: tree t occupies the single tree's slots shifted by t * per_tree
        assert sorted(s for nd in tree.nodes.values() for s in nd.kv_indices) == list(range(t * per_tree, (t + 1) * per_tree))
        trees.append(tree)
    forest = deft_amd.Forest(trees)
    md = forest.metadata()
    q_np, kv_np = seeded_inputs(name, geom, 8)
    kv = torch.from_numpy(kv_np[:per_tree]).cuda()
    for t in range(n_trees):
        pool.kv_data[0][t * per_tree: (t + 1) * per_tree].copy_(kv)
    q = torch.from_numpy(np.concatenate([q_np] * n_trees)).cuda()
    o = torch.full((md.query_num, Hq, D), float("nan"), dtype=torch.float16, device="cuda")
    if mode == "flatten":
        deft_amd.tree_attention_subtree_fwd(q, pool.get_key_buffer(0), pool.get_value_buffer(0), o, md.block_len, md.block_q,
                                            md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv, md.block_lens)
    else:
        deft_amd.tree_attention_fwd(q, pool.get_key_buffer(0), pool.get_value_buffer(0), o, md.node_kv, md.node_kv_offset,
                                    md.node_kv_len, md.node_q, md.node_q_offset, md.node_q_len)
    torch.cuda.synchronize()
    out = o.cpu().numpy()
    ref = golden(name)["o_%s_%d_%d_%d" % ((mode,) + geom)]
    assert md.query_num == 8 * n_trees and md.total_kv_len == n_trees * per_tree
    for t in range(n_trees):
        assert max_abs(out[8 * t: 8 * t + 8], ref) < TOL, t
    if mode == "flatten":  # the same tree eight times: the same blocks, chunks and fp32 order per tree, so the same bits
        for t in range(1, n_trees):  # (Node mode packs small entries ACROSS neighbouring trees: same values, other order)
            assert np.array_equal(out[8 * t: 8 * t + 8], out[:8]), t


@pytest.mark.parametrize("mode", ["flatten", "node"])
@pytest.mark.parametrize("geom", [(8, 2, 64), (8, 4, 64), (16, 16, 64), (12, 6, 64), (6, 3, 64)])
@pytest.mark.parametrize("name", ["multilevel", "wide40", "spec_mock"])
def test_head_dim_64_head_pairs_against_truth(name, geom, mode):
    """head_dim 64 on the tile-parallel kernel (two adjacent KV heads per 256-byte pool row, stage1_np.h HD2) with GQA, with MHA,
    with a head-pair count that is not a power of two -- and with an ODD number of KV heads, which keeps the tile-per-workgroup
    kernel: the module path (paged append fused in) against fp64 per-leaf attention, and against the separate append + operator."""
    from deft_amd.tree_attention import flatten_append_attention, node_append_attention
    from deft_amd.utils.synthetic import dyadic_normal

    Hq, Hkv, D = geom
    outs = []
    for fused in (True, False):
        tree = product_tree(name, device="cuda", heads=(Hkv, D))
        for leaf in list(tree.leaves.values()):
            leaf.append_token(9)
        updater = tree.alloc()
        md = deft_amd.TreeMetadata.from_tree_cache(tree)
        nq = md.query_num
        pool = tree.token_to_kv_pool
        kv = torch.from_numpy(dyadic_normal(tuple(pool.kv_data[0].shape), 21)).cuda()
        pool.kv_data[0].copy_(kv)
        q = torch.from_numpy(dyadic_normal((nq, Hq, D), 22)).cuda()
        k_new = torch.from_numpy(dyadic_normal((nq, Hkv, D), 23)).cuda()
        v_new = torch.from_numpy(dyadic_normal((nq, Hkv, D), 24)).cuda()
        o = torch.full((nq, Hq, D), float("nan"), dtype=torch.float16, device="cuda")
        node_args = (md.node_kv, md.node_kv_offset, md.node_kv_len, md.node_q, md.node_q_offset, md.node_q_len)
        if fused and mode == "flatten":
            flatten_append_attention(q, pool.kv_data[0], o, updater.cache_loc, k_new, v_new, *_flatten_args(md))
        elif fused:
            node_append_attention(q, pool.kv_data[0], o, updater.cache_loc, k_new, v_new, *node_args)
        else:
            deft_amd.kv_append(pool.kv_data[0], updater.cache_loc, k_new, v_new)
            if mode == "flatten":
                deft_amd.tree_attention_subtree_fwd(q, pool.get_key_buffer(0), pool.get_value_buffer(0), o, *_flatten_args(md))
            else:
                deft_amd.tree_attention_fwd(q, pool.get_key_buffer(0), pool.get_value_buffer(0), o, *node_args)
        torch.cuda.synchronize()
        outs.append(o)
        if fused:
            kvd = pool.kv_data[0].double()
            leaves = sorted(tree.leaves.values(), key=lambda n: n.id)
            for i, lf in enumerate(leaves):
                slots = torch.tensor(tree.leaf_path_slots(lf), device="cuda")
                kk = kvd[slots, 0].repeat_interleave(Hq // Hkv, dim=1).transpose(0, 1)
                vv = kvd[slots, 1].repeat_interleave(Hq // Hkv, dim=1).transpose(0, 1)
                s = torch.einsum("hd,hsd->hs", q[i].double(), kk) / D ** 0.5
                ref = torch.einsum("hs,hsd->hd", torch.softmax(s, dim=-1), vv)
                assert (o[i].double() - ref).abs().max().item() < TOL_EXACT, (i, geom, mode)
    assert torch.equal(outs[0], outs[1])


def test_c_program_through_the_c_abi(tmp_path):
    """tests/c_abi/decode_from_c.c: hipMalloc / hipMemcpy from C, a TreeMetadata written by hand, deft_flatten_decode_f16, and a
    double-precision restatement inside the program -- no Python and no torch anywhere between the caller and the kernels."""
    import subprocess

    from test_host_logic import _build_c_program

    exe = _build_c_program(str(tmp_path / "decode_from_c"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    assert "worst |err| vs double" in r.stdout
