"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz,
made by tools/gen_golden.py from /root/reference under Triton's CPU interpreter).

  * tree state + metadata: bit-exact (integer work)
  * attention: within 1e-3 of the reference's fp16 output (north-star tolerance)
    and of fp64 sequential ground truth
"""
import os

import numpy as np
import pytest

from helpers import leaf_paths, max_abs, oracle_metadata, oracle_tree, seeded_inputs
from oracle import attention as oa
from oracle.metadata import ARRAY_FIELDS
from scenarios import FULL_GEOMETRY, GQA_GEOMETRY, SCENARIOS, SMALL_GEOMETRIES, small_d_cases

TOL = 1e-3  # BASELINE.json north_star: "within 1e-3 fp16 tolerance"


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_tree_state_matches_reference(name, golden):
    g = golden(name)
    tree = oracle_tree(name)
    ids = sorted(tree.nodes)
    assert ids == g["node_ids"].tolist()
    assert [len(tree.nodes[i].kv_indices) for i in ids] == g["node_kv_lens_by_id"].tolist()
    assert [s for i in ids for s in tree.nodes[i].kv_indices] == g["node_kv_by_id"].tolist()
    assert np.array_equal(tree.pool.mem_state, g["pool_refcounts"])
    assert [lf.id for lf in tree.leaf_order()] == g["leaf_ids"].tolist()


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_metadata_bit_exact(name, golden):
    g = golden(name)
    md = oracle_metadata(name)
    for k in ARRAY_FIELDS:
        assert md[k].dtype == np.int64
        assert np.array_equal(md[k], g[k]), k
    assert [md["query_num"], md["node_num"], md["total_kv_len"], md["block_len"]] == g["scalars"].tolist()


def test_appendix_d_worked_example(golden):
    """SURVEY.md Appendix D, verified by hand against the reference builder."""
    md = oracle_metadata("appendix_d")
    assert md["block_q"].tolist() == [0, 1, 2, 3, 0, 1, 2, 3, 2, 3, 1]
    assert md["block_kv"].tolist() == [0, 1, 2, 3, 0, 1, 2, 3, 4, 5, 8, 11, 4, 5, 8, 11, 6, 9, 13, 14, 7, 10, 12, -1]
    assert md["block_bitmasks"].tolist() == [3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 1, 1, 3, 0, 0, 0, 3, 3, 1, 2, 1, 1, 1, 0]
    assert md["node_kv_len"].tolist() == [5, 5, 3, 2, 1, 1, 3]


def _kernel_cases():
    for name, sc in SCENARIOS.items():
        if sc.kernels:
            for geom in SMALL_GEOMETRIES:
                yield name, geom
    for name, geom in GQA_GEOMETRY.items():
        yield name, geom
    yield from small_d_cases()


@pytest.mark.parametrize("name,geom", list(_kernel_cases()))
def test_attention_matches_reference_and_truth(name, geom, golden):
    g = golden(name)
    tag = "_%d_%d_%d" % geom
    tree = oracle_tree(name)
    md = oracle_metadata(name, tree)
    q, kv = seeded_inputs(name, geom, md["query_num"])
    truth = oa.sequential_truth(q, kv, leaf_paths(tree))
    for mode, fwd in (("flatten", oa.flatten_forward), ("node", oa.node_forward)):
        ref = g["o_%s%s" % (mode, tag)]
        exact = fwd(q, kv, md, merge="exact")
        quirk = fwd(q, kv, md, merge="reference")
        assert max_abs(exact, truth) < 5e-4, (mode, "exact vs truth")
        assert max_abs(exact, ref) < TOL, (mode, "exact vs reference")
        assert max_abs(quirk, ref) < TOL, (mode, "reference-merge vs reference")
        assert max_abs(ref, truth) < TOL, (mode, "reference vs truth")


def test_node_stage1_partials_match_reference(golden):
    g = golden("cfgA_256x2")
    geom = (4, 4, 128)
    md = oracle_metadata("cfgA_256x2")
    q, kv = seeded_inputs("cfgA_256x2", geom, md["query_num"])
    po, pl = oa.node_stage1(q, kv, md)
    assert max_abs(po, g["node_partial_o_4_4_128"]) < 2e-5
    assert max_abs(pl, g["node_partial_lse_4_4_128"]) < 2e-5


@pytest.mark.parametrize("name", list(FULL_GEOMETRY))
def test_full_geometry_goldens(name, golden):
    """Llama-2-7B geometry (Hq=Hkv=32, D=128) on the 1k x 32 and 4k x 32 trees."""
    g = golden(name)
    geom = FULL_GEOMETRY[name]
    tag = "_%d_%d_%d" % geom
    tree = oracle_tree(name)
    md = oracle_metadata(name, tree)
    q, kv = seeded_inputs(name, geom, md["query_num"])
    out = oa.flatten_forward(q, kv, md)
    assert max_abs(out, g["o_flatten" + tag]) < TOL
    assert max_abs(oa.node_forward(q, kv, md), g["o_node" + tag]) < TOL
    # spot-check four leaves against fp64 truth (full truth is 32 x 4k x 32 heads)
    paths = leaf_paths(tree)
    rows = [r for r in (0, 7, 19, 31) if r < len(paths)] + ([1] if len(paths) == 2 else [])
    truth = oa.sequential_truth(q[rows], kv, [paths[r] for r in rows])
    assert max_abs(out[rows], truth) < 5e-4


@pytest.mark.parametrize("name,geom", [("cfgA_256x2", (32, 32, 128)), ("multilevel", (8, 2, 128)), ("wide40", (8, 2, 128))])
def test_cpu_baseline_port_computes_sequential_attention(name, geom):
    """bench.py's `cpu_baseline` leg times oracle/cpu_baseline.py::sequential_attention_cpu (BASELINE configs[0]: "PyTorch SDPA
    sequential-attention on CPU"): what is timed must also be RIGHT -- its output against the fp64 per-leaf truth, in both
    dtypes the bench reports (fp32: SDPA's own rounding; fp16: the fp16 accumulation of the CPU kernels)."""
    import torch

    from oracle.cpu_baseline import sequential_attention_cpu

    tree = oracle_tree(name)
    md = oracle_metadata(name, tree)
    q, kv = seeded_inputs(name, geom, md["query_num"])
    paths = leaf_paths(tree)
    truth = oa.sequential_truth(q, kv, paths)
    tp = [torch.as_tensor(np.asarray(p), dtype=torch.int64) for p in paths]
    o32 = sequential_attention_cpu(torch.from_numpy(q).float(), torch.from_numpy(kv).float(), tp).numpy()
    assert max_abs(o32, truth) < 2e-5
    o16 = sequential_attention_cpu(torch.from_numpy(q), torch.from_numpy(kv), tp).float().numpy()
    assert max_abs(o16, truth) < 4e-3


def test_flatten_equals_node_equals_truth_property():
    """Size-independent property: both decompositions reproduce sequential attention."""
    tree = oracle_tree("after_cut")
    md = oracle_metadata("after_cut", tree)
    q, kv = seeded_inputs("after_cut", (8, 2, 128), md["query_num"])
    truth = oa.sequential_truth(q, kv, leaf_paths(tree))
    a = oa.flatten_forward(q, kv, md)
    b = oa.node_forward(q, kv, md)
    assert max_abs(a, truth) < 5e-4 and max_abs(b, truth) < 5e-4 and max_abs(a, b) < 5e-4


def test_reference_merge_quirk_documented():
    """One query over two 1-token nodes, V=1 and V=3 (truth 2.0): the reference's
    zero-initialised row max returns 2.0 at lse=-2 but 0.0 at lse=-72.
    The exact merge returns 2.0 in both cases."""
    for lse, want_quirk in ((-2.0, 2.0), (-72.0, 0.0)):
        po = np.array([[[1.0], [3.0]]], dtype=np.float32)
        pl = np.full((1, 2), lse, dtype=np.float32)
        rows = np.array([0, 0])
        with np.errstate(all="ignore"):
            quirk = oa.merge_reference(rows, po, pl, 1)
        exact = oa.merge_exact(rows, po, pl, 1)
        assert float(exact[0, 0, 0]) == 2.0
        got = float(quirk[0, 0, 0])
        assert (np.isnan(got) or got == want_quirk) if lse < -70 else got == want_quirk


SEQ_GOLDEN = {"cfgA_256x2": [(4, 4, 128), (8, 2, 128)], "multilevel": [(4, 4, 128), (8, 2, 128), (4, 4, 64)],
              "wide40": [(8, 2, 128)], "chain_300": [(4, 4, 128)], "medusa64_tree": [(8, 2, 128)]}


@pytest.mark.parametrize("name", sorted(SEQ_GOLDEN))
def test_sequential_comparator_oracle_is_bit_exact_on_reference_outputs(name):
    """tests/golden/seq_*.npz = the reference's token_attention_fwd run on its own page table (tools/gen_golden_seq.py)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "seq_" + name + ".npz"))
    otree = oracle_tree(name)
    paths = leaf_paths(otree)
    lens = g["b_seq_len"]
    assert [len(p) for p in paths] == lens.tolist()  # the oracle tree's page table = the reference's
    for i, p in enumerate(paths):
        assert g["req_rows"][i][: lens[i]].tolist() == list(p)
    assert g["b_start_loc"].tolist() == np.concatenate([[0], np.cumsum(lens)[:-1]]).tolist()
    for geom in SEQ_GOLDEN[name]:
        q, kv = seeded_inputs(name, geom, len(lens))
        o = oa.token_attention_forward(q, kv, g["req_rows"], lens)
        assert np.array_equal(o, g["o_seq_%d_%d_%d" % geom])
        assert max_abs(o, oa.sequential_truth(q, kv, paths)) < 2.5e-3  # the reference's fp16 logits
