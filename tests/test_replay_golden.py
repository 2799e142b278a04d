"""The decode LOOP pinned on the reference's own branch functions (VERDICT r4 "next round" item 1).

tests/golden/replay_*.npz hold what the REFERENCE's TreeCache held, step by step, while the reference's
`example_branch_Func1_SimpleTree` / `Func3_FromTreeTemplate` / `Func4_SpeculativeDecoding`
(DeFT/deft/tree_decoding/generation/branch_func_example.py:12-62, :293-372, :374-442) drove it in the order of
`tree_generate.py:92-236` (tools/gen_golden_replay.py, build container only).  Here `deft_amd.replay` runs the same
templates on the same scores and must reproduce, bit for bit and at EVERY step: the slots `alloc()` hands out, the live
leaves (by id, and in the order the branch functions walk them), every leaf's last token and position, the node count
and the twelve `TreeMetadata` arrays (a digest per step, the arrays themselves and the pool's reference counts at the
snapshot steps) -- on the CPU through the host builder, and on the GPU through `DecodeSession` (the device builder inside
the captured step).  A swapped sibling order or a leaf walked in another order fails these tests
(`test_a_wrong_walk_order_is_caught`)."""
import hashlib
import json
import os
import random

import numpy as np
import pytest
import torch

import deft_amd
from deft_amd import replay as rp
from deft_amd.utils.synthetic import permutation_scores

GOLD_DIR = os.path.join(os.path.dirname(__file__), "golden")
TEMPLATES = json.load(open(os.path.join(GOLD_DIR, "templates.json")))
ARRAYS = ("node_q", "node_kv", "node_q_len", "node_kv_len", "node_q_offset", "node_kv_offset",
          "block_q", "block_q_cnts", "block_q_offset", "block_bitmasks", "block_kv", "block_lens")
WORKLOADS = {  # name -> (task, how the template is rebuilt WITHOUT the reference)
    "simple_w6": "few_shot", "simple_4kx32": "few_shot", "docmergeToT": "reasoning", "sorting128ToT": "reasoning",
    "keywordToT": "reasoning", "set128ToT": "reasoning", "speculative64": "speculative_decoding",
    "speculative256": "speculative_decoding",  # 256 candidates: eight query chunks below the root
}


def _digest(arrays) -> np.uint64:
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(np.asarray(a, dtype=np.int64))
        h.update(np.int64(a.size).tobytes())
        h.update(a.tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


def _template(name, g):
    task = WORKLOADS[name]
    prompt_len, max_gen_len, pool_size, width, vocab = (int(x) for x in g["config"])
    if task == "few_shot":
        return rp.synthetic_few_shot_template(width)
    if task == "reasoning":
        tpl = rp.TreeTemplate.from_node_table(TEMPLATES["reasoning"][name]["data"])
        assert int(tpl.value[0]) == prompt_len
        return tpl
    if name != "speculative64":  # (the fitted list as recorded; the fit itself is checked on tree_size64 below)
        return rp.TreeTemplate.flat(int(g["tree_size"][0]), g["accept_lengths"].tolist())
    sd = TEMPLATES["speculative"]["tree_size64"]
    tpl = rp.TreeTemplate.flat(sd["Token_Tree_size"], sd["Accept_length_0"])
    rp.fit_accept_lengths(tpl, max_gen_len, random.Random(0))  # data_loader.py:200-235 under random.seed(0)
    assert tpl.accept_lengths == g["accept_lengths"].tolist()
    return tpl


class _Checker:
    """What the trace hook compares at every decode step."""

    def __init__(self, g, fields=ARRAYS, digest_key="digest"):
        self.g, self.fields, self.digest_key = g, fields, digest_key
        self.k = 0  # step index
        self.at = 0  # offset into the concatenated per-leaf arrays
        self.snaps = set(int(x) for x in g["snap_iters"])
        self.snaps_seen = 0

    def __call__(self, it, tree, cache_loc, md_arrays, refcounts):
        g, k = self.g, self.k
        assert k < len(g["iter"]), f"the replay runs longer than the reference's ({len(g['iter'])} steps)"
        assert it == int(g["iter"][k])
        nq = int(g["nq"][k])
        sl = slice(self.at, self.at + nq)
        leaves = sorted(tree.leaves.values(), key=lambda n: n.id)
        assert [lf.id for lf in leaves] == g["leaf_ids"][sl].tolist(), f"live leaves differ at iteration {it}"
        assert list(tree.leaves.keys()) == g["leaf_walk"][sl].tolist(), f"tree.leaves iterates in another order at iteration {it}"
        assert np.asarray(cache_loc).tolist() == g["cache_loc"][sl].tolist(), f"cache_loc differs at iteration {it}"
        assert [lf.token_ids[-1] for lf in leaves] == g["last_token"][sl].tolist(), f"tokens differ at iteration {it}"
        assert [lf.positions[-1] for lf in leaves] == g["last_pos"][sl].tolist(), f"positions differ at iteration {it}"
        assert len(tree.nodes) == int(g["node_cnt"][k])
        if it in self.snaps:  # (first, so that a mismatch names the array instead of a digest)
            for a in self.fields:
                np.testing.assert_array_equal(md_arrays[a], g[f"s{it}_{a}"], err_msg=f"{a} at iteration {it}")
            rc = np.asarray(refcounts).astype(np.int64)
            used = np.flatnonzero(rc)
            np.testing.assert_array_equal(used, g[f"s{it}_ref_slots"], err_msg=f"slots in use at iteration {it}")
            np.testing.assert_array_equal(rc[used], g[f"s{it}_ref_counts"], err_msg=f"reference counts at iteration {it}")
            self.snaps_seen += 1
        assert _digest(md_arrays[a] for a in self.fields) == g[self.digest_key][k], f"TreeMetadata differs at iteration {it}"
        self.k += 1
        self.at += nq

    def finish(self, tree, pool):
        g = self.g
        assert self.k == len(g["iter"]) and self.at == len(g["cache_loc"]) and self.snaps_seen == len(self.snaps)
        nodes, leaves, used, tokens, finished = (int(x) for x in g["end_state"])
        assert len(tree.all_finished_seqs) == finished  # (branches output by the branch function's last iteration)
        assert (len(tree.nodes), len(tree.leaves)) == (nodes, leaves)
        assert int((pool.mem_state != 0).sum()) == used
        assert tree.get_tree_token_number() == tokens


def _run_cpu(name, mutate=None):
    g = np.load(os.path.join(GOLD_DIR, f"replay_{name}.npz"))
    prompt_len, max_gen_len, pool_size, width, vocab = (int(x) for x in g["config"])
    tpl = _template(name, g)
    if mutate is not None:
        mutate(tpl)
    r = rp.TemplateReplay(1, 1, 8, layers=1, mode="flatten", device="cpu", attention=False)
    chk = _Checker(g)

    def hook(it, tree, cache_loc, md, sess):
        chk(it, tree, cache_loc.cpu().numpy(), {a: getattr(md, a).cpu().numpy() for a in ARRAYS},
            np.asarray(tree.token_to_kv_pool.mem_state))

    r.trace_hook = hook
    r.run(tpl, WORKLOADS[name], prompt_len, max_gen_len, max_tokens=pool_size, max_leaves=300,
          scores_fn=lambda it, rows: permutation_scores(it, rows, vocab))
    chk.finish(r.tree, r.pool)
    return chk


@pytest.mark.parametrize("name", sorted(WORKLOADS))
def test_replay_reproduces_the_reference_loop_step_for_step(name):
    chk = _run_cpu(name)
    assert chk.k >= 20


@pytest.mark.parametrize("name", ["simple_w6", "keywordToT", "speculative64"])
def test_reference_shaped_front_ends_drive_the_same_loop(name):
    """The same runs through `deft_amd.branch_func_example` (the reference's function names and signatures) and
    `Branch_Controller`, in a loop written the way `tree_generate.py:92-236` is -- `model` is a bare object with a `.tree` --
    instead of `TemplateReplay.run`: what a script ported from the reference would execute."""
    import types

    from deft_amd import branch_func_example as bf

    g = np.load(os.path.join(GOLD_DIR, f"replay_{name}.npz"))
    prompt_len, max_gen_len, pool_size, width, vocab = (int(x) for x in g["config"])
    tpl = _template(name, g)
    fn = {"few_shot": bf.example_branch_Func1_SimpleTree, "reasoning": bf.example_branch_Func3_FromTreeTemplate,
          "speculative_decoding": bf.example_branch_Func4_SpeculativeDecoding}[WORKLOADS[name]]
    req = deft_amd.ReqToTokenPool(308, pool_size + 8, device="cpu")
    pool = deft_amd.TokenToKVPool(pool_size, torch.float16, 1, 8, 0, device="cpu")
    tree = deft_amd.TreeCache(torch.float16, 1, 8, 1, req, pool, None, True, False)
    model = types.SimpleNamespace(tree=tree)
    ctl = bf.Branch_Controller(branching_function=fn)
    ctl.set_execution_graph(tree_templates=None if WORKLOADS[name] == "few_shot" else tpl)

    def branch(it, rows):
        return ctl.apply_branching(model=model, iter=it, max_gen_len=max_gen_len, width=width, depth=0,
                                   logits=permutation_scores(it, rows, vocab), execution_graph=ctl.tree_templates)

    chk = _Checker(g)
    tree.init_prompt(torch.arange(1, prompt_len + 1, dtype=torch.int32))
    stop = branch(0, 1)
    it = 1
    while not stop and it < max_gen_len:
        leaves = sorted(tree.leaves.values(), key=lambda x: x.id)
        if not leaves:
            break
        tree.leaf_to_q = {lf.id: i for i, lf in enumerate(leaves)}
        upd = tree.alloc()
        md = deft_amd.TreeMetadata.from_tree_cache(tree, device="cpu")
        chk(it, tree, upd.cache_loc.cpu().numpy(), {a: getattr(md, a).cpu().numpy() for a in ARRAYS}, np.asarray(pool.mem_state))
        stop = branch(it, len(leaves))
        it += 1
    chk.finish(tree, pool)


@pytest.mark.parametrize("name", ["simple_w6", "set128ToT", "speculative64"])
def test_tree_generate_drives_the_same_loop(name):
    """`deft_amd.tree_generate.tree_generate` (the reference's driver signature, tree_generate.py:20-32) with a stub in place of the
    Llama model: the stub's `forward_tree_decode` sees, at every step, the reference's slots, tokens, positions and metadata."""
    import types

    from deft_amd import branch_func_example as bf
    from deft_amd.tree_generate import tree_generate

    g = np.load(os.path.join(GOLD_DIR, f"replay_{name}.npz"))
    prompt_len, max_gen_len, pool_size, width, vocab = (int(x) for x in g["config"])
    tpl = _template(name, g)
    fn = {"few_shot": bf.example_branch_Func1_SimpleTree, "reasoning": bf.example_branch_Func3_FromTreeTemplate,
          "speculative_decoding": bf.example_branch_Func4_SpeculativeDecoding}[WORKLOADS[name]]
    req = deft_amd.ReqToTokenPool(308, pool_size + 8, device="cpu")
    pool = deft_amd.TokenToKVPool(pool_size, torch.float16, 1, 8, 0, device="cpu")
    tree = deft_amd.TreeCache(torch.float16, 1, 8, 1, req, pool, None, True, False)
    chk = _Checker(g)
    end = {}

    class StubModel:
        use_paged_memory, use_tree_index = True, False

        def __init__(self):
            self.tree, self.it = tree, 0

        def forward_prefill(self, input_ids, req_pool_indices, seq_lens, prefix_lens, position_ids_offsets, kv_updater, flag):
            assert kv_updater.cache_loc.tolist() == list(range(prompt_len)) and req_pool_indices.tolist() == [tree.leaf_to_req[0]]
            return torch.log(torch.from_numpy(permutation_scores(0, 1, vocab))), None

        def forward_tree_decode(self, forward_mode, token_ids, positions, kv_updater, flag, md):
            self.it += 1
            leaves = sorted(tree.leaves.values(), key=lambda n: n.id)
            assert token_ids.tolist() == [lf.token_ids[-1] for lf in leaves] and positions.tolist() == [lf.positions[-1] for lf in leaves]
            chk(self.it, tree, kv_updater.cache_loc.cpu().numpy(), {a: getattr(md, a).cpu().numpy() for a in ARRAYS},
                np.asarray(pool.mem_state))
            return (torch.log(torch.from_numpy(permutation_scores(self.it, len(leaves), vocab))),), 0.0

    real_free = tree.free

    def free_after_recording():
        end.update(nodes=len(tree.nodes), leaves=len(tree.leaves), used=int((np.asarray(pool.mem_state) != 0).sum()),
                   tokens=tree.get_tree_token_number(), finished=len(tree.all_finished_seqs))
        real_free()

    tree.free = free_after_recording
    seen = types.SimpleNamespace(updates=0, printed=None)
    perf = types.SimpleNamespace(update=lambda **kw: setattr(seen, "updates", seen.updates + 1),
                                 print_latency=lambda **kw: setattr(seen, "printed", kw))
    tree_generate(model=StubModel(), mode=deft_amd.ForwardMode.TREE_DECODE_FLATTEN, tokenizer=None,
                  prompt_ids=torch.arange(1, prompt_len + 1, dtype=torch.int32).reshape(1, -1), max_seq_len=prompt_len + max_gen_len,
                  width=width, depth=0, branch_controller=bf.Branch_Controller(branching_function=fn),
                  tree_template=None if WORKLOADS[name] == "few_shot" else tpl, perf_metrics=perf)
    assert chk.k == len(g["iter"]) and chk.snaps_seen == len(chk.snaps)
    nodes, leaves_n, used, tokens, finished = (int(x) for x in g["end_state"])
    assert (end["nodes"], end["leaves"], end["used"], end["tokens"], end["finished"]) == (nodes, leaves_n, used, tokens, finished)
    assert seen.updates == chk.k and seen.printed["prompt_len"] == prompt_len and seen.printed["generated_len"] == tokens - prompt_len
    assert len(tree.nodes) == 0 and tree.root is None  # tree_generate.py:274


def test_a_wrong_walk_order_is_caught():
    """The pin has teeth: with two siblings swapped in the template's child lists (the order in which a branch hands out
    node ids and the top-k tokens), or with the leaves walked in id order where the reference walks its dict, the replay no
    longer reproduces the reference."""
    def swap_children(tpl):
        it = sorted(tpl.branch_at)[1]
        node, kids = tpl.branch_at[it][0]
        tpl.branch_at[it][0] = (node, [kids[1], kids[0]] + list(kids[2:]))
        # the children's ids are what the template's later events name: make the swap visible by exchanging their events too
        for ev in tpl.prune_at.values():
            for j, n in enumerate(ev):
                ev[j] = kids[1] if n == kids[0] else kids[0] if n == kids[1] else n

    with pytest.raises(AssertionError):
        _run_cpu("docmergeToT", mutate=swap_children)

    real = rp.BRANCH_FUNCS["speculative_decoding"]

    def sorted_walk(tree, it, max_gen_len, logits, tpl):
        if it == 0:
            return real(tree, it, max_gen_len, logits, tpl)
        leaves = sorted(tree.leaves.values(), key=lambda n: -n.id)  # NOT the reference's order (branch_func_example.py:409)
        before = len(tree.root.kv_indices)
        if it == len(tpl.accept_lengths):
            return True
        for i in range(tpl.accept_lengths[it]):
            tree.merge_nodes(tree.root, leaves[i], pruneB_flag=False)
        tree.reset_nodes_KV(leaves, len(tree.root.kv_indices) - before)
        return False

    rp.BRANCH_FUNCS["speculative_decoding"] = sorted_walk
    try:
        with pytest.raises(AssertionError):
            _run_cpu("speculative64")
    finally:
        rp.BRANCH_FUNCS["speculative_decoding"] = real


# ---------------------------------------------------------------------------------------------------------------
# the same loops through DecodeSession: the device builder inside the (captured) step
# ---------------------------------------------------------------------------------------------------------------
def _session_arrays(sess, fields):
    """The arrays the session's step wrote, read back from the epoch's buffers with the lengths the host derives from node
    lengths alone (the way DecodeSession's own callers size their views)."""
    from deft_amd._lib import check, lib
    from deft_amd.tree_cache import _FIELDS, _lens_from_sizes, _ptr

    dt = sess.dt
    mq, bl, mbl = dt.cfg
    sizes = np.zeros(9, dtype=np.int64)
    check(lib.deft_tree_md_sizes(sess.tree._native, mq, bl, mbl, 0, _ptr(sizes)), "deft_tree_md_sizes")
    lens = _lens_from_sizes(sizes)
    torch.cuda.synchronize()
    out, off = {}, 0
    host = sess.md_out.cpu().numpy()  # (the session's own arrays; dt.out belongs to TreeMetadata.from_tree_cache's device builder)
    for k in _FIELDS:
        if k in fields:
            out[k] = host[off : off + lens[k]].copy()
        off += sess.md_caps[k]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["flatten", "node"])
@pytest.mark.parametrize("name", sorted(WORKLOADS))
def test_session_reproduces_the_reference_loop_step_for_step(name, mode):
    g = np.load(os.path.join(GOLD_DIR, f"replay_{name}.npz"))
    prompt_len, max_gen_len, pool_size, width, vocab = (int(x) for x in g["config"])
    tpl = _template(name, g)
    fields = ARRAYS[6:] if mode == "flatten" else ARRAYS[:6]
    chk = _Checker(g, fields, "digest_block" if mode == "flatten" else "digest_node")
    # (incremental=False: the twelve arrays are rebuilt on EVERY step, which is what this test reads; window plans rebuild them once
    #  per window -- their outputs are checked against fp64 attention by tests/test_fuzz_slices.py and tests/test_session.py)
    r = rp.TemplateReplay(2, 1, 128, layers=1, mode=mode, device="cuda", attention=True, session=True, incremental=False)
    assert r.session

    def hook(it, tree, cache_loc, md, sess):
        assert sess is not None
        arrays = _session_arrays(sess, fields)
        chk(it, tree, sess.cache_loc[: sess.nq].cpu().numpy(), arrays, np.asarray(tree.token_to_kv_pool.mem_state))

    r.trace_hook = hook
    r.run(tpl, WORKLOADS[name], prompt_len, max_gen_len, max_tokens=pool_size, max_leaves=300,
          scores_fn=lambda it, rows: permutation_scores(it, rows, vocab))
    chk.finish(r.tree, r.pool)
    assert r.graph_captures >= 1  # the steps really ran from captured graphs
