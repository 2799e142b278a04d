"""The captured decode loop (deft_amd.FlattenDecodeSession: one hipGraph per structural epoch of the tree) against the eager
path (tree.alloc + TreeMetadata.from_tree_cache + DeFTAttention), step for step, through branches and cuts."""
import numpy as np
import pytest
import torch

import deft_amd

pytestmark = pytest.mark.gpu


def _agree(out, ref, incremental, tag=None):
    """Legacy steps (incremental=False) are the eager path bit for bit.  A window-plan step (csrc/window.h) folds the SAME keys in
    another partition -- the step's tokens sit in overflow tiles -- so it differs in the order of fp32 additions: held to the
    operator's tolerance against the eager result: 1e-3, plus ONE fp16 step of the value (two correctly rounded fp16 results of
    nearly equal sums may land on neighbouring grid points; tools/fuzz_session.py found exactly that on an output of magnitude 4)."""
    if not incremental:
        assert torch.equal(out, ref), tag
        return
    err = (out.float() - ref.float()).abs()
    tol = 1e-3 + ref.float().abs() * 2.0 ** -10
    assert bool((err <= tol).all()), (tag, float(err.max()))


def _mk(Hkv, D, layers, prefix, width, size):
    req = deft_amd.ReqToTokenPool(64, size, device="cuda")
    pool = deft_amd.TokenToKVPool(size, torch.float16, Hkv, D, layers, device="cuda")
    tree = deft_amd.TreeCache(torch.float16, Hkv, D, layers, req, pool, None, True, False)
    tree.init_prompt(torch.arange(1, prefix + 1, dtype=torch.int32))
    tree.branch(tree.root, width)
    return tree, pool


@pytest.mark.parametrize("incremental", [False, True])
@pytest.mark.parametrize("mode", ["flatten", "node"])
@pytest.mark.parametrize("use_graph", [True, False])
def test_session_equals_eager_step_for_step(use_graph, mode, incremental):
    Hq, Hkv, D, layers, prefix, width = 8, 2, 128, 3, 700, 6
    g = torch.Generator(device="cuda").manual_seed(3)
    kv_init = torch.randn((layers, 4096, 2, Hkv, D), dtype=torch.float16, device="cuda", generator=g)
    trees = []
    for _ in range(2):  # the same tree twice: one decoded eagerly, one through the session
        tree, pool = _mk(Hkv, D, layers, prefix, width, 4096)
        pool._storage.copy_(kv_init)
        trees.append((tree, pool))
    (te, pe), (ts, ps) = trees
    cap = 16
    q = torch.randn((layers, cap, Hq * D), dtype=torch.float16, device="cuda", generator=g)
    k = torch.randn((layers, cap, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    v = torch.randn((layers, cap, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    nq_now = [width]
    sess = deft_amd.DecodeSession(ts, Hq, Hkv, D, layers, lambda l: (q[l, : nq_now[0]], k[l, : nq_now[0]], v[l, : nq_now[0]]),
                                  use_graph=use_graph, mode=mode, incremental=incremental)
    attn = [deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, l) for l in range(layers)]
    fmode = deft_amd.forward_mode_from_cli(mode)

    def both_steps(steps):
        for _ in range(steps):
            for tree in (te, ts):
                for leaf in tree.leaves.values():
                    leaf.append_token(7)
            upd = te.alloc()
            md = deft_amd.TreeMetadata.from_tree_cache(te)
            deft_amd.register_tree_metadata(md)
            meta = deft_amd.InputMetadata(fmode, upd, pe)
            n = md.query_num
            nq_now[0] = n
            ref = [attn[l](q[l, :n], k[l, :n], v[l, :n], meta) for l in range(layers)]
            out = sess.step()
            torch.cuda.synchronize()
            for l in range(layers):
                _agree(out[l][:n], ref[l], incremental, l)
            assert torch.equal(pe._storage, ps._storage)  # the fused append wrote the same rows
            ea = te.req_to_token_pool.req_to_token
            sa = ts.req_to_token_pool.req_to_token
            assert torch.equal(ea, sa)  # and the page tables agree

    both_steps(140)  # crosses 128-slot block boundaries many times inside one epoch (window plans: several windows)
    per_epoch = 2 if incremental else 1  # (window plans: a replan graph and a patch graph)
    assert sess.captures == (per_epoch if use_graph else 0)
    if incremental:  # most steps only patched the plan; the overflow (one 128-slot tile, 6 tokens a step) was re-planned every ~21 steps
        assert sess.step_kinds["patch"] >= 125 and 5 <= sess.step_kinds["replan"] <= 9 and sess.step_kinds["legacy"] == 0, sess.step_kinds
    for tree in (te, ts):  # structural change: cut two leaves, branch one
        lv = sorted(tree.leaves.values(), key=lambda n: n.id)
        tree.cut(lv[1])
        tree.cut(lv[4])
        tree.branch(lv[0], 3)
    both_steps(20)
    assert sess.captures == (2 * per_epoch if use_graph else 0)
    assert sess.device_errors() == 0  # (no kernel of any step flagged anything: room, capacities, journal, window tables)


@pytest.mark.parametrize("incremental", [False, True])
@pytest.mark.parametrize("mode", ["flatten", "node"])
def test_session_after_device_built_metadata_of_the_same_tree(mode, incremental):
    """The session's tree already has a current device copy when an epoch's first step arrives (a device-built
    `TreeMetadata.from_tree_cache(tree)` -- the default on GPU pools -- ran before it): nothing is uploaded then, so the step
    has to append its own slots to the device copy (it ran with advance=False and every leaf lost a slot)."""
    Hq, Hkv, D, layers, prefix, width = 8, 2, 128, 2, 300, 5
    g = torch.Generator(device="cuda").manual_seed(11)
    kv_init = torch.randn((layers, 2048, 2, Hkv, D), dtype=torch.float16, device="cuda", generator=g)
    (te, pe), (ts, ps) = [_mk(Hkv, D, layers, prefix, width, 2048) for _ in range(2)]
    for p in (pe, ps):
        p._storage.copy_(kv_init)
    cap = 16
    q = torch.randn((layers, cap, Hq * D), dtype=torch.float16, device="cuda", generator=g)
    k = torch.randn((layers, cap, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    v = torch.randn((layers, cap, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    nq_now = [width]
    sess = deft_amd.DecodeSession(ts, Hq, Hkv, D, layers, lambda l: (q[l, : nq_now[0]], k[l, : nq_now[0]], v[l, : nq_now[0]]), mode=mode,
                                  incremental=incremental)
    attn = [deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, l) for l in range(layers)]
    fmode = deft_amd.forward_mode_from_cli(mode)

    def both_steps(steps, peek_every=0):
        for s in range(steps):
            for tree in (te, ts):
                for leaf in tree.leaves.values():
                    leaf.append_token(7)
            upd = te.alloc()
            md = deft_amd.TreeMetadata.from_tree_cache(te)
            deft_amd.register_tree_metadata(md)
            n = md.query_num
            nq_now[0] = n
            ref = [attn[l](q[l, :n], k[l, :n], v[l, :n], deft_amd.InputMetadata(fmode, upd, pe)) for l in range(layers)]
            out = sess.step()
            if s > 0:  # (every step after an epoch's first writes its page-table entries in the step's first kernel, not by an index_put)
                assert sess.page_table_folded
            torch.cuda.synchronize()
            for l in range(layers):
                _agree(out[l][:n], ref[l], incremental, (s, l))
            assert torch.equal(pe._storage, ps._storage)
            if peek_every and s % peek_every == 0:  # a metadata build of the session's tree in the middle of an epoch
                m2 = deft_amd.TreeMetadata.from_tree_cache(ts)
                assert torch.equal(m2.block_lens, md.block_lens) and torch.equal(m2.node_kv, md.node_kv)

    for tree in (te, ts):  # one ordinary step first (a tree whose leaves hold no token yet cannot be built: the reference's error too)
        for leaf in tree.leaves.values():
            leaf.append_token(7)
        tree.alloc()
    deft_amd.TreeMetadata.from_tree_cache(ts)  # the device copy of ts is current BEFORE the session's first step
    both_steps(6, peek_every=2)
    for tree in (te, ts):
        lv = sorted(tree.leaves.values(), key=lambda n: n.id)
        tree.cut(lv[2])
        for leaf in tree.branch(lv[0], 2):
            leaf.append_token(9)
        for leaf in tree.leaves.values():
            if leaf.id != lv[0].id and len(leaf.kv_indices):
                leaf.append_token(9)
        tree.alloc()
    deft_amd.TreeMetadata.from_tree_cache(ts)  # and again between two epochs
    both_steps(5)
    # a second session on the same tree starts from a device copy the first one left current
    sess2 = deft_amd.DecodeSession(ts, Hq, Hkv, D, layers, lambda l: (q[l, : nq_now[0]], k[l, : nq_now[0]], v[l, : nq_now[0]]), mode=mode,
                                   incremental=incremental)
    sess = sess2
    both_steps(4)


@pytest.mark.parametrize("incremental", [False, True])
@pytest.mark.parametrize("mode", ["flatten", "node"])
def test_session_speculative_decoding_steps_replay_one_graph(mode, incremental):
    """The captured loop through speculative-decoding steps (merge accepted leaves into the root, reset every leaf: the journal
    of absorbed changes rides in the step's upload and is replayed by the step's first kernel): bit-identical to the eager path,
    and the graph is captured once per epoch -- not once per step."""
    Hq, Hkv, D, layers, prefix, width = 8, 2, 128, 2, 500, 40
    g = torch.Generator(device="cuda").manual_seed(17)
    kv_init = torch.randn((layers, 4096, 2, Hkv, D), dtype=torch.float16, device="cuda", generator=g)
    (te, pe), (ts, ps) = [_mk(Hkv, D, layers, prefix, width, 4096) for _ in range(2)]
    for p in (pe, ps):
        p._storage.copy_(kv_init)
    q = torch.randn((layers, width, Hq * D), dtype=torch.float16, device="cuda", generator=g)
    k = torch.randn((layers, width, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    v = torch.randn((layers, width, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    sess = deft_amd.DecodeSession(ts, Hq, Hkv, D, layers, lambda l: (q[l], k[l], v[l]), mode=mode, incremental=incremental)
    attn = [deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, l) for l in range(layers)]
    fmode = deft_amd.forward_mode_from_cli(mode)
    rng = np.random.default_rng(3)
    for step in range(40):
        for tree in (te, ts):
            for leaf in tree.leaves.values():
                leaf.append_token(7)
        upd = te.alloc()
        md = deft_amd.TreeMetadata.from_tree_cache(te)
        deft_amd.register_tree_metadata(md)
        ref = [attn[l](q[l], k[l], v[l], deft_amd.InputMetadata(fmode, upd, pe)) for l in range(layers)]
        out = sess.step()
        torch.cuda.synchronize()
        for l in range(layers):
            _agree(out[l], ref[l], incremental, (step, l))
        assert torch.equal(pe._storage, ps._storage)
        accept = int(rng.integers(1, 5))
        for tree in (te, ts):
            lv = sorted(tree.leaves.values(), key=lambda n: n.id)
            before = len(tree.root.kv_indices)
            for lf in lv[:accept]:
                tree.merge_nodes(tree.root, lf, pruneB_flag=False)
            tree.reset_nodes_KV(lv, len(tree.root.kv_indices) - before)
    # epochs: the first step; the first merge into a root without room.  Everything after replays the second epoch's graph(s).
    assert sess.captures <= (3 if incremental else 2), sess.captures
    if incremental:  # a leaf is reset and refilled every step: its overflow position is reused, the window lasts
        assert sess.step_kinds["patch"] >= 30 and sess.step_kinds["legacy"] == 0, sess.step_kinds
    assert sess.device_errors() == 0


def test_session_falls_back_when_a_window_plan_cannot_host_the_step():
    """More (query chunk, pass) regions than a window plan's tables hold -- 200 leaves under 16 query heads per KV head: 7 chunks x 16
    passes -- and the session runs the rebuild-every-step form instead (deft_window_supported = 0): bit-identical to the eager path."""
    Hq, Hkv, D, layers, prefix, width = 16, 1, 128, 1, 200, 200
    assert deft_amd.lib.deft_window_supported(width, 32, Hq, Hkv) == 0
    g = torch.Generator(device="cuda").manual_seed(9)
    kv_init = torch.randn((layers, 8192, 2, Hkv, D), dtype=torch.float16, device="cuda", generator=g)
    trees = []
    for _ in range(2):
        req = deft_amd.ReqToTokenPool(256, 8192, device="cuda")
        pool = deft_amd.TokenToKVPool(8192, torch.float16, Hkv, D, layers, device="cuda")
        tree = deft_amd.TreeCache(torch.float16, Hkv, D, layers, req, pool, None, True, False)
        tree.init_prompt(torch.arange(1, prefix + 1, dtype=torch.int32))
        tree.branch(tree.root, width)
        pool._storage.copy_(kv_init)
        trees.append((tree, pool))
    (te, pe), (ts, ps) = trees
    q = torch.randn((layers, width, Hq * D), dtype=torch.float16, device="cuda", generator=g)
    k = torch.randn((layers, width, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    v = torch.randn((layers, width, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    sess = deft_amd.DecodeSession(ts, Hq, Hkv, D, layers, lambda l: (q[l], k[l], v[l]))  # (incremental=True asked for)
    attn = [deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, l) for l in range(layers)]
    fmode = deft_amd.forward_mode_from_cli("flatten")
    for step in range(6):
        for tree in (te, ts):
            for leaf in tree.leaves.values():
                leaf.append_token(7)
        upd = te.alloc()
        md = deft_amd.TreeMetadata.from_tree_cache(te)
        deft_amd.register_tree_metadata(md)
        ref = attn[0](q[0], k[0], v[0], deft_amd.InputMetadata(fmode, upd, pe))
        out = sess.step()
        torch.cuda.synchronize()
        assert torch.equal(out[0], ref), step
    assert sess.step_kinds["legacy"] == 5 and sess.step_kinds["patch"] == 0 and sess.device_errors() == 0, sess.step_kinds


@pytest.mark.parametrize("mode", ["flatten", "node"])
def test_session_fetch_by_kernel_form(mode):
    """`staging="kernel"`: the step's words read from the pinned ring by the step's first kernel (csrc/window.h StageFetch) instead of
    copied in front of the step -- not the default (profiles/r6_staging_kernel_vs_copy.txt), kept correct: window plans and the
    rebuild-every-step form against the eager path, through an epoch change."""
    Hq, Hkv, D, layers, prefix, width = 8, 2, 128, 2, 300, 5
    for incremental in (True, False):
        g = torch.Generator(device="cuda").manual_seed(5)
        kv_init = torch.randn((layers, 2048, 2, Hkv, D), dtype=torch.float16, device="cuda", generator=g)
        (te, pe), (ts, ps) = [_mk(Hkv, D, layers, prefix, width, 2048) for _ in range(2)]
        for p in (pe, ps):
            p._storage.copy_(kv_init)
        cap = 16
        q = torch.randn((layers, cap, Hq * D), dtype=torch.float16, device="cuda", generator=g)
        k = torch.randn((layers, cap, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
        v = torch.randn((layers, cap, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
        nq_now = [width]
        sess = deft_amd.DecodeSession(ts, Hq, Hkv, D, layers, lambda l: (q[l, : nq_now[0]], k[l, : nq_now[0]], v[l, : nq_now[0]]), mode=mode,
                                      incremental=incremental, staging="kernel")
        attn = [deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, l) for l in range(layers)]
        fmode = deft_amd.forward_mode_from_cli(mode)
        for phase in range(2):
            for _ in range(30):
                for tree in (te, ts):
                    for leaf in tree.leaves.values():
                        leaf.append_token(7)
                upd = te.alloc()
                md = deft_amd.TreeMetadata.from_tree_cache(te)
                deft_amd.register_tree_metadata(md)
                n = md.query_num
                nq_now[0] = n
                ref = [attn[l](q[l, :n], k[l, :n], v[l, :n], deft_amd.InputMetadata(fmode, upd, pe)) for l in range(layers)]
                out = sess.step()
                torch.cuda.synchronize()
                for l in range(layers):
                    _agree(out[l][:n], ref[l], incremental, (incremental, phase, l))
                assert torch.equal(pe._storage, ps._storage)
            for tree in (te, ts):
                lv = sorted(tree.leaves.values(), key=lambda x: x.id)
                tree.cut(lv[1])
                tree.branch(lv[0], 2)
        assert sess.device_errors() == 0 and sess.staging == "kernel"


@pytest.mark.parametrize("capture_after", [1, 3, "auto"])
def test_session_head_dim_64_and_lazy_capture(capture_after):
    """head_dim 64 through the captured loop (two KV heads to a 256-byte pool row: the tile-parallel kernel's head pairs read a
    per-step plan, so one captured launch serves the epoch), and `capture_after`: the epoch's first steps run eagerly, the graph
    is captured once the epoch has lasted -- bit-identical to the eager operators either way."""
    Hq, Hkv, D, layers, prefix, width = 8, 4, 64, 2, 500, 5
    g = torch.Generator(device="cuda").manual_seed(23)
    kv_init = torch.randn((layers, 2048, 2, Hkv, D), dtype=torch.float16, device="cuda", generator=g)
    (te, pe), (ts, ps) = [_mk(Hkv, D, layers, prefix, width, 2048) for _ in range(2)]
    for p in (pe, ps):
        p._storage.copy_(kv_init)
    cap = 16
    q = torch.randn((layers, cap, Hq * D), dtype=torch.float16, device="cuda", generator=g)
    k = torch.randn((layers, cap, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    v = torch.randn((layers, cap, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    nq_now = [width]
    sess = deft_amd.DecodeSession(ts, Hq, Hkv, D, layers, lambda l: (q[l, : nq_now[0]], k[l, : nq_now[0]], v[l, : nq_now[0]]),
                                  capture_after=capture_after, incremental=False)
    attn = [deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, l) for l in range(layers)]
    fmode = deft_amd.forward_mode_from_cli("flatten")

    def both_steps(steps):
        for _ in range(steps):
            for tree in (te, ts):
                for leaf in tree.leaves.values():
                    leaf.append_token(7)
            upd = te.alloc()
            md = deft_amd.TreeMetadata.from_tree_cache(te)
            deft_amd.register_tree_metadata(md)
            n = md.query_num
            nq_now[0] = n
            ref = [attn[l](q[l, :n], k[l, :n], v[l, :n], deft_amd.InputMetadata(fmode, upd, pe)) for l in range(layers)]
            out = sess.step()
            torch.cuda.synchronize()
            for l in range(layers):
                assert torch.equal(out[l][:n], ref[l]), l
            assert torch.equal(pe._storage, ps._storage)

    both_steps(2)
    assert sess.captures == (1 if capture_after in (1, "auto") else 0)  # the second step of an epoch is captured at once, or not yet
    both_steps(8)
    assert sess.captures == 1
    for rounds in range(3):  # epochs of two steps each: `auto` stops capturing after the first short one
        for tree in (te, ts):
            lv = sorted(tree.leaves.values(), key=lambda n: n.id)
            tree.cut(lv[-1])
            tree.branch(lv[0], 2)
        both_steps(2)
    assert sess.captures == {1: 4, 3: 1, "auto": 2}[capture_after]
    both_steps(6)  # ... and the last epoch lasts: everyone captures it
    assert sess.captures == {1: 4, 3: 2, "auto": 3}[capture_after]


@pytest.mark.parametrize("incremental", [False, True])
@pytest.mark.parametrize("mode", ["flatten", "node"])
def test_second_device_copy_between_steps_cannot_starve_the_session_of_journalled_changes(mode, incremental):
    """a session holds its device copy and a captured graph; between two steps the tree absorbs a merge + reset (journalled,
    not yet taken) and somebody builds metadata of the SAME tree with ANOTHER max_q_len -- a second device copy, whose upload image
    carries the journalled changes and clears the journal.  The session's copy never saw them: the fetch has to end the epoch
    for it (it uploads again) instead of leaving it with stale node lengths and wrong attention."""
    Hq, Hkv, D, layers, prefix, width = 8, 2, 128, 2, 400, 12
    g = torch.Generator(device="cuda").manual_seed(23)
    kv_init = torch.randn((layers, 4096, 2, Hkv, D), dtype=torch.float16, device="cuda", generator=g)
    (te, pe), (ts, ps) = [_mk(Hkv, D, layers, prefix, width, 4096) for _ in range(2)]
    for p in (pe, ps):
        p._storage.copy_(kv_init)
    q = torch.randn((layers, width, Hq * D), dtype=torch.float16, device="cuda", generator=g)
    k = torch.randn((layers, width, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    v = torch.randn((layers, width, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    sess = deft_amd.DecodeSession(ts, Hq, Hkv, D, layers, lambda l: (q[l], k[l], v[l]), mode=mode, capture_after=1, incremental=incremental)
    attn = [deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, l) for l in range(layers)]
    fmode = deft_amd.forward_mode_from_cli(mode)

    def step_and_compare(tag):
        for tree in (te, ts):
            for leaf in tree.leaves.values():
                leaf.append_token(7)
        upd = te.alloc()
        md = deft_amd.TreeMetadata.from_tree_cache(te)
        deft_amd.register_tree_metadata(md)
        ref = [attn[l](q[l], k[l], v[l], deft_amd.InputMetadata(fmode, upd, pe)) for l in range(layers)]
        out = sess.step()
        torch.cuda.synchronize()
        for l in range(layers):
            _agree(out[l], ref[l], incremental, (tag, l))
        assert torch.equal(pe._storage, ps._storage)

    def speculative_update(accept):
        for tree in (te, ts):
            lv = sorted(tree.leaves.values(), key=lambda n: n.id)
            before = len(tree.root.kv_indices)
            for lf in lv[:accept]:
                tree.merge_nodes(tree.root, lf, pruneB_flag=False)
            tree.reset_nodes_KV(lv, len(tree.root.kv_indices) - before)

    for i in range(6):  # reach the epoch in which merges are absorbed and the step is captured
        step_and_compare(("warm", i))
        speculative_update(2)
    assert sess.graph is not None
    step_and_compare("every leaf holds its token's slot again")
    # between two steps the tree absorbs a merge (the root takes a leaf's slots; every node still holds KV, so metadata can be
    # built) -- journalled, not yet handed to anybody ...
    for tree in (te, ts):
        lv = sorted(tree.leaves.values(), key=lambda n: n.id)
        tree.merge_nodes(tree.root, lv[0], pruneB_flag=False)
    epoch_before = ts._epoch()
    # ... and the interloper builds metadata of the session's tree under another configuration: a second device copy
    other = deft_amd.TreeMetadata.from_tree_cache(ts, max_q_len=16, copy=True)
    assert other.query_num == width
    assert ts._epoch() != epoch_before, "a fetch that swallowed a pending journal must end the epoch for the other copies"
    step_and_compare("after the second copy")
    speculative_update(3)
    step_and_compare("and the step after")


@pytest.mark.parametrize("incremental", [False, True])
def test_session_node_chunk_equals_eager_step_for_step(incremental):
    """`--mode node_chunk` (BLOCK_CONFIG["MAX_BLOCK_LEN"] = 128: every node cut into 128-token entries, which the Node plan folds
    again -- round 5) through the captured session against the eager calls, bit for bit, across block boundaries and a branch."""
    Hq, Hkv, D, layers, prefix, width = 8, 2, 128, 2, 700, 5
    fmode = deft_amd.forward_mode_from_cli("node_chunk")
    try:
        assert deft_amd.BLOCK_CONFIG["MAX_BLOCK_LEN"] == 128
        g = torch.Generator(device="cuda").manual_seed(41)
        kv_init = torch.randn((layers, 4096, 2, Hkv, D), dtype=torch.float16, device="cuda", generator=g)
        (te, pe), (ts, ps) = [_mk(Hkv, D, layers, prefix, width, 4096) for _ in range(2)]
        for p in (pe, ps):
            p._storage.copy_(kv_init)
        cap = 16
        q = torch.randn((layers, cap, Hq * D), dtype=torch.float16, device="cuda", generator=g)
        k = torch.randn((layers, cap, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
        v = torch.randn((layers, cap, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
        nq_now = [width]
        sess = deft_amd.DecodeSession(ts, Hq, Hkv, D, layers, lambda l: (q[l, : nq_now[0]], k[l, : nq_now[0]], v[l, : nq_now[0]]), mode="node",
                                      incremental=incremental)
        attn = [deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, l) for l in range(layers)]

        def both_steps(steps):
            for _ in range(steps):
                for tree in (te, ts):
                    for leaf in tree.leaves.values():
                        leaf.append_token(7)
                upd = te.alloc()
                md = deft_amd.TreeMetadata.from_tree_cache(te)
                assert int(md.node_kv_len.max()) <= 128  # the entries really are chunks
                deft_amd.register_tree_metadata(md)
                n = md.query_num
                nq_now[0] = n
                ref = [attn[l](q[l, :n], k[l, :n], v[l, :n], deft_amd.InputMetadata(fmode, upd, pe)) for l in range(layers)]
                out = sess.step()
                torch.cuda.synchronize()
                for l in range(layers):
                    _agree(out[l][:n], ref[l], incremental, l)
                assert torch.equal(pe._storage, ps._storage)

        both_steps(140)  # the leaves cross their first 128-token boundary inside the epoch
        for tree in (te, ts):
            lv = sorted(tree.leaves.values(), key=lambda n: n.id)
            tree.branch(lv[0], 3)
        both_steps(10)
        assert sess.captures == (4 if incremental else 2)
    finally:
        deft_amd.BLOCK_CONFIG["MAX_BLOCK_LEN"] = -1


def test_sessions_hand_their_pinned_ring_on():
    """The staging ring (pinned host memory) and the capture stream are kept per process: a session made after another one of the same
    size was dropped takes over its ring -- no hipHostMalloc per session -- while the first session's last steps may still be in the
    queue (the ring waits behind an event), and decodes correctly from it."""
    import gc

    from deft_amd import session as S

    Hq, Hkv, D, layers, prefix, width = 8, 2, 128, 2, 300, 5
    g = torch.Generator(device="cuda").manual_seed(21)
    kv_init = torch.randn((layers, 4096, 2, Hkv, D), dtype=torch.float16, device="cuda", generator=g)
    q = torch.randn((layers, width, Hq * D), dtype=torch.float16, device="cuda", generator=g)
    k = torch.randn((layers, width, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    v = torch.randn((layers, width, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    attn = [deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, l) for l in range(layers)]
    fmode = deft_amd.forward_mode_from_cli("flatten")
    rings, sides = [], []
    for round_ in range(3):
        trees = []
        for _ in range(2):
            tree, pool = _mk(Hkv, D, layers, prefix, width, 4096)
            pool._storage.copy_(kv_init)
            trees.append((tree, pool))
        (te, pe), (ts, ps) = trees
        sess = deft_amd.DecodeSession(ts, Hq, Hkv, D, layers, lambda l: (q[l], k[l], v[l]))
        for step in range(12):
            for tree in (te, ts):
                for leaf in tree.leaves.values():
                    leaf.append_token(3)
            upd = te.alloc()
            md = deft_amd.TreeMetadata.from_tree_cache(te)
            deft_amd.register_tree_metadata(md)
            meta = deft_amd.InputMetadata(fmode, upd, pe)
            ref = [attn[l](q[l], k[l], v[l], meta) for l in range(layers)]
            out = sess.step()
            for l in range(layers):
                _agree(out[l], ref[l], True, (round_, step, l))
        assert sess.step_kinds["patch"] > 0 and sess.device_errors() == 0
        rings.append(sess._ring.data_ptr())
        sides.append(sess._side.cuda_stream)
        del sess  # (no synchronisation: its last steps may still be running)
        gc.collect()
        assert any(r.data_ptr() == rings[-1] for have in S._PINNED_RINGS.values() for r, _ in have)
    assert rings[0] == rings[1] == rings[2] and sides[0] == sides[1] == sides[2]
