"""A batch of independent trees as ONE tree object (`TreeCache.init_forest`: a root without tokens): host builder, device
metadata kernels, operators and the captured decode session all serve it unchanged; results equal per-leaf attention over
each leaf's own root-to-leaf slots (fp64), i.e. what separate trees would give."""
import numpy as np
import pytest
import torch

import deft_amd
import deft_amd.tree_cache as tc
from deft_amd.tree_cache import _FIELDS


def _forest(device, Hkv=2, D=128, layers=1, prompts=(300, 200, 77), width=3, size=4096):
    req = deft_amd.ReqToTokenPool(64, size, device=device)
    pool = deft_amd.TokenToKVPool(size, torch.float16, Hkv, D, layers, device=device)
    tree = deft_amd.TreeCache(torch.float16, Hkv, D, layers, req, pool, None, True, False)
    tree.init_forest([torch.arange(1, n + 1, dtype=torch.int32) for n in prompts])
    for kid in list(tree.leaves.values()):
        tree.branch(kid, width)
    return tree, pool


def _step(tree):
    for leaf in tree.leaves.values():
        leaf.append_token(7)
    return tree.alloc()


def test_forest_tree_host_side():
    """Slots, positions and page tables of a virtual-root tree; every query sees exactly its own tree's path."""
    tree, pool = _forest("cpu", layers=0)
    for _ in range(5):
        _step(tree)
    md = deft_amd.TreeMetadata.from_tree_cache(tree)
    order = sorted(tree.leaves.values(), key=lambda n: n.id)
    assert md.query_num == 9 and md.total_kv_len == 300 + 200 + 77 + 9 * 5
    assert [lf.positions[-1] for lf in order] == [304] * 3 + [204] * 3 + [81] * 3  # positions restart with every tree
    tab = tree.req_to_token_pool.req_to_token
    for lf in order:
        n = lf.positions[-1] + 1
        assert tab[tree.leaf_to_req[lf.id], :n].tolist() == tree.leaf_path_slots(lf)
    bq, cnt, off, masks, bkv, bl = [t.numpy() for t in (md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks,
                                                         md.block_kv, md.block_lens)]
    seen = {q: set() for q in range(md.query_num)}
    for b in range(len(cnt)):
        for k in range(bl[b]):
            for i in range(cnt[b]):
                if (masks[b * 128 + k] >> i) & 1:
                    seen[int(bq[off[b] + i])].add(int(bkv[b * 128 + k]))
    for q, lf in enumerate(order):
        assert seen[q] == set(tree.leaf_path_slots(lf))


@pytest.mark.gpu
def test_forest_tree_device_metadata_equals_host_builder(monkeypatch):
    tree, pool = _forest("cuda")
    for step in range(40):  # crosses block boundaries; the blocks straddle tree boundaries
        _step(tree)
        if step % 13 == 0:
            monkeypatch.setattr(tc, "DEVICE_METADATA", True)
            dev = deft_amd.TreeMetadata.from_tree_cache(tree)
            monkeypatch.setattr(tc, "DEVICE_METADATA", False)
            host = deft_amd.TreeMetadata.from_tree_cache(tree)
            for f in _FIELDS:
                assert torch.equal(getattr(dev, f).cpu(), getattr(host, f).cpu()), (step, f)


@pytest.mark.gpu
@pytest.mark.parametrize("incremental", [False, True])
@pytest.mark.parametrize("mode", ["flatten", "node"])
def test_forest_tree_session_and_truth(mode, incremental):
    """DecodeSession over the virtual-root tree == the eager path bit for bit, and both == fp64 attention of every leaf over
    its own path."""
    Hq, Hkv, D, layers = 8, 2, 128, 2
    g = torch.Generator(device="cuda").manual_seed(11)
    kv_init = torch.randn((layers, 4096, 2, Hkv, D), dtype=torch.float16, device="cuda", generator=g)
    (te, pe), (ts, ps) = [_forest("cuda", Hkv, D, layers) for _ in range(2)]
    pe._storage.copy_(kv_init)
    ps._storage.copy_(kv_init)
    nq = 9
    q = torch.randn((layers, nq, Hq * D), dtype=torch.float16, device="cuda", generator=g)
    k = torch.randn((layers, nq, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    v = torch.randn((layers, nq, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    sess = deft_amd.DecodeSession(ts, Hq, Hkv, D, layers, lambda l: (q[l], k[l], v[l]), mode=mode, incremental=incremental)
    attn = [deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, l) for l in range(layers)]
    fmode = deft_amd.forward_mode_from_cli(mode)
    for step in range(30):
        for tree in (te, ts):
            for leaf in tree.leaves.values():
                leaf.append_token(7)
        upd = te.alloc()
        md = deft_amd.TreeMetadata.from_tree_cache(te)
        deft_amd.register_tree_metadata(md)
        try:
            ref = [attn[l](q[l], k[l], v[l], deft_amd.InputMetadata(fmode, upd, pe)) for l in range(layers)]
        finally:
            deft_amd.unregister_tree_metadata()
        out = sess.step()
        torch.cuda.synchronize()
        for l in range(layers):
            if incremental:  # (a window-plan step: the same keys in another partition, tests/test_session.py::_agree)
                err = (out[l].float() - ref[l].float()).abs()
                assert bool((err <= 1e-3 + ref[l].float().abs() * 2.0 ** -10).all()), (step, l, float(err.max()))
            else:
                assert torch.equal(out[l], ref[l]), (step, l)
        assert torch.equal(pe._storage, ps._storage)
    assert sess.captures == (2 if incremental else 1)
    # fp64 truth, layer 0: every leaf over its own root-to-leaf slots
    order = sorted(te.leaves.values(), key=lambda n: n.id)
    kv = pe.kv_data[0].double()
    o = (out[0] if incremental else ref[0]).view(nq, Hq, D).double()  # (the session's own output where it is not the eager one bit for bit)
    G = Hq // Hkv
    for qi, lf in enumerate(order):
        slots = torch.tensor(te.leaf_path_slots(lf), device="cuda")
        K, V = kv[slots, 0], kv[slots, 1]  # [n, Hkv, D]
        for h in range(Hq):
            s = (K[:, h // G] @ q[0, qi].view(Hq, D)[h].double()) * D ** -0.5
            truth = torch.softmax(s, 0) @ V[:, h // G]
            assert (o[qi, h] - truth).abs().max() < 1e-3, (qi, h)


def test_reset_nodes_kv_equals_reset_node_kv_leaf_by_leaf():
    """The batched form (one native call, one refcount update) leaves tree, pool refcounts and positions exactly as the
    reference's leaf-by-leaf loop does (branch_func_example.py:430-436)."""
    states = []
    for batched in (False, True):
        tree, pool = _forest("cpu", layers=0, prompts=(40,), width=5)
        for _ in range(3):
            _step(tree)
        leaves = sorted(tree.leaves.values(), key=lambda n: n.id)
        if batched:
            tree.reset_nodes_KV(leaves, 2)
        else:
            for lf in leaves:
                tree.reset_node_KV(lf, 2)
        states.append((pool.mem_state.copy(), pool.alloc_ct, [list(lf.kv_indices) for lf in leaves],
                       [lf.positions for lf in leaves], [lf.position_offset for lf in leaves]))
    a, b = states
    assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]
    assert all(len(x) == 0 for x in b[2])


def test_an_empty_node_that_is_not_the_root_is_still_an_error():
    """Only a ROOT without tokens is skipped (the forest's virtual root); elsewhere the reference raises and so do we."""
    tree, pool = _forest("cpu", layers=0, prompts=(40, 30), width=2)
    with pytest.raises(deft_amd.DeftLibraryError):
        deft_amd.TreeMetadata.from_tree_cache(tree)  # the new leaves hold no token yet


def test_epoch_capacities_cover_every_growth():
    """The buffers of a structural epoch must hold the metadata of EVERY growth 0 .. SLACK + 4 of the leaves.  The block arrays
    are not monotone in the growth (block boundaries move over the nodes), so sizing for the longest tree alone under-allocates
    -- the multi-level trees below need more block_q entries at some intermediate growth than at the largest one (found by
    tools/fuzz_replay.py).  deft_tree_md_sizes_upto is the exact element-wise maximum (one pass over the blocks per growth);
    the buffers are sized by deft_tree_md_caps, O(nodes) upper bounds of it, which must cover it without being wasteful."""
    from deft_amd._lib import lib, check
    from deft_amd.tree_cache import _ptr

    worst = 0
    for prompt, widths, lens in ((129, (3, 4), (7, 40)), (300, (5, 2, 3), (3, 130, 2)), (1000, (6, 6), (200, 1)), (5, (2, 2, 2), (40, 40, 40))):
        req = deft_amd.ReqToTokenPool(256, 8192, device="cpu")
        pool = deft_amd.TokenToKVPool(16384, torch.float16, 1, 128, 0, device="cpu")
        tree = deft_amd.TreeCache(torch.float16, 1, 128, 0, req, pool, None, True, False)
        tree.init_prompt(torch.arange(prompt, dtype=torch.int32))
        for wd, ln in zip(widths, lens):
            for leaf in list(tree.leaves.values()):
                tree.branch(leaf, wd)
            for _ in range(ln):
                _step(tree)
        sizes = np.zeros(5, dtype=np.int64)
        check(lib.deft_tree_layout(tree._native, 256, _ptr(sizes)), "deft_tree_layout")
        upto, at_max = np.zeros(9, dtype=np.int64), np.zeros(9, dtype=np.int64)
        check(lib.deft_tree_md_sizes_upto(tree._native, 32, 128, -1, 260, _ptr(upto)), "upto")
        check(lib.deft_tree_md_sizes(tree._native, 32, 128, -1, 260, _ptr(at_max)), "sizes")
        caps = np.zeros(9, dtype=np.int64)
        check(lib.deft_tree_md_caps(tree._native, 32, 128, -1, 260, _ptr(caps)), "caps")
        assert (upto <= caps).all(), (prompt, upto, caps)
        assert (caps[[0, 1, 2, 3, 4, 8]] == upto[[0, 1, 2, 3, 4, 8]]).all()  # the monotone ones are exact
        assert caps[5] <= 2 * upto[5] + 8 and caps[6] <= 2 * upto[6] + 8, (prompt, upto, caps)
        for g in range(0, 261):
            cur = np.zeros(9, dtype=np.int64)
            check(lib.deft_tree_md_sizes(tree._native, 32, 128, -1, g, _ptr(cur)), "sizes")
            assert (cur <= upto).all(), (prompt, g, cur, upto)
            worst = max(worst, int(cur[6] - at_max[6]))
    assert worst > 0  # (at least one of these trees is a case the old sizing got wrong)


def test_epoch_capacity_bounds_on_random_trees():
    """deft_tree_md_caps >= deft_tree_md_sizes_upto element-wise on random multi-level trees and batches of trees as one tree object
    (several max_q_len / block_len / growth settings), and equal on the arrays that grow with the leaves."""
    import random
    from deft_amd._lib import lib, check
    from deft_amd.tree_cache import _ptr

    rng = random.Random(11)
    for _ in range(60):
        req = deft_amd.ReqToTokenPool(512, 8192, device="cpu")
        pool = deft_amd.TokenToKVPool(1 << 16, torch.float16, 1, 128, 0, device="cpu")
        tree = deft_amd.TreeCache(torch.float16, 1, 128, 0, req, pool, None, True, False)
        if rng.random() < 0.3:
            tree.init_forest([torch.arange(rng.choice([1, 5, 127, 128, 129, 300, 1000]), dtype=torch.int32) for _ in range(rng.randint(2, 4))])
        else:
            tree.init_prompt(torch.arange(rng.choice([1, 5, 127, 128, 129, 300, 1000, 3000]), dtype=torch.int32))
        for _lvl in range(rng.randint(0, 3)):
            for leaf in list(tree.leaves.values()):
                if len(tree.leaves) < 60 and rng.random() < 0.8:
                    tree.branch(leaf, rng.randint(1, 5))
            for _s in range(rng.choice([1, 2, 7, 40, 130])):
                _step(tree)
        if len(tree.leaves) > 2 and rng.random() < 0.5:
            tree.cut(rng.choice(list(tree.leaves.values())))
        sizes = np.zeros(5, dtype=np.int64)
        check(lib.deft_tree_layout(tree._native, 256, _ptr(sizes)), "deft_tree_layout")
        for mq, bl, mbl, grow in ((32, 128, -1, 260), (4, 16, 64, 37), (32, 128, 128, 0)):
            upto, caps = np.zeros(9, dtype=np.int64), np.zeros(9, dtype=np.int64)
            check(lib.deft_tree_md_sizes_upto(tree._native, mq, bl, mbl, grow, _ptr(upto)), "upto")
            check(lib.deft_tree_md_caps(tree._native, mq, bl, mbl, grow, _ptr(caps)), "caps")
            assert (upto <= caps).all(), (mq, bl, mbl, grow, upto, caps)
            assert (caps[[0, 1, 2, 3, 4, 8]] == upto[[0, 1, 2, 3, 4, 8]]).all()
