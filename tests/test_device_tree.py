"""Device-side TreeMetadata (deft_amd/csrc/tree_plan.h) against the host builder deft_md_build -- which is pinned bit
for bit on the reference's own from_tree_cache outputs (tests/golden/*.npz, test_host_logic.py) -- and against those
goldens directly.  Integer work: everything here is bit-exact."""
import numpy as np
import pytest
import torch

import deft_amd
from deft_amd import tree_cache as tc
from product_helpers import MD_FIELDS, md_numpy, product_metadata, product_tree
from scenarios import SCENARIOS

pytestmark = pytest.mark.gpu


def _both(tree, **kw):
    dev = deft_amd.TreeMetadata.from_tree_cache(tree, device_build=True, **kw)
    host = deft_amd.TreeMetadata.from_tree_cache(tree, device_build=False, **kw)
    torch.cuda.synchronize()
    return dev, host


def _assert_same(dev, host):
    a, b = md_numpy(dev), md_numpy(host)
    for k in MD_FIELDS:
        assert a[k].shape == b[k].shape, k
        assert np.array_equal(a[k], b[k]), k
    assert (dev.query_num, dev.node_num, dev.total_kv_len, dev.leaf_to_q) == (host.query_num, host.node_num, host.total_kv_len,
                                                                              host.leaf_to_q)


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_device_metadata_equals_reference_golden(name, golden):
    """Every scripted tree (BASELINE's configurations among them), built on the GPU: the twelve arrays equal the
    reference's own `from_tree_cache` output."""
    sc = SCENARIOS[name]
    tree = product_tree(name, device="cuda")
    saved = dict(deft_amd.BLOCK_CONFIG)
    deft_amd.BLOCK_CONFIG["BLOCK_LEN"] = sc.block_len
    try:
        md = deft_amd.TreeMetadata.from_tree_cache(tree, max_q_len=sc.max_q_len, max_block_len=sc.max_block_len, device_build=True)
        torch.cuda.synchronize()
    finally:
        deft_amd.BLOCK_CONFIG.update(saved)
    g = golden(name)
    got = md_numpy(md)
    for k in MD_FIELDS:
        assert np.array_equal(got[k], g[k]), k
    assert [md.query_num, md.node_num, md.total_kv_len, md.block_len] == g["scalars"].tolist()
    dims = tree._device_tree.dims()
    assert dims[9] == 0 and dims[5] == len(g["block_lens"]) and dims[6] == len(g["block_q"])


def test_device_tree_advances_without_uploads():
    """A decode loop: after the first build the tree is uploaded ONCE per structural epoch; every alloc() advances the
    device copy by a kernel (cache_loc is the only thing that crosses PCIe) and the metadata stays equal to the host
    builder's at every step -- through 128-slot block boundaries and past the 32-query chunk limit."""
    Hkv, D = 2, 128
    req = deft_amd.ReqToTokenPool(64, 4096, device="cuda")
    pool = deft_amd.TokenToKVPool(4096, torch.float16, Hkv, D, 1, device="cuda")
    tree = deft_amd.TreeCache(torch.float16, Hkv, D, 1, req, pool, None, True, False)
    tree.init_prompt(torch.arange(1, 301, dtype=torch.int32))
    tree.branch(tree.root, 40)
    uploads = 0
    orig = tc._DeviceTree._upload

    def counting(self):
        nonlocal uploads
        uploads += 1
        return orig(self)

    tc._DeviceTree._upload = counting
    try:
        for step in range(70):
            for leaf in list(tree.leaves.values()):
                leaf.append_token(7)
            tree.alloc()
            dev, host = _both(tree)
            _assert_same(dev, host)
            assert tree._device_tree.dims()[9] == 0
        assert uploads == 1  # branch -> first build; 70 steps of growth absorbed by the layout's slack
        # structural change: cut three leaves (their slots return to the pool and come back LOWER than the survivors'
        # newest slots, so the device has to insert, not append) and branch one
        lv = sorted(tree.leaves.values(), key=lambda n: n.id)
        for leaf in (lv[3], lv[17], lv[39]):
            tree.cut(leaf)
        tree.branch(lv[5], 3)
        for step in range(12):
            for leaf in list(tree.leaves.values()):
                leaf.append_token(7)
            tree.alloc()
            dev, host = _both(tree)
            _assert_same(dev, host)
        assert uploads == 2
    finally:
        tc._DeviceTree._upload = orig


def test_leaf_outgrowing_its_room_triggers_a_new_layout():
    Hkv, D = 1, 128
    req = deft_amd.ReqToTokenPool(16, 2048, device="cuda")
    pool = deft_amd.TokenToKVPool(2048, torch.float16, Hkv, D, 1, device="cuda")
    tree = deft_amd.TreeCache(torch.float16, Hkv, D, 1, req, pool, None, True, False)
    tree.init_prompt(torch.arange(1, 51, dtype=torch.int32))
    tree.branch(tree.root, 3)
    saved = tc._DeviceTree.SLACK
    tc._DeviceTree.SLACK = 8
    try:
        epochs = set()
        for step in range(30):
            for leaf in list(tree.leaves.values()):
                leaf.append_token(7)
            tree.alloc()
            dev, host = _both(tree)
            _assert_same(dev, host)
            epochs.add(tree._device_tree.epoch)
        assert len(epochs) >= 3  # 30 steps with room for 8: re-laid out several times, never wrong
    finally:
        tc._DeviceTree.SLACK = saved


def test_speculative_decoding_mock_on_the_device_tree(golden):
    """merge_nodes / reset_node_KV every step (branch_func_example.py:374-442): each step is a structural change, the
    device copy is re-uploaded and the attention over the device-built metadata matches fp64 truth."""
    from helpers import leaf_paths, max_abs, oracle_tree, seeded_inputs
    from oracle import attention as oa

    name, geom = "spec_mock", (4, 4, 128)
    Hq, Hkv, D = geom
    tree = product_tree(name, device="cuda", heads=(Hkv, D))
    md = deft_amd.TreeMetadata.from_tree_cache(tree, device_build=True)
    q_np, kv_np = seeded_inputs(name, geom, md.query_num)
    tree.token_to_kv_pool.kv_data[0].copy_(torch.from_numpy(kv_np))
    pool = tree.token_to_kv_pool
    q = torch.from_numpy(q_np).cuda()
    o = torch.full((md.query_num, Hq, D), float("nan"), dtype=torch.float16, device="cuda")
    deft_amd.tree_attention_subtree_fwd(q, pool.get_key_buffer(0), pool.get_value_buffer(0), o, md.block_len, md.block_q,
                                        md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv, md.block_lens)
    torch.cuda.synchronize()
    assert max_abs(o.cpu().numpy(), golden(name)["o_flatten_4_4_128"]) < 1e-3
    assert max_abs(o.cpu().numpy(), oa.sequential_truth(q_np, kv_np, leaf_paths(oracle_tree(name)))) < 5e-4


def test_speculative_decoding_steps_stay_in_one_epoch():
    """The reference's speculative-decoding mock (branch_func_example.py:420-437) EVERY step: the accepted leaves' slots move into
    the root (merge_nodes), every leaf's KV is released (reset_node_KV), then alloc() gives every leaf a slot -- which the pool
    hands out LOWER than the root's newest slots, so the device copy has to merge the root's new slots in, not append them.
    After the first such step (the root is laid out with room from then on) no step uploads anything, and the device-built
    metadata equals the host builder's at every step."""
    Hkv, D = 2, 128
    req = deft_amd.ReqToTokenPool(128, 8192, device="cuda")
    pool = deft_amd.TokenToKVPool(8192, torch.float16, Hkv, D, 1, device="cuda")
    tree = deft_amd.TreeCache(torch.float16, Hkv, D, 1, req, pool, None, True, False)
    tree.init_prompt(torch.arange(1, 1017, dtype=torch.int32))
    leaves = tree.branch(tree.root, 64)
    uploads = 0
    orig = tc._DeviceTree._upload

    def counting(self):
        nonlocal uploads
        uploads += 1
        return orig(self)

    tc._DeviceTree._upload = counting
    rng = np.random.default_rng(5)
    try:
        for step in range(60):
            for leaf in leaves:
                leaf.append_token(7)
            tree.alloc()
            dev, host = _both(tree)
            _assert_same(dev, host)
            assert tree._device_tree.dims()[9] == 0
            accept = int(rng.integers(1, 5))
            before = len(tree.root.kv_indices)
            order = rng.permutation(64)[:accept] if step % 3 == 2 else range(accept)  # (any leaves: the merge is not only of the newest slots)
            for i in order:
                tree.merge_nodes(tree.root, leaves[int(i)], pruneB_flag=False)
            tree.reset_nodes_KV(leaves, len(tree.root.kv_indices) - before)
        assert uploads == 2  # the first build, and the first merge into a root that had no room yet
        assert len(tree.root.kv_indices) > 1016 + 60
    finally:
        tc._DeviceTree._upload = orig


def test_device_built_metadata_aliases_unless_copied():
    """Device-built TreeMetadata tensors are views of the tree's per-epoch buffer: the next build of the same
    tree overwrites them; `copy=True` returns tensors of their own, as the reference does."""
    Hkv, D = 1, 128
    req = deft_amd.ReqToTokenPool(16, 2048, device="cuda")
    pool = deft_amd.TokenToKVPool(2048, torch.float16, Hkv, D, 1, device="cuda")
    tree = deft_amd.TreeCache(torch.float16, Hkv, D, 1, req, pool, None, True, False)
    tree.init_prompt(torch.arange(1, 201, dtype=torch.int32))
    tree.branch(tree.root, 3)

    def step():
        for leaf in tree.leaves.values():
            leaf.append_token(7)
        tree.alloc()

    step()
    kept = deft_amd.TreeMetadata.from_tree_cache(tree, copy=True)
    view = deft_amd.TreeMetadata.from_tree_cache(tree)
    before = kept.block_kv.clone()
    assert torch.equal(view.block_kv, before) and view.block_kv.data_ptr() != kept.block_kv.data_ptr()
    step()
    nxt = deft_amd.TreeMetadata.from_tree_cache(tree)
    torch.cuda.synchronize()
    assert torch.equal(kept.block_kv, before)            # the copy kept its step
    assert nxt.block_kv.data_ptr() == view.block_kv.data_ptr()  # same buffer: `view` now shows the new step
    assert not torch.equal(view.block_kv[: before.shape[0]], before)


def test_journal_path_reproduces_the_reference_golden(golden):
    """The `spec_mock` script (two merge / reset iterations, tests/scenarios.py) with a device build after every alloc, so that the
    device copy exists and the second iteration's merge + resets reach it through the JOURNAL (the first finds the root without
    room and starts an epoch): the final device-built arrays equal the REFERENCE's own `from_tree_cache` output for that script."""
    sc = SCENARIOS["spec_mock"]
    req = deft_amd.ReqToTokenPool(128, sc.pool_size + 8, device="cuda")
    pool = deft_amd.TokenToKVPool(sc.pool_size, torch.float16, 1, 8, 1, device="cuda")
    tree = deft_amd.TreeCache(torch.float16, 1, 8, 1, req, pool, None, True, False)
    uploads, replayed = 0, 0
    orig_up, orig_j = tc._DeviceTree._upload, tc._DeviceTree.apply_journal

    def counting_up(self):
        nonlocal uploads
        uploads += 1
        return orig_up(self)

    def counting_j(self):
        nonlocal replayed
        n = orig_j(self)
        replayed += max(n, 0)
        return n

    tc._DeviceTree._upload, tc._DeviceTree.apply_journal = counting_up, counting_j
    try:
        tree.init_prompt(torch.arange(1, 101, dtype=torch.int32))
        tree.branch(tree.root, 8)
        for leaf in list(tree.leaves.values()):
            leaf.append_token(7)
        tree.alloc()
        deft_amd.TreeMetadata.from_tree_cache(tree, device_build=True)
        for accepted in (2, 1):
            leaves = list(tree.leaves.values())
            before = len(tree.root.kv_indices)
            for i in range(accepted):
                tree.merge_nodes(tree.root, leaves[i], pruneB_flag=False)
            diff = len(tree.root.kv_indices) - before
            for leaf in leaves:
                tree.reset_node_KV(leaf, diff)
            tree.alloc()
            md = deft_amd.TreeMetadata.from_tree_cache(tree, device_build=True)
        torch.cuda.synchronize()
    finally:
        tc._DeviceTree._upload, tc._DeviceTree.apply_journal = orig_up, orig_j
    assert uploads == 2 and replayed > 0  # first build; first merge (no room).  The second iteration went through the journal.
    g = golden("spec_mock")
    got = md_numpy(md)
    for k in MD_FIELDS:
        assert np.array_equal(got[k], g[k]), k
    assert [md.query_num, md.node_num, md.total_kv_len, md.block_len] == g["scalars"].tolist()
    assert np.array_equal(pool.mem_state, g["pool_refcounts"])


def _two_leaf_tree(prompt=50, slots=4096):
    req = deft_amd.ReqToTokenPool(16, slots, device="cuda")
    pool = deft_amd.TokenToKVPool(slots, torch.float16, 1, 128, 1, device="cuda")
    tree = deft_amd.TreeCache(torch.float16, 1, 128, 1, req, pool, None, True, False)
    tree.init_prompt(torch.arange(1, prompt + 1, dtype=torch.int32))
    tree.branch(tree.root, 2)
    return tree


def _step(tree):
    for leaf in list(tree.leaves.values()):
        leaf.append_token(7)
    tree.alloc()


def test_journal_too_long_for_one_replay_uploads_and_does_not_advance_twice():
    """A journal longer than one eager replay holds (64 + 8 nq + 4 n words: a 100-slot extend of a leaf on a 2-leaf tree) makes
    `alloc()` upload the tree instead -- AFTER the step's slots went into the native tree, so the image already ends in them and the
    device copy must not append them a second time (round 3 did: a duplicate slot per leaf from then on)."""
    tree = _two_leaf_tree()
    _step(tree)
    _assert_same(*_both(tree))  # the device copy exists and is current
    leaf = sorted(tree.leaves.values(), key=lambda n: n.id)[0]
    tree.extend_leaf(leaf, torch.arange(100, dtype=torch.int32))  # absorbed: the leaf has room for 256; journalled, 103 words
    epoch = tree._epoch()
    _step(tree)  # journal_take -> too long -> new epoch + upload inside alloc()
    assert tree._epoch() == epoch + 1
    _assert_same(*_both(tree))
    for _ in range(3):
        _step(tree)
        _assert_same(*_both(tree))
    assert tree._device_tree.dims()[9] == 0


def test_second_device_copy_inside_an_epoch_does_not_replay_a_stale_journal():
    """A pending journal and a NEW device copy in the same epoch (another max_q_len): the fresh image carries the journalled slots,
    the journal must not be replayed on top of it."""
    tree = _two_leaf_tree()
    _step(tree)
    _assert_same(*_both(tree))
    leaf = sorted(tree.leaves.values(), key=lambda n: n.id)[1]
    tree.extend_leaf(leaf, torch.arange(5, dtype=torch.int32))  # journalled EXTEND, not yet taken
    first = tree._device_tree
    _assert_same(*_both(tree, max_q_len=16))  # different cfg: a new _DeviceTree, uploaded with the extend inside
    assert tree._device_tree is not first
    _step(tree)  # must find an empty journal
    _assert_same(*_both(tree, max_q_len=16))
    _step(tree)
    _assert_same(*_both(tree, max_q_len=16))


def test_malformed_journal_words_raise_the_error_flag_instead_of_spinning():
    """The replay kernel reads its word count at run time from a reused staging buffer: a RESET word that carries slots (k != 0)
    must be rejected like any other malformed op -- it used to pass validation, match no branch and never advance the walk."""
    from deft_amd._lib import check, lib

    tree = _two_leaf_tree()
    _step(tree)
    _assert_same(*_both(tree))
    dt = tree._device_tree
    ops = torch.tensor([5, 2, 1, 2, 99, 98], dtype=torch.int32, device="cuda")  # {RESET, node 1, k = 2, two stray words}
    stream = torch.cuda.current_stream().cuda_stream
    check(lib.deft_tree_dev_apply_ops(*dt._tree_args(), ops.data_ptr(), dt.scratch.data_ptr(), stream), "deft_tree_dev_apply_ops")
    torch.cuda.synchronize()
    assert dt.dims()[9] & 4


def test_a_slot_merged_into_a_node_twice_keeps_the_device_list_whole():
    """`merge_nodes(A, B, pruneB_flag=False)` twice without the reset between them hands A a slot it already holds
    (tree_cache.py:300-325 allows it; the speculative-decoding mock always resets).  The device replay of that EXTEND used the LOWER
    bound for the new slot's place: its twin, shifted by the new slots strictly below it, landed on the same position, one of the two
    was lost and a stale word stayed in the node's list -- a wild slot number in stage 1 (round 5: an aperture violation on the
    GPU).  Device-built metadata must equal the host builder's through the whole sequence."""
    tree = _two_leaf_tree()
    _step(tree)
    _assert_same(*_both(tree))  # the device copy exists and is current
    leaves = sorted(tree.leaves.values(), key=lambda n: n.id)
    tree.merge_nodes(tree.root, leaves[0], pruneB_flag=False)  # journalled EXTEND of the root by the leaf's slot
    _step(tree)
    _assert_same(*_both(tree))
    tree.merge_nodes(tree.root, leaves[0], pruneB_flag=False)  # ... again: [the slot the root already holds, the new one]
    _step(tree)
    _assert_same(*_both(tree))
    root_slots = list(tree.root.kv_indices)
    assert len(root_slots) == 50 + 1 + 2 and len(set(root_slots)) == 50 + 2  # one slot twice
    for lf in leaves:
        tree.reset_node_KV(lf, 0)
    _step(tree)
    _assert_same(*_both(tree))
    assert tree._device_tree.dims()[9] == 0  # no error flag raised by the replay
