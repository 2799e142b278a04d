"""CPU tests of the product's host side: the C-ABI library loads and exports every
symbol include/deft_amd.h declares, and the tree / pool / native metadata builder
reproduce the reference's golden vectors bit for bit.  No GPU, no oracle."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import deft_amd
from deft_amd._lib import EXPORTED, LIB_PATH
from product_helpers import MD_FIELDS, md_numpy, product_metadata, product_tree
from scenarios import SCENARIOS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "deft_amd.h")).read()
    declared = set(re.findall(r"\b(deft_[a-z0-9_]+)\s*\(", header))
    assert declared == set(EXPORTED), declared ^ set(EXPORTED)
    so = ctypes.CDLL(LIB_PATH)
    for name in declared:
        assert hasattr(so, name), name
    assert so.deft_abi_version() == 2


def test_library_exports_nothing_but_the_declared_symbols():
    """`nm -D`: the dynamic symbol table of the shipped library is the header's list -- no kernel handles, no C++ helpers, no
    libstdc++ instantiations (-fvisibility=hidden + deft_amd/csrc/exports.map)."""
    import shutil
    import subprocess

    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", LIB_PATH], check=True, capture_output=True, text=True).stdout
    defined = {line.split()[-1] for line in out.splitlines() if line.strip()}
    header = open(os.path.join(ROOT, "include", "deft_amd.h")).read()
    declared = set(re.findall(r"\b(deft_[a-z0-9_]+)\s*\(", header))
    assert defined == declared, sorted(defined ^ declared)


def test_supported_geometries():
    lib = deft_amd.lib
    assert lib.deft_supported(32, 32, 128) == 1 and lib.deft_supported(32, 8, 128) == 1
    assert lib.deft_supported(4, 4, 64) == 1
    assert lib.deft_supported(32, 5, 128) == 0 and lib.deft_supported(32, 8, 96) == 0


def test_argument_errors_are_reported_not_thrown():
    lib = deft_amd.lib
    rc = lib.deft_flatten_decode_f16(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 4, 4, 128, 0.1, 0, 0, 0, 0)
    assert rc == -1 and b"null" in lib.deft_last_error()
    assert lib.deft_md_free(12345) == -1
    assert lib.deft_md_build(0, 0, 0, 0, 0, 0, 32, 128, -1) == -1


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_tree_state_matches_reference(name, golden):
    g = golden(name)
    tree = product_tree(name)
    ids = sorted(tree.nodes)
    assert ids == g["node_ids"].tolist()
    assert [s for i in ids for s in tree.nodes[i].kv_indices] == g["node_kv_by_id"].tolist()
    assert np.array_equal(tree.token_to_kv_pool.mem_state, g["pool_refcounts"])
    assert sorted(tree.leaves) == g["leaf_ids"].tolist()


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_native_metadata_bit_exact(name, golden):
    g = golden(name)
    tree = product_tree(name)
    md = product_metadata(name, tree)
    got = md_numpy(md)
    for k in MD_FIELDS:
        assert got[k].dtype == np.int64
        assert np.array_equal(got[k], g[k]), k
    assert [md.query_num, md.node_num, md.total_kv_len, md.block_len] == g["scalars"].tolist()
    assert md.leaf_to_q == {int(l): i for i, l in enumerate(g["leaf_ids"])}


def test_page_table_rows_are_root_to_leaf_paths():
    """ReqToTokenPool rows (memory_pool.py:11-45) list each leaf's path slots in order."""
    tree = product_tree("multilevel")
    table = tree.req_to_token_pool.req_to_token
    for leaf in tree.leaves.values():
        path = tree.leaf_path_slots(leaf)
        row = table[tree.leaf_to_req[leaf.id], : len(path)].tolist()
        assert row == path


def test_empty_node_is_rejected_like_upstream():
    """The reference raises on a live node without KV (range() step 0, tree_cache.py:746-748)."""
    tree = product_tree("cfgA_256x2")
    kids = tree.branch(next(iter(tree.leaves.values())), 2)
    kids[0].append_token(1)
    kids[1].append_token(1)  # no alloc(): children have no slot yet
    with pytest.raises(deft_amd.DeftLibraryError, match="no KV slot"):
        deft_amd.TreeMetadata.from_tree_cache(tree, device="cpu")


def test_pool_exhaustion_returns_none_like_upstream():
    pool = deft_amd.TokenToKVPool(4, torch.float16, 1, 8, 1, device="cpu")
    assert pool.alloc(3) is not None
    assert pool.alloc(2) is None  # memory_pool.py:76-77
    pool.free(np.array([1]))
    assert pool.alloc(2).tolist() == [1, 3]  # lowest free slots first


def test_out_of_scope_modes_fail_loudly():
    with pytest.raises(NotImplementedError):
        deft_amd.forward_mode_from_cli("tree")
    with pytest.raises(NotImplementedError):
        deft_amd.forward_mode_from_cli("flatten", "unpaged")
    assert deft_amd.forward_mode_from_cli("seq") is deft_amd.ForwardMode.DECODE  # the sequential comparator
    assert deft_amd.forward_mode_from_cli("deft_flatten") is deft_amd.ForwardMode.TREE_DECODE_FLATTEN
    assert deft_amd.forward_mode_from_cli("deft_node") is deft_amd.ForwardMode.TREE_DECODE_NODE
    with pytest.raises(NotImplementedError):
        deft_amd.TreeCache(torch.float16, 1, 8, 1, None, None, None, use_paged_memory=False)


def test_operators_refuse_cpu_tensors():
    q = torch.zeros(1, 4, 128, dtype=torch.float16)
    kv = torch.zeros(8, 4, 128, dtype=torch.float16)
    i64 = torch.zeros(128, dtype=torch.int64)
    with pytest.raises(deft_amd.DeftLibraryError, match="no CPU path"):
        deft_amd.tree_attention_subtree_fwd(q, kv, kv, q.clone(), 128, i64[:1], i64[:1], i64[:1], i64, i64, i64[:1])


def test_forest_metadata_is_the_offset_concatenation_of_its_trees():
    """deft_amd.Forest: several trees in one pool -> one TreeMetadata whose arrays are the per-tree arrays one
    after the other, query rows / offsets shifted per tree.  Checked against each tree's own
    metadata (itself pinned bit-exact to the reference goldens above)."""
    names = ["multilevel", "wide40", "chain_300", "edge_fill"]
    req = deft_amd.ReqToTokenPool(256, 4096, device="cpu")
    pool = deft_amd.TokenToKVPool(8192, torch.float16, 1, 8, 1, device="cpu")
    trees = []
    for n in names:
        t = deft_amd.TreeCache(torch.float16, 1, 8, 1, req, pool, None, True, False)
        SCENARIOS[n].script(t, lambda k: torch.arange(1, k + 1, dtype=torch.int32))
        trees.append(t)
    forest = deft_amd.Forest(trees)
    md = forest.metadata(device="cpu")
    singles = [deft_amd.TreeMetadata.from_tree_cache(t, device="cpu") for t in trees]
    assert md.query_num == sum(s.query_num for s in singles) == forest.query_num
    assert md.node_num == sum(s.node_num for s in singles)
    assert md.total_kv_len == sum(s.total_kv_len for s in singles)
    qb = nq = nkv = bq = 0
    pos = {k: 0 for k in MD_FIELDS}
    for t, s in enumerate(singles):
        assert md.q_bases[t] == qb
        for k in MD_FIELDS:
            a = getattr(s, k).numpy().copy()
            if k in ("node_q", "block_q"):
                a += qb
            elif k == "node_q_offset":
                a += nq
            elif k == "node_kv_offset":
                a += nkv
            elif k == "block_q_offset":
                a += bq
            got = getattr(md, k).numpy()[pos[k] : pos[k] + len(a)]
            assert np.array_equal(got, a), (t, k)
            pos[k] += len(a)
        for leaf, qi in s.leaf_to_q.items():
            assert md.leaf_to_q[(t, leaf)] == qb + qi
        qb += s.query_num
        nq += len(s.node_q)
        nkv += len(s.node_kv)
        bq += len(s.block_q)
    for k in MD_FIELDS:
        assert pos[k] == len(getattr(md, k))
    # a forest step hands out one slot per live leaf, in batch query order
    before = [len(l.kv_indices) for t in trees for l in sorted(t.leaves.values(), key=lambda n: n.id)]
    for t in trees:
        for leaf in t.leaves.values():
            leaf.append_token(9)
    upd = forest.alloc()
    leaves = [l for t in trees for l in sorted(t.leaves.values(), key=lambda n: n.id)]
    assert upd.cache_loc.tolist() == [l.kv_indices[-1] for l in leaves]
    assert [len(l.kv_indices) for l in leaves] == [b + 1 for b in before]
    assert len(forest.leaf_paths()) == forest.query_num


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_native_tree_tracks_every_mutation(name):
    """Every TreeCache mutation is an operation on the native tree (deft_tree_*); the metadata built from it equals
    the one built by marshalling the Python-visible tree into the stateless deft_md_build."""
    from deft_amd import tree_cache as tc

    sc = SCENARIOS[name]
    tree = product_tree(name)
    assert tc._mirror_consistent(tree)
    a = tc.build_metadata_host(tree, sc.max_q_len, sc.block_len, sc.max_block_len, use_mirror=True)
    b = tc.build_metadata_host(tree, sc.max_q_len, sc.block_len, sc.max_block_len, use_mirror=False)
    for k in MD_FIELDS:
        assert np.array_equal(a[k], b[k]), k
    assert a["leaf_to_q"] == b["leaf_to_q"] and a["query_num"] == b["query_num"]
    # the reference's `refs` sets, recomputed from the native tree: the live leaves below every node
    for node in tree.nodes.values():
        below = set()
        stack = [node]
        while stack:
            cur = stack.pop()
            if cur.id in tree.leaves:
                below.add(cur.id)
            stack.extend(cur.children.values())
        assert {r.id for r in node.refs} == below


def test_direct_edits_of_kv_indices_reach_the_metadata():
    """Scripts written against the reference append to / assign node.kv_indices directly: the list is a view of the
    native tree, so the edit is in the next metadata; edits of tree.nodes / tree.leaves behind TreeCache's back are
    refused instead of producing metadata of a different tree."""
    from deft_amd import tree_cache as tc

    tree = product_tree("multilevel")
    leaf = sorted(tree.leaves.values(), key=lambda n: n.id)[0]
    slot = tree.token_to_kv_pool.alloc_host(1)
    before = len(leaf.kv_indices)
    leaf.kv_indices.append(int(slot[0]))
    assert len(leaf.kv_indices) == before + 1 and leaf.kv_indices[-1] == int(slot[0])
    assert tc._mirror_consistent(tree)
    md = deft_amd.TreeMetadata.from_tree_cache(tree, device="cpu")
    assert int(slot[0]) in md.block_kv.tolist()
    saved = list(leaf.kv_indices)
    leaf.kv_indices = saved[:-1]
    assert leaf.kv_indices == saved[:-1]
    tree.leaves.pop(leaf.id)  # behind TreeCache's back
    assert not tc._mirror_consistent(tree)
    with pytest.raises(RuntimeError, match="behind TreeCache"):
        deft_amd.TreeMetadata.from_tree_cache(tree, device="cpu")




def test_pool_allocator_picks_the_lowest_free_slots():
    """TokenToKVPool.alloc_host searches from a hint (lowest slot that can be free) in chunks instead of scanning the whole pool
    per decode step; it must hand out exactly what the reference's `nonzero(mem_state == 0)[:n]` would (memory_pool.py:75-84)
    through any mix of allocs, partial frees, extra references and clears."""
    import random

    from deft_amd.memory_pool import TokenToKVPool

    rng = random.Random(1)
    for _trial in range(12):
        size = rng.choice([50, 5000, 20000])
        pool = TokenToKVPool(size, torch.float16, 1, 8, 0, device="cpu")
        ref = np.zeros(size, dtype=np.int16)
        held = []
        for _op in range(300):
            r = rng.random()
            if r < 0.5:
                need = rng.choice([0, 1, 3, 17, 64, 300, 5000])
                exp = np.flatnonzero(ref == 0)[:need]
                got = pool.alloc_host(need)
                if exp.shape[0] < need:
                    assert got is None
                else:
                    assert got is not None and np.array_equal(got, exp)
                    np.add.at(ref, exp, 1)
                    held.append(exp)
            elif r < 0.85 and held:
                blk = held.pop(rng.randrange(len(held)))
                k = rng.randint(0, len(blk))
                if k < len(blk):
                    held.append(blk[k:])
                if k:
                    pool.free(torch.from_numpy(blk[:k])) if rng.random() < 0.5 else pool.decrease_refs(blk[:k])
                    np.subtract.at(ref, blk[:k], 1)
            elif r < 0.93 and held:
                blk = held[rng.randrange(len(held))][:3]
                pool.add_refs(blk)
                np.add.at(ref, blk, 1)
                held.append(blk.copy())
            elif r < 0.96:
                pool.clear()
                ref[:] = 0
                held = []
            assert np.array_equal(pool.mem_state, ref)
    # entries cleared behind the allocator's back (below its hint) are found once the hinted search runs dry
    pool = TokenToKVPool(100, torch.float16, 1, 8, 0, device="cpu")
    assert np.array_equal(pool.alloc_host(90), np.arange(90))
    pool.mem_state[:40] = 0
    got = pool.alloc_host(30)
    assert got is not None and len(set(got.tolist())) == 30 and (pool.mem_state[got] == 1).all()
    assert pool.alloc_host(21) is None and pool.alloc_host(20) is not None


def _small_tree(prefix=5, size=64):
    req = deft_amd.ReqToTokenPool(16, size + 8, device="cpu")
    pool = deft_amd.TokenToKVPool(size, torch.float16, 1, 8, 1, device="cpu")
    tree = deft_amd.TreeCache(torch.float16, 1, 8, 1, req, pool, None, True, False)
    tree.init_prompt(torch.arange(1, prefix + 1, dtype=torch.int32))
    return tree


def test_cut_keeps_an_ancestor_that_still_has_children():
    """branch 2, new_node(root), cut both leaves -- the native tree erased the root although the node made by
    new_node() still hung below it, and the next walk over that node threw std::out_of_range across the C ABI (abort).
    The walk now stops at an ancestor that keeps other children; every later call returns instead of aborting."""
    from deft_amd._lib import lib

    tree = _small_tree()
    a, b = tree.branch(tree.root, 2)
    extra = tree.new_node(tree.root)  # holds no live leaf
    for leaf in (a, b):
        leaf.append_token(3)
    tree.alloc()
    gone = tree.cut(a)
    assert [n.id for n in gone] == [a.id]
    gone = tree.cut(b)
    assert [n.id for n in gone] == [b.id]  # the root keeps `extra`, so it stays
    assert 0 in tree.nodes and extra.id in tree.nodes
    tree.add_ref(extra)  # (used to abort the process)
    assert tree.leaf_path_slots(extra) == tree.root.kv_indices.tolist()
    assert [n.id for n in tree.root.refs] == [extra.id]
    assert lib.deft_tree_set_leaf(tree._native, 999, 1) == -1  # unknown ids are errors, not exceptions


def test_device_path_sizes_reject_what_the_host_builder_rejects():
    """`from_tree_cache` right after `branch()` with no `alloc()`: deft_md_build (and the reference, range() with step 0)
    raise; the size call in front of the device kernels must too, instead of letting them skip the empty nodes."""
    from deft_amd._lib import DeftLibraryError, check, lib
    from deft_amd.tree_cache import _ptr

    tree = _small_tree()
    tree.branch(tree.root, 3)
    with pytest.raises(DeftLibraryError, match="no KV slot"):
        deft_amd.TreeMetadata.from_tree_cache(tree)  # host builder (CPU pool)
    sizes = np.zeros(5, dtype=np.int64)
    check(lib.deft_tree_layout(tree._native, 256, _ptr(sizes)), "layout")
    out = np.zeros(9, dtype=np.int64)
    assert lib.deft_tree_md_sizes(tree._native, 32, 128, -1, 0, _ptr(out)) == -1
    assert b"no KV slot" in lib.deft_last_error()
    assert lib.deft_tree_md_caps(tree._native, 32, 128, -1, 260, _ptr(out)) == -1
    for leaf in tree.leaves.values():
        leaf.append_token(1)
    tree.alloc()
    check(lib.deft_tree_layout(tree._native, 256, _ptr(sizes)), "layout")
    check(lib.deft_tree_md_sizes(tree._native, 32, 128, -1, 0, _ptr(out)), "sizes")
    assert out[0] == 3 and out[2] == 5 + 3


def test_absorbed_changes_are_journalled_not_new_epochs():
    """merge_nodes into a node that has room in the GPU layout and reset_node(s)_KV keep the structural epoch: the native tree
    journals them for the device copy (the reference's speculative-decoding mock does both every step,
    branch_func_example.py:420-437).  The first merge into the root has no room yet -- only leaves are laid out with room -- and
    starts an epoch; the root is remembered as growing and laid out with room from then on."""
    from deft_amd._lib import check, lib
    from deft_amd.tree_cache import _ptr

    tree = _small_tree(prefix=8, size=256)  # (a node's room is rounded up to four slots: 8 leaves none)
    leaves = tree.branch(tree.root, 4)

    def step():
        for lf in tree.leaves.values():
            lf.append_token(5)
        return tree.alloc().cache_loc.tolist()

    def layout():
        sizes = np.zeros(5, dtype=np.int64)
        check(lib.deft_tree_layout(tree._native, 256, _ptr(sizes)), "layout")
        return int(sizes[4])

    def spec_step(accept):
        before = len(tree.root.kv_indices)
        for lf in leaves[:accept]:
            tree.merge_nodes(tree.root, lf, pruneB_flag=False)
        tree.reset_nodes_KV(leaves, len(tree.root.kv_indices) - before)

    step()
    e0 = layout()
    buf = np.zeros(256, dtype=np.int32)
    spec_step(2)  # root without room: a structural change
    assert tree._epoch() > e0 and lib.deft_tree_journal_take(tree._native, _ptr(buf), 256) == 0
    slots = step()
    e1 = layout()  # (what an upload does) -- the root now has room
    root_before = tree.root.kv_indices.tolist()
    spec_step(3)
    assert tree._epoch() == e1  # absorbed
    assert tree.root.kv_indices.tolist() == root_before + slots[:3]
    assert all(len(lf.kv_indices) == 0 for lf in leaves)
    n = int(lib.deft_tree_journal_take(tree._native, _ptr(buf), 256))
    words = buf[:n].tolist()
    # the three one-slot merges into the root (DFS index 0) as ONE EXTEND, then a RESET per leaf (DFS indices 1..4)
    assert words[:6] == [1, 0, 3, slots[0], slots[1], slots[2]]
    assert words[6:] == [2, 1, 0, 2, 2, 0, 2, 3, 0, 2, 4, 0]
    assert lib.deft_tree_journal_take(tree._native, _ptr(buf), 256) == 0  # handed over once
    step()
    assert tree._epoch() == e1  # a decode step still fits
    # the host builder sees the same tree (root grown, leaves one token each)
    md = deft_amd.TreeMetadata.from_tree_cache(tree)
    assert md.total_kv_len == len(tree.root.kv_indices) + 4
    # a journal that does not fit the caller's buffer becomes a structural change
    spec_step(1)
    assert lib.deft_tree_journal_take(tree._native, _ptr(buf), 4) == -5 and tree._epoch() > e1
    # pool refcounts: accepted slots are held once (by the root), released leaf slots are free again
    held = sorted(s for nd in tree.nodes.values() for s in nd.kv_indices)
    assert sorted(np.nonzero(tree.token_to_kv_pool.mem_state)[0].tolist()) == held
    assert set(np.asarray(tree.token_to_kv_pool.mem_state)[held].tolist()) == {1}


def test_layout_fetch_hands_over_the_journal_with_the_image():
    """deft_tree_layout_fetch writes the tree as it is NOW -- journalled changes included -- so the journal is empty afterwards:
    a second device copy made inside one epoch must not replay an EXTEND its image already holds."""
    from deft_amd._lib import check, lib
    from deft_amd.tree_cache import _ptr

    tree = _small_tree(prefix=8, size=256)
    leaves = tree.branch(tree.root, 2)
    for lf in tree.leaves.values():
        lf.append_token(5)
    tree.alloc()
    sizes = np.zeros(5, dtype=np.int64)
    check(lib.deft_tree_layout(tree._native, 256, _ptr(sizes)), "layout")
    e = int(sizes[4])
    tree.extend_leaf(leaves[0], torch.arange(3, dtype=torch.int32))
    assert tree._epoch() == e  # absorbed by the layout, journalled
    n, nq, nqw, total_cap = (int(x) for x in sizes[:4])
    start, ln, cap = (np.zeros(n, dtype=np.int32) for _ in range(3))
    refs, leaf_node, slots = np.zeros(n * nqw, dtype=np.uint64), np.zeros(nq, dtype=np.int32), np.zeros(total_cap, dtype=np.int32)
    check(lib.deft_tree_layout_fetch(tree._native, _ptr(start), _ptr(ln), _ptr(cap), _ptr(refs), _ptr(leaf_node), _ptr(slots)), "fetch")
    assert ln.tolist() == [8, 1 + 3, 1]  # the image has the extend
    buf = np.zeros(64, dtype=np.int32)
    assert lib.deft_tree_journal_take(tree._native, _ptr(buf), 64) == 0  # ... and took the journal with it
    # any OTHER device copy of epoch e never saw that extend and can no longer get it from the journal, so a fetch that
    # swallowed a non-empty journal ends the epoch (the fetching copy reads the epoch after the call and adopts it) ...
    assert tree._epoch() == e + 1
    check(lib.deft_tree_layout(tree._native, 256, _ptr(sizes)), "layout")
    assert int(sizes[4]) == e + 1 and [int(x) for x in sizes[:4]] == [n, nq, nqw, total_cap]  # ... while the layout itself stays
    check(lib.deft_tree_layout_fetch(tree._native, _ptr(start), _ptr(ln), _ptr(cap), _ptr(refs), _ptr(leaf_node), _ptr(slots)), "fetch")
    assert tree._epoch() == e + 1  # a fetch with an empty journal changes nothing


def _build_c_program(out_path):
    """The plain-C program of tests/c_abi against include/deft_amd.h and libdeft_amd.so, with gcc as a C11 compiler."""
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "deft_amd", "lib")
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no gcc / no ROCm headers")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(root, "include"), "-I", "/opt/rocm/include",
           os.path.join(root, "tests", "c_abi", "decode_from_c.c"), "-L", lib_dir, "-ldeft_amd", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", out_path]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return out_path


def test_header_and_library_serve_a_plain_c_program(tmp_path):
    """include/deft_amd.h is a C header (no C++ in the boundary), and a C11 program links against the shared library with gcc alone;
    running it needs a GPU (tests/test_gpu_parity.py::test_c_program_through_the_c_abi)."""
    exe = _build_c_program(str(tmp_path / "decode_from_c"))
    assert os.path.getsize(exe) > 0


def test_stage_copy_refuses_a_slot_header_that_names_too_much():
    """deft_stage_copy reads `used` from the slot's header on the host: more than the slot or the destination holds is DEFT_EINVAL
    before any HIP call (so is a null pointer); zero bytes is a no-op."""
    lib = deft_amd.lib
    ring = np.zeros(2 * 4096, dtype=np.uint8)
    p = ring.ctypes.data_as(ctypes.c_void_p)
    dst = ctypes.c_void_p(0x1000)  # (never dereferenced on these paths)
    assert lib.deft_stage_copy(None, 4096, 0, dst, 4096, None) == -1
    assert lib.deft_stage_copy(p, 4096, 0, None, 4096, None) == -1
    assert lib.deft_stage_copy(p, 4096, -1, dst, 4096, None) == -1
    assert lib.deft_stage_copy(p, 4090, 0, dst, 4096, None) == -1  # (slots are multiples of 16 bytes)
    ring[4096:4100].view(np.uint32)[0] = 4090  # slot 1: 4090 + 16 > 4096
    assert lib.deft_stage_copy(p, 4096, 1, dst, 1 << 20, None) == -1
    assert b"header" in lib.deft_last_error()
    ring[4096:4100].view(np.uint32)[0] = 512
    assert lib.deft_stage_copy(p, 4096, 1, dst, 256, None) == -1  # (the destination is smaller)
    assert lib.deft_stage_copy(p, 4096, 0, dst, 4096, None) == 0  # used = 0: nothing to copy


def test_window_books_from_a_plain_c_program(tmp_path):
    """The window plan's host-side entry points (deft_window_supported / _create / _step / _free) called from C11 through the header
    alone: host-only, so this one RUNS here (tests/c_abi/window_books_from_c.c)."""
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "deft_amd", "lib")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "window_books_from_c")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_abi", "window_books_from_c.c"),
           "-L", lib_dir, "-ldeft_amd", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_finished_branches_are_recorded_and_printed(capsys):
    """TreeCache.output_branch / print_finished_branches (tree_cache.py:525-567): the branch below the root, its cumulative
    log-probability and perplexity."""
    import math

    tree = _small_tree(prefix=4, size=64)
    a, b = tree.branch(tree.root, 2)
    a.append_token(11, logprob=math.log(0.5))
    a.append_token(12, logprob=math.log(0.25))
    b.append_token(21)
    tree.output_branch(dstnode=a)
    tree.output_branch(dstnode=b)
    s0, s1 = tree.all_finished_seqs
    assert (s0.id, s0.token_ids, s1.id, s1.token_ids) == (0, [11, 12], 1, [21])
    assert s0.cumulative_logprob == pytest.approx(math.log(0.125)) and s0.PPL == pytest.approx(math.exp(-math.log(0.125) / 2))
    assert s1.PPL == 1.0 and s0.get_len() == 2

    class Tok:
        def decode(self, ids, skip_special_tokens=True):
            return " ".join(str(i) for i in ids)

    tree.print_finished_branches(Tok())
    out = capsys.readouterr().out
    assert "Total number of generated branches=2" in out and "Generated Text: 11 12" in out and "Token length : 1" in out


def test_window_books_patch_lists():
    """The host-side books of a window plan (deft_amd/csrc/window_host.cpp; deft_amd.DecodeSession's incremental steps): overflow
    positions are handed out in order per REGION (one per (query chunk, 32-row pass) pair), the rows a step read from k_new / v_new
    get their pool slot on the NEXT step, a RESET node keeps its positions and refills them, a RESET of a node the static plan
    still holds -- and a full region -- ask for a replan."""
    import ctypes as C

    from deft_amd._lib import lib

    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    leaf = np.asarray([1, 2, 3], dtype=np.int32)
    refs = np.asarray([0b111, 0b001, 0b010, 0b100], dtype=np.uint64)  # root above all three leaves

    def make(group, max_q_len=32):
        w = int(lib.deft_window_create(4, 3, 1, 1, ptr(leaf), ptr(refs), max_q_len, group, 64))
        assert w > 0
        return w

    out = np.zeros(1 + 64 + 3 * 64, dtype=np.int32)
    none = np.zeros(1, dtype=np.int32)

    def step(w, replan, journal, loc):
        j = np.asarray(journal, dtype=np.int32) if len(journal) else none
        n = int(lib.deft_window_step(w, replan, ptr(j), len(journal), ptr(np.asarray(loc, dtype=np.int32)), ptr(out), out.size))
        if n < 0:
            return None
        assert n == 65 + 3 * out[0]
        return [int(x) for x in out[1:65] if x], {(int(k) >> 20, int(k) & 0xfffff): (int(a), int(b)) for k, a, b in out[65:n].reshape(-1, 3)}

    # (an entry = position: (row mask, slot | -1 - new row); MHA: leaf r is row r of the one region, the root is above all three)
    w = make(1)  # MHA, three queries: one chunk, one pass, one region
    assert step(w, 0, [], [10, 11, 12]) is None  # no window yet
    assert step(w, 1, [], [10, 11, 12]) == ([1], {(0, 0): (1, -1), (0, 1): (2, -2), (0, 2): (4, -3)})
    # the next step: last step's rows are pool rows now, this step's go behind them
    assert step(w, 0, [], [13, 14, 15])[1] == {(0, 0): (1, 10), (0, 1): (2, 11), (0, 2): (4, 12), (0, 3): (1, -1), (0, 4): (2, -2), (0, 5): (4, -3)}
    # a RESET of a leaf whose first tokens sit in the static part of the plan cannot be patched
    assert step(w, 0, [2, 1, 0], [16, 17, 18]) is None
    assert step(w, 0, [], [16, 17, 18]) is None  # ... and the books stay invalid until a replan
    # a speculative-decoding loop: the replan step's journal resets every leaf (they are "clean" from now on) ...
    assert step(w, 1, [1, 0, 1, 15, 2, 1, 0, 2, 2, 0, 2, 3, 0], [20, 21, 22])[1] == {(0, 0): (1, -1), (0, 1): (2, -2), (0, 2): (4, -3)}
    # ... and every later step merges a slot into the root (node 0: a new position, all three rows), drops the leaves' slots and
    # refills the SAME positions
    assert step(w, 0, [1, 0, 1, 20, 2, 1, 0, 2, 2, 0, 2, 3, 0], [23, 24, 25])[1] == {(0, 3): (7, 20), (0, 0): (1, -1), (0, 1): (2, -2), (0, 2): (4, -3)}
    assert step(w, 0, [1, 0, 2, 23, 24, 2, 1, 0, 2, 2, 0, 2, 3, 0], [26, 27, 28])[1] == {(0, 4): (7, 23), (0, 5): (7, 24), (0, 0): (1, -1), (0, 1): (2, -2),
                                                                                       (0, 2): (4, -3)}
    # a RESET without a refill: the positions are cleared (no row sees them)
    assert step(w, 1, [2, 1, 0, 2, 2, 0, 2, 3, 0], [50, 51, 52]) is not None
    r = step(w, 0, [2, 1, 0], [53, 54, 55])
    assert r[1] == {(0, 0): (1, -1), (0, 1): (2, 51), (0, 2): (4, 52), (0, 3): (2, -2), (0, 4): (4, -3)}  # leaf 1's position: cleared, then refilled
    # without resets the region fills up: 128 positions
    assert step(w, 1, [], [10, 11, 12]) is not None and step(w, 0, [], [13, 14, 15]) is not None
    fill, steps = 6, 0
    while True:
        r = step(w, 0, [1, 0, 3, 1, 2, 3], [30, 31, 32])  # three more slots into the root per step, the leaves keep growing
        if r is None:
            break
        fill += 6
        steps += 1
        assert r[0] == [(fill + 127) // 128]
    assert steps == (128 - 6) // 6 and fill <= 128
    assert step(w, 1, [], [40, 41, 42])[1] == {(0, 0): (1, -1), (0, 1): (2, -2), (0, 2): (4, -3)}  # the replan starts afresh
    assert lib.deft_window_free(w) == 0 and lib.deft_window_free(w) != 0
    # GQA, 16 query heads per KV head: two queries to a 32-row pass -- queries 0, 1 in region 0, query 2 in region 1; a leaf's token
    # goes to ITS region only, a slot merged into the root to both (masks: 16 rows per query)
    w = make(16)
    lo, hi, full = 0xffff, -0x10000, -1  # (int32 views of 0x0000ffff, 0xffff0000, 0xffffffff)
    assert step(w, 1, [], [10, 11, 12]) == ([1, 1], {(0, 0): (lo, -1), (0, 1): (hi, -2), (1, 0): (lo, -3)})
    assert step(w, 0, [1, 0, 1, 99], [13, 14, 15])[1] == {(0, 0): (lo, 10), (0, 1): (hi, 11), (1, 0): (lo, 12), (0, 2): (full, 99), (1, 1): (lo, 99),
                                                          (0, 3): (lo, -1), (0, 4): (hi, -2), (1, 2): (lo, -3)}
    lib.deft_window_free(w)
    # one query per chunk (max_q_len = 1): three chunks, three regions
    w = make(1, max_q_len=1)
    assert step(w, 1, [], [10, 11, 12]) == ([1, 1, 1], {(0, 0): (1, -1), (1, 0): (1, -2), (2, 0): (1, -3)})
    lib.deft_window_free(w)
