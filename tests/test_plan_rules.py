"""The plan's launch-shape rules that round 6 added, pinned on the shapes they were measured on (profiles/r6_chunk_sweep_short.txt,
profiles/r6_union_len_sweep.txt): what the chunk leaders of a device-built Flatten plan look like -- tiles per leader, by the leader's
query rows -- for the north-star tree at several branch lengths and for configs[1]'s 1k-prefix tree at short branches."""
from collections import Counter

import pytest
import torch

import deft_amd
from deft_amd._lib import check, lib
from deft_amd.utils.workloads import WORKLOADS, Workload, build_tree

pytestmark = pytest.mark.gpu

Hq = Hkv = 32
D = 128


def _leaders(workload, branch_len):
    """[(virtual query rows, tiles)] of the plan's chunk leaders (one KV head's worth)."""
    w = Workload(**{**WORKLOADS[workload].__dict__, "branch_len": branch_len})
    tree, pool = build_tree(w, 1, "cuda:0")
    md = deft_amd.TreeMetadata.from_tree_cache(tree)
    mdl = [md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv, md.block_lens]
    NB, P = md.block_q_cnts.shape[0], md.block_q.shape[0]
    nbytes = lib.deft_flatten_plan_bytes(NB, P, Hq, Hkv)
    plan = torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0")
    check(lib.deft_flatten_build_plan(*[t.data_ptr() for t in mdl], NB, P, Hq, Hkv, Hq * D, D, pool.kv_data[0].stride(0), None, 0, 0,
                                      plan.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream), "deft_flatten_build_plan")
    torch.cuda.synchronize()
    hdr = plan[:16].view(torch.int32).cpu().numpy()
    assert int(hdr[2]) == 0, "plan error flags"
    NL = int(hdr[1])
    desc = plan[4096 : 4096 + 2048 * NL].view(-1, 2048)[:, 1536:1568].contiguous().view(torch.int32).view(-1, 8).cpu().numpy()
    assert int(desc[:, 4].sum()) == NB  # every block is a tile of exactly one leader's chunk
    return [(int(r[0]), int(r[4])) for r in desc]


def test_prefix_chunks_of_the_north_star_tree_by_branch_length():
    """A 4096-token prefix under 32 branches (MHA): 8-tile chunks at 200 tokens per branch; while the prefix dominates the tree (short
    branches) it is cut into 7 or 6 chunks per KV head instead of 4."""
    for bl, chunks in ((200, 4), (50, 6), (10, 7)):
        lead = _leaders("northstar_4kx32", bl)
        prefix = [t for rows, t in lead if rows == 32 and t >= 3]
        assert sum(prefix) == 32 and len(prefix) == chunks, (bl, prefix)


def test_leaf_blocks_in_threes_or_fours_by_list_scheduling():
    """Large MHA launch: leaf blocks go in groups of three -- and of four where threes would finish before the prefix's chunks do
    (~205-270 tokens per branch on this tree: 254 tokens 41.4 -> 37.9 us per layer)."""
    for bl, length in ((150, 3), (200, 3), (230, 4), (254, 4), (300, 3), (400, 3)):
        lead = _leaders("northstar_4kx32", bl)
        groups = Counter(t for rows, t in lead if rows < 32)
        other = 7 - length
        assert groups[length] >= (32 * bl // 128) // length - 1 and groups[other] == 0, (bl, dict(groups))


def test_a_short_prefix_under_single_tile_branches_is_cut_into_two_tile_chunks():
    """configs[1]'s tree (1024-token prefix x 32 branches) at 50 tokens per branch: four 2-tile chunks of the prefix (its 4-tile chain was the
    launch's critical path); at 200 tokens per branch -- more than four times the prefix's tiles beside it -- 4-tile chunks again."""
    short = [t for rows, t in _leaders("fewshot_1kx32", 50) if rows == 32]
    long_ = [t for rows, t in _leaders("fewshot_1kx32", 200) if rows == 32]
    assert short == [2, 2, 2, 2] and long_ == [4, 4], (short, long_)
