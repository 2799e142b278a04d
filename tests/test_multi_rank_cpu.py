"""N > 1 host path on CPU (gloo, world_size 2): forest sharding is deterministic, disjoint and
balanced; each rank builds its own trees' metadata with the native builder; the only
collectives are the timing MAX and the optional output all-gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deft_amd.utils.sharding import all_gather_outputs, cfg5_shard, max_over_ranks, shard_trees


def test_shard_trees_lpt():
    sizes = [8704, 4128, 10496, 1080, 7680, 258, 16896, 4128]
    shards = shard_trees(sizes, 4)
    assert sorted(i for s in shards for i in s) == list(range(len(sizes)))
    loads = [sum(sizes[i] for i in s) for s in shards]
    assert max(loads) <= 1.35 * (sum(sizes) / 4)
    assert shards == shard_trees(sizes, 4)  # deterministic
    assert shard_trees([5, 5], 4)[2:] == [[], []]  # more ranks than trees


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        here = os.path.dirname(os.path.abspath(__file__))
        sys.path.insert(0, here)
        from product_helpers import md_numpy, product_metadata, product_tree

        names = ["cfgA_256x2", "multilevel", "wide40", "edge128", "after_cut", "spec_mock"]
        sizes = []
        for n in names:
            t = product_tree(n)
            sizes.append(sum(len(nd.kv_indices) for nd in t.nodes.values()))
        mine = shard_trees(sizes, world)[rank]
        # each rank builds ONLY its trees' metadata (host builder in libdeft_amd.so), no exchange
        digest = 0
        for i in mine:
            md = product_metadata(names[i])
            digest += int(sum(int(v.sum()) for v in md_numpy(md).values()))
        slow = max_over_ranks(0.001 * (rank + 1), torch.device("cpu"))
        outs = all_gather_outputs(torch.full((len(mine), 4), float(rank)))
        # ragged forests: rank r holds r + 2 rows (the per-rank broadcast branch), and one of the ranks may hold none
        ragged = all_gather_outputs(torch.full((rank + 2, 3), 10.0 + rank))
        assert [tuple(o.shape) for o in ragged] == [(r + 2, 3) for r in range(world)]
        assert all(bool((o == 10.0 + r).all()) for r, o in enumerate(ragged))
        empty = all_gather_outputs(torch.full((0 if rank == 0 else 5, 2), 7.0))
        assert [tuple(o.shape) for o in empty] == [(0, 2)] + [(5, 2)] * (world - 1) and bool((empty[1] == 7.0).all())
        # bench.py's multi-GPU selection (BASELINE configs[4]): 8 trees per rank, disjoint, every tree once
        share = cfg5_shard(world, rank)
        both = [None] * world
        dist.all_gather_object(both, share)
        assert len(share) == 8 and sorted(i for s_ in both for i in s_) == list(range(8 * world))
        ret[rank] = (mine, digest, slow, [tuple(o.shape) for o in outs], [float(o.flatten()[0]) if o.numel() else -1 for o in outs])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_ranks_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    assert sorted(a[0] + b[0]) == list(range(6)) and not set(a[0]) & set(b[0])
    assert a[2] == b[2] == pytest.approx(0.002)  # MAX over ranks
    assert a[3] == b[3] and a[4] == b[4] == [0.0, 1.0]  # both ranks see both outputs
    assert a[1] > 0 and b[1] > 0
