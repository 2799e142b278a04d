#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

Build-container only: imports LINs-lab/DeFT from /root/reference (read-only,
never copied) with

  * TRITON_INTERPRET=1            the reference's Triton kernels run in Triton's
                                  CPU interpreter, unmodified
  * a TorchFunctionMode shim      rewrites device="cuda*" -> "cpu" for the pool /
                                  tree / metadata code that hard-codes CUDA
                                  (DeFT/deft/memory_pool.py:13-16,57-65,
                                   DeFT/deft/tree_decoding/tree_cache.py:813-857)
  * torch.cuda.synchronize = nop  GlobalTimer calls it (timer.py:16,24)

and replays the scripted trees of tests/scenarios.py on the reference's own
`TreeCache`, then calls `TreeMetadata.from_tree_cache`,
`tree_attention_subtree_fwd` and `tree_attention_fwd` exactly as
`DeFTAttention.deft_flatten_forward / deft_node_forward` do
(DeFT/deft/layers/attention/deft_attention.py:136-148, :94-105).

What is stored per fixture (data only — inputs are regenerated from seeds by
deft_amd.utils.synthetic, bit-identically on any machine):

  metadata int64 arrays + scalars, pool slot of every KV token per node,
  per-leaf root->leaf slot paths, and for kernel fixtures the reference's
  fp16 outputs `o_flatten`, `o_node` (and fp32 stage-1 partials for one small
  case).

Usage:  PYTHONDONTWRITEBYTECODE=1 TRITON_INTERPRET=1 python tools/gen_golden.py [--only NAME ...]
"""
from __future__ import annotations

import argparse
import os
import sys
import time

os.environ.setdefault("TRITON_INTERPRET", "1")
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.overrides import TorchFunctionMode  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/DeFT"
sys.path.insert(0, REF)

from scenarios import (  # noqa: E402
    FULL_GEOMETRY,
    GQA_GEOMETRY,
    SCENARIOS,
    SMALL_D_GEOMETRY,
    SMALL_GEOMETRIES,
    input_seeds,
)
from deft_amd.utils.synthetic import dyadic_normal  # noqa: E402


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        dev = kwargs.get("device")
        if dev is not None and str(dev).startswith("cuda"):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


torch.cuda.synchronize = lambda *a, **k: None  # type: ignore[assignment]

ARRAYS = (
    "node_q", "node_kv", "node_q_len", "node_kv_len", "node_q_offset", "node_kv_offset",
    "block_q", "block_q_cnts", "block_q_offset", "block_bitmasks", "block_kv", "block_lens",
)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)

    with CudaToCpu():
        from deft.memory_pool import ReqToTokenPool, TokenToKVPool
        from deft.tree_decoding import tree_cache as ref_tc
        import deft.layers.attention.tree_attention as ref_ta

        for name, sc in SCENARIOS.items():
            if args.only and name not in args.only:
                continue
            t0 = time.time()
            # layer_num=0: the reference pool would otherwise allocate KV storage we do not need here
            req_pool = ReqToTokenPool(size=128, max_context_len=sc.pool_size + 8)
            kv_pool = TokenToKVPool(size=sc.pool_size, dtype=torch.float16, head_num=1, head_dim=8, layer_num=0)
            tree = ref_tc.TreeCache(
                torch.float16, 1, 8, 1,
                req_to_token_pool=req_pool, token_to_kv_pool=kv_pool, tree_index_pool=None,
                use_paged_memory=True, use_tree_index=False,
            )
            sc.script(tree, lambda n: torch.arange(1, n + 1, dtype=torch.int32))
            ref_tc.BLOCK_CONFIG["BLOCK_LEN"] = sc.block_len
            ref_tc.BLOCK_CONFIG["MAX_BLOCK_LEN"] = -1
            md = ref_tc.TreeMetadata.from_tree_cache(tree, max_q_len=sc.max_q_len, max_block_len=sc.max_block_len)
            ref_tc.BLOCK_CONFIG["BLOCK_LEN"] = 128

            out = {k: getattr(md, k).numpy().astype(np.int64) for k in ARRAYS}
            out["scalars"] = np.asarray([md.query_num, md.node_num, md.total_kv_len, md.block_len], dtype=np.int64)
            # tree state pins: per-node slots (DFS-free: by node id) and leaf page-table rows
            node_ids = sorted(tree.nodes.keys())
            out["node_ids"] = np.asarray(node_ids, dtype=np.int64)
            out["node_kv_lens_by_id"] = np.asarray([len(tree.nodes[i].kv_indices) for i in node_ids], dtype=np.int64)
            out["node_kv_by_id"] = np.asarray([s for i in node_ids for s in tree.nodes[i].kv_indices], dtype=np.int64)
            out["pool_refcounts"] = kv_pool.mem_state.numpy().astype(np.int16)
            leaves = sorted(tree.leaves.values(), key=lambda n: n.id)
            out["leaf_ids"] = np.asarray([lf.id for lf in leaves], dtype=np.int64)

            geoms = []
            if sc.kernels:
                geoms += list(SMALL_GEOMETRIES)
            if name in FULL_GEOMETRY:
                geoms.append(FULL_GEOMETRY[name])
            if name in GQA_GEOMETRY:
                geoms.append(GQA_GEOMETRY[name])
            geoms += SMALL_D_GEOMETRY.get(name, [])
            for (Hq, Hkv, D) in geoms:
                seeds = input_seeds(name, (Hq, Hkv, D))
                nq = md.query_num
                q = torch.from_numpy(dyadic_normal((nq, Hq, D), seeds["q"]))
                kv = torch.from_numpy(dyadic_normal((sc.pool_size, 2, Hkv, D), seeds["kv"]))
                kbuf, vbuf = kv[:, 0], kv[:, 1]  # memory_pool.py:68-72
                tag = f"_{Hq}_{Hkv}_{D}"
                o = torch.zeros((nq, Hq, D), dtype=torch.float16)  # deft_attention.py:120
                ref_ta.tree_attention_subtree_fwd(
                    q, kbuf, vbuf, o, md.block_len, md.block_q, md.block_q_cnts, md.block_q_offset,
                    md.block_bitmasks, md.block_kv, md.block_lens,
                )
                out["o_flatten" + tag] = o.numpy().copy()
                o2 = torch.zeros((nq, Hq, D), dtype=torch.float16)
                ref_ta.tree_attention_fwd(
                    q, kbuf, vbuf, o2, md.node_kv, md.node_kv_offset, md.node_kv_len,
                    md.node_q, md.node_q_offset, md.node_q_len,
                )
                out["o_node" + tag] = o2.numpy().copy()
                if name == "cfgA_256x2" and (Hq, Hkv, D) == (4, 4, 128):
                    # stage-1 partials of the Node path, straight from the reference kernel
                    P = md.node_q.shape[0]
                    po = torch.zeros((Hq, P, D), dtype=torch.float32)
                    pl = torch.zeros((Hq, P), dtype=torch.float32)
                    ref_ta.DeFT_splitBynode_Triton_stage1(
                        q.transpose(0, 1), kbuf.transpose(0, 1), vbuf.transpose(0, 1),
                        md.node_kv, md.node_kv_offset, md.node_kv_len,
                        md.node_q, md.node_q_offset, md.node_q_len, P, po, pl,
                    )
                    out["node_partial_o" + tag] = po.numpy().copy()
                    out["node_partial_lse" + tag] = pl.numpy().copy()
            path = os.path.join(args.out, name + ".npz")
            np.savez_compressed(path, **out)
            print(f"{name}: nq={md.query_num} NB={len(out['block_lens'])} NE={md.node_num} "
                  f"geoms={geoms} -> {os.path.getsize(path) / 1024:.1f} KiB in {time.time() - t0:.1f}s", flush=True)


if __name__ == "__main__":
    main()
