#!/usr/bin/env python3
"""Golden vectors for the sequential comparator: tests/golden/seq_<scenario>.npz, made by running the REFERENCE's
`token_attention_fwd` (DeFT/deft/layers/attention/token_attention.py:297-335) on CPU under TRITON_INTERPRET=1, on the
page table of the reference's own TreeCache.  Build-container only, same shim as tools/gen_golden.py (the reference is
imported from /root/reference, never copied).  Stored: the page-table rows, request indices and sequence lengths the
operator was called with, and its fp16 output per geometry; q / kv inputs are regenerated from seeds.

Usage:  PYTHONDONTWRITEBYTECODE=1 TRITON_INTERPRET=1 python tools/gen_golden_seq.py [scenario ...]
"""
from __future__ import annotations

import os
import sys
import time

os.environ.setdefault("TRITON_INTERPRET", "1")
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, "/root/reference/DeFT")

from gen_golden import CudaToCpu  # noqa: E402  (also neutralises torch.cuda.synchronize)
from scenarios import SCENARIOS, input_seeds  # noqa: E402
from deft_amd.utils.synthetic import dyadic_normal  # noqa: E402

SEQ_CASES = {"cfgA_256x2": [(4, 4, 128), (8, 2, 128)], "multilevel": [(4, 4, 128), (8, 2, 128), (4, 4, 64)],
             "wide40": [(8, 2, 128)], "chain_300": [(4, 4, 128)],
             # the Medusa token tree itself (configs[2] read literally): 42 leaves at depths 1-4, each page-table row = prompt + its chain
             "medusa64_tree": [(8, 2, 128)]}


def main() -> None:
    out_dir = os.path.join(ROOT, "tests", "golden")
    with CudaToCpu():
        from deft.memory_pool import ReqToTokenPool, TokenToKVPool
        from deft.tree_decoding import tree_cache as ref_tc
        import deft.layers.attention.token_attention as ref_tok

        for name, geoms in SEQ_CASES.items():
            if len(sys.argv) > 1 and name not in sys.argv[1:]:
                continue
            sc = SCENARIOS[name]
            t0 = time.time()
            req_pool = ReqToTokenPool(size=128, max_context_len=sc.pool_size + 8)
            kv_pool = TokenToKVPool(size=sc.pool_size, dtype=torch.float16, head_num=1, head_dim=8, layer_num=0)
            tree = ref_tc.TreeCache(torch.float16, 1, 8, 1, req_to_token_pool=req_pool, token_to_kv_pool=kv_pool,
                                    tree_index_pool=None, use_paged_memory=True, use_tree_index=False)
            sc.script(tree, lambda n: torch.arange(1, n + 1, dtype=torch.int32))
            # what InputMetadata.from_tree hands to radix_attention_forward (model_runner.py:162-231)
            leaves = sorted(tree.leaves.values(), key=lambda n: n.id)
            reqs = [v for _, v in sorted(tree.leaf_to_req.items(), key=lambda x: x[0])]
            lens = []
            for lf in leaves:
                n, node = 0, lf
                while node is not None:
                    n += len(node.kv_indices)
                    node = node.parent
                lens.append(n)
            b_req_idx = torch.tensor(reqs, dtype=torch.int32)
            b_seq_len = torch.tensor(lens, dtype=torch.int32)
            b_start_loc = torch.zeros(len(lens), dtype=torch.int32)
            b_start_loc[1:] = torch.cumsum(b_seq_len[:-1], dim=0)
            total = int(b_seq_len.sum())
            table = req_pool.req_to_token
            out = {"b_req_idx": b_req_idx.numpy(), "b_seq_len": b_seq_len.numpy(), "b_start_loc": b_start_loc.numpy(),
                   "req_rows": np.stack([table[r, : max(lens)].numpy() for r in reqs]).astype(np.int32)}
            nq = len(lens)
            for (Hq, Hkv, D) in geoms:
                seeds = input_seeds(name, (Hq, Hkv, D))
                q = torch.from_numpy(dyadic_normal((nq, Hq, D), seeds["q"]))
                kv = torch.from_numpy(dyadic_normal((sc.pool_size, 2, Hkv, D), seeds["kv"]))
                o = torch.zeros((nq, Hq, D), dtype=torch.float16)
                att_m = torch.empty((Hq, total), dtype=torch.float16)
                ref_tok.token_attention_fwd(q, kv[:, 0], kv[:, 1], o, table, b_req_idx, b_start_loc, b_seq_len,
                                            int(max(lens)), None, total, att_m=att_m)
                out[f"o_seq_{Hq}_{Hkv}_{D}"] = o.numpy().copy()
            path = os.path.join(out_dir, "seq_" + name + ".npz")
            np.savez_compressed(path, **out)
            print(f"seq_{name}: nq={nq} total_tokens={total} geoms={geoms} -> {os.path.getsize(path) / 1024:.1f} KiB "
                  f"in {time.time() - t0:.1f}s", flush=True)


if __name__ == "__main__":
    main()
