#!/usr/bin/env python3
"""Golden vectors for the DECODE LOOP: tests/golden/replay_*.npz, made by running the REFERENCE's own
branch functions on the REFERENCE's own TreeCache in the build container.

What runs (all of it imported from /root/reference, nothing copied):

  example_branch_Func1_SimpleTree            DeFT/deft/tree_decoding/generation/branch_func_example.py:12-62
  example_branch_Func3_FromTreeTemplate      :293-372   on the first complete tree of docmergeToT.json / sorting128ToT.json
  example_branch_Func4_SpeculativeDecoding   :374-442   on record 0 of Speculative_Decoding/tree_size64.json, its accepted
                                                        lengths fitted by data_loader.generate_accepted_len_list under
                                                        random.seed(0) (run_DeFT_llama_paged.py:32, :258-263)
  the loop around them                        tree_generate.py:92-169, :171-236: init_prompt, branch(iter 0), then per
                                              iteration leaf_to_q (leaves by id), tree.alloc(), TreeMetadata.from_tree_cache,
                                              [the model's forward], the branch function on softmax-like scores

The model is a stub with a `.tree` (the branch functions use nothing else; `deft.model_runner` is stubbed because it imports
the Llama layers).  The scores are deft_amd.utils.synthetic.permutation_scores(iter, rows): tie-free, integer-derived.

Stored per workload (data only):
  per step      nq, the step's cache_loc (slots handed out by tree.alloc()), the live leaves in id order AND in the order
                `tree.leaves` iterates (what the branch functions walk), every leaf's last token id and position, node count
  snapshots     (the first few steps, then every n-th, the step of and the step after every branch event and the first three
                prune events of a template, the last)
                the twelve TreeMetadata arrays + scalars and the pool's reference counts
  every step    64-bit digests of the twelve arrays, of the six node_* and of the six block_* arrays (so every step's metadata
                is pinned; the snapshots say where a mismatch is)

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_replay.py [--only NAME ...]
"""
from __future__ import annotations

import argparse
import hashlib
import os
import random
import sys
import time
import types

os.environ.setdefault("TRITON_INTERPRET", "1")
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.overrides import TorchFunctionMode  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/DeFT")

from deft_amd.utils.synthetic import permutation_scores  # noqa: E402

ARRAYS = ("node_q", "node_kv", "node_q_len", "node_kv_len", "node_q_offset", "node_kv_offset",
          "block_q", "block_q_cnts", "block_q_offset", "block_bitmasks", "block_kv", "block_lens")
BASE = "/root/reference/dataset/generation"
VOCAB = 4096


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        dev = kwargs.get("device")
        if dev is not None and str(dev).startswith("cuda"):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


torch.cuda.synchronize = lambda *a, **k: None  # type: ignore[assignment]

# `deft.model_runner` pulls in the Llama layers (vllm-style imports that do not exist here); the branch functions only
# name its ModelRunner in annotations
stub = types.ModuleType("deft.model_runner")
stub.ModelRunner = type("ModelRunner", (), {})
stub.ForwardMode = type("ForwardMode", (), {})
sys.modules["deft.model_runner"] = stub


def md_digest(md, fields=ARRAYS) -> np.uint64:
    h = hashlib.sha256()
    for k in fields:
        a = np.ascontiguousarray(getattr(md, k).numpy().astype(np.int64))
        h.update(np.int64(a.size).tobytes())
        h.update(a.tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


# name -> (branch function name, template source, prompt_len, max_gen_len, pool size, dense snapshots, sparse stride)
WORKLOADS = {
    "simple_w6": ("example_branch_Func1_SimpleTree", ("width", 6), 300, 40, 2048, 6, 8),
    "simple_4kx32": ("example_branch_Func1_SimpleTree", ("width", 32), 4096, 24, 8192, 2, 8),
    "docmergeToT": ("example_branch_Func3_FromTreeTemplate", ("reasoning", "docmergeToT"), None, 100000, 32768, 3, 800),
    "sorting128ToT": ("example_branch_Func3_FromTreeTemplate", ("reasoning", "sorting128ToT"), None, 100000, 65536, 3, 1200),
    # the other two shipped reasoning templates (61 and 91 lifetime nodes, a branch every ~94 / ~36 steps)
    "keywordToT": ("example_branch_Func3_FromTreeTemplate", ("reasoning", "keywordToT"), None, 100000, 16384, 3, 150),
    "set128ToT": ("example_branch_Func3_FromTreeTemplate", ("reasoning", "set128ToT"), None, 100000, 16384, 3, 100),
    "speculative64": ("example_branch_Func4_SpeculativeDecoding", ("speculative", "tree_size64"), 1016, 120, 8192, 6, 10),
    # 256 candidates: more than max_q_len queries below one node -- the root's blocks alternate between eight query chunks
    "speculative256": ("example_branch_Func4_SpeculativeDecoding", ("speculative", "tree_size256"), 1016, 60, 8192, 3, 8),
}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    with CudaToCpu():
        from deft import data_loader as ref_dl
        from deft.memory_pool import ReqToTokenPool, TokenToKVPool
        from deft.tree_decoding import tree_cache as ref_tc
        from deft.tree_decoding.branch_controller import Branch_Controller
        from deft.tree_decoding.generation import branch_func_example as ref_bf

        for name, (fn_name, src, prompt_len, max_gen_len, pool_size, dense, stride) in WORKLOADS.items():
            if args.only and name not in args.only:
                continue
            t0 = time.time()
            width = 0
            template = None
            extra = {}
            if src[0] == "width":
                width = src[1]
            elif src[0] == "reasoning":
                dataset = ref_dl.load_dataset(os.path.join(BASE, "Reasoning", src[1] + ".json"))
                item = next(it for it in dataset if not it.get("incompleted"))
                template = ref_dl.build_trees([item])[0]
                prompt_len = int(template.root.value)  # the root node of a Reasoning tree is the prompt
            else:
                random.seed(0)  # run_DeFT_llama_paged.py:32
                template = ref_dl.load_prompts(os.path.join(BASE, "Speculative_Decoding", src[1] + ".json"))[0]
                ref_dl.generate_accepted_len_list(max_gen_len=max_gen_len, tree=template)  # run_DeFT_llama_paged.py:258-263
                extra["accept_lengths"] = np.asarray(template.accepted_len_list, dtype=np.int64)
                extra["tree_size"] = np.asarray([template.node_num], dtype=np.int64)
            req_pool = ReqToTokenPool(size=256, max_context_len=pool_size)
            kv_pool = TokenToKVPool(size=pool_size, dtype=torch.float16, head_num=1, head_dim=8, layer_num=0)
            tree = ref_tc.TreeCache(torch.float16, 1, 8, 1, req_to_token_pool=req_pool, token_to_kv_pool=kv_pool,
                                    tree_index_pool=None, use_paged_memory=True, use_tree_index=False)
            model = types.SimpleNamespace(tree=tree, use_paged_memory=True)
            ctl = Branch_Controller(branching_function=getattr(ref_bf, fn_name))
            ctl.set_execution_graph(tree_templates=template)

            def branch(it, prob):
                return ctl.apply_branching(model=model, iter=it, max_gen_len=max_gen_len, width=width, depth=0,
                                           logits=torch.from_numpy(prob), execution_graph=ctl.tree_templates)

            events = set()
            if template is not None and src[0] == "reasoning":
                events = set(template.branch_record) | set(sorted(template.prune_record)[:3])
            tree.init_prompt(torch.arange(1, prompt_len + 1, dtype=torch.int32))  # tree_generate.py:55
            stop = branch(0, permutation_scores(0, 1, VOCAB))  # :188-197
            steps = {k: [] for k in ("iter", "nq", "node_cnt", "digest", "digest_node", "digest_block")}
            cat = {k: [] for k in ("cache_loc", "leaf_ids", "leaf_walk", "last_token", "last_pos")}
            snaps = {}
            it = 1
            while not stop and it < max_gen_len:  # tree_generate.py:200-
                leaves = sorted(tree.leaves.values(), key=lambda x: x.id)  # :97-99
                if not leaves:
                    break
                leaf_to_q = {lf.id: i for i, lf in enumerate(leaves)}
                upd = tree.alloc()  # :106
                md = ref_tc.TreeMetadata.from_tree_cache(tree)  # :124
                tree.leaf_to_q = leaf_to_q  # :219
                nq = len(leaves)
                steps["iter"].append(it)
                steps["nq"].append(nq)
                steps["node_cnt"].append(len(tree.nodes))
                steps["digest"].append(md_digest(md))
                steps["digest_node"].append(md_digest(md, ARRAYS[:6]))   # what DeFT-Node reads (a session in that mode builds only these)
                steps["digest_block"].append(md_digest(md, ARRAYS[6:]))  # what DeFT-Flatten reads
                cat["cache_loc"] += upd.cache_loc.tolist()
                cat["leaf_ids"] += [lf.id for lf in leaves]
                cat["leaf_walk"] += list(tree.leaves.keys())
                cat["last_token"] += [lf.token_ids[-1] for lf in leaves]
                cat["last_pos"] += [lf.positions[-1] for lf in leaves]
                k = len(steps["iter"])
                near_event = any((it - d) in events for d in (0, 1))
                if k <= dense or k % stride == 0 or near_event:
                    snaps[it] = md
                    snaps[it]._refc = kv_pool.mem_state.numpy().astype(np.int16).copy()
                last = (it, md, kv_pool.mem_state.numpy().astype(np.int16).copy())
                stop = branch(it, permutation_scores(it, nq, VOCAB))
                it += 1
            if last[0] not in snaps:
                snaps[last[0]] = last[1]
                snaps[last[0]]._refc = last[2]
            out = {k: np.asarray(v, dtype=np.uint64 if k.startswith("digest") else np.int64) for k, v in steps.items()}
            out.update({k: np.asarray(v, dtype=np.int64) for k, v in cat.items()})
            out.update(extra)
            out["config"] = np.asarray([prompt_len, max_gen_len, pool_size, width, VOCAB], dtype=np.int64)
            out["snap_iters"] = np.asarray(sorted(snaps), dtype=np.int64)
            for s_it, md in snaps.items():
                for a in ARRAYS:
                    out[f"s{s_it}_{a}"] = getattr(md, a).numpy().astype(np.int64)
                out[f"s{s_it}_scalars"] = np.asarray([md.query_num, md.node_num, md.total_kv_len, md.block_len], dtype=np.int64)
                used = np.flatnonzero(md._refc)
                out[f"s{s_it}_ref_slots"] = used.astype(np.int32)
                out[f"s{s_it}_ref_counts"] = md._refc[used]
            out["end_state"] = np.asarray([len(tree.nodes), len(tree.leaves), int((kv_pool.mem_state != 0).sum()),
                                           tree.get_tree_token_number(), len(tree.all_finished_seqs)], dtype=np.int64)
            path = os.path.join(args.out, f"replay_{name}.npz")
            np.savez_compressed(path, **out)
            print(f"{name}: {len(steps['iter'])} steps, {len(snaps)} snapshots, max nq {max(steps['nq'])}, end {out['end_state'].tolist()} "
                  f"-> {os.path.getsize(path) / 1024:.1f} KiB in {time.time() - t0:.1f}s", flush=True)


if __name__ == "__main__":
    main()
