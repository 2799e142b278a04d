#!/usr/bin/env python3
"""The drop-in claim executed: the REFERENCE's own driver code -- `tree_generate.tree_generate`
(DeFT/deft/tree_decoding/generation/tree_generate.py:20-275), its `Branch_Controller`, its three branch functions, its template
loader AND its `TreeMetadata.from_tree_cache` -- run UNCHANGED (imported from /root/reference, build container only) on
deft_amd's `TreeCache` / `TreeNode` / `ReqToTokenPool` / `TokenToKVPool` objects, with a stub in place of the Llama model
(`forward_prefill` / `forward_tree_decode` return the scores the golden runs were made with).  At every decode step what the
reference's code sees and builds ON THE PRODUCT'S OBJECTS -- the slots `alloc()` hands out, the leaves, tokens, positions and the
twelve arrays of the reference's own metadata builder walking deft_amd's tree -- must equal what the same code saw on the
reference's own objects (tests/golden/replay_*.npz, tools/gen_golden_replay.py).

So this checks the class surface by USE, attribute by attribute (`node.children`, `node.refs`, `node.kv_indices`,
`tree.leaves` order, `tree.leaf_to_req`, `pool.alloc` / `free` / `add_refs`, `req_to_token` ...), where tests/test_surface.py checks
signatures.  tests/test_dropin_reference_driver.py runs it where /root/reference exists (this container) and skips elsewhere.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/check_dropin_with_reference_driver.py [NAME ...]
"""
from __future__ import annotations

import contextlib
import io
import os
import random
import sys
import types

sys.dont_write_bytecode = True
os.environ.setdefault("TRITON_INTERPRET", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.overrides import TorchFunctionMode  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/DeFT"
BASE = "/root/reference/dataset/generation"
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

ARRAYS = ("node_q", "node_kv", "node_q_len", "node_kv_len", "node_q_offset", "node_kv_offset",
          "block_q", "block_q_cnts", "block_q_offset", "block_bitmasks", "block_kv", "block_lens")
RUNS = {  # name -> (the reference's branch function, its template source)
    "simple_w6": ("example_branch_Func1_SimpleTree", None),
    "keywordToT": ("example_branch_Func3_FromTreeTemplate", ("Reasoning", "keywordToT")),
    "set128ToT": ("example_branch_Func3_FromTreeTemplate", ("Reasoning", "set128ToT")),
    "speculative64": ("example_branch_Func4_SpeculativeDecoding", ("Speculative_Decoding", "tree_size64")),
    "speculative256": ("example_branch_Func4_SpeculativeDecoding", ("Speculative_Decoding", "tree_size256")),
    # (the two long templates: 2375 and 3708 decode steps -- run by the tool, not by the test)
    "docmergeToT": ("example_branch_Func3_FromTreeTemplate", ("Reasoning", "docmergeToT")),
    "sorting128ToT": ("example_branch_Func3_FromTreeTemplate", ("Reasoning", "sorting128ToT")),
}
QUICK = ("simple_w6", "keywordToT", "set128ToT", "speculative64", "speculative256")


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        dev = kwargs.get("device")
        if dev is not None and str(dev).startswith("cuda"):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


def available() -> bool:
    return os.path.isdir(REF) and os.path.isdir(BASE)


def run(names=None, verbose=True):
    import deft_amd
    from deft_amd.utils.synthetic import permutation_scores
    from test_replay_golden import GOLD_DIR, _Checker  # the same comparisons the product's own replay is held to

    sys.path.insert(0, REF)
    stub = types.ModuleType("deft.model_runner")  # (the real one imports the Llama layers; the driver only names these two)
    stub.ModelRunner = type("ModelRunner", (), {})
    stub.ForwardMode = deft_amd.ForwardMode
    saved = {k: sys.modules.get(k) for k in ("deft.model_runner",)}
    sys.modules["deft.model_runner"] = stub
    real_sync, real_cuda = torch.cuda.synchronize, torch.Tensor.cuda
    torch.cuda.synchronize = lambda *a, **k: None  # GlobalTimer (timer.py:16, :24)
    torch.Tensor.cuda = lambda self, *a, **k: self  # tree_generate.py:135-136
    done = []
    try:
        with CudaToCpu():
            from deft import data_loader as ref_dl
            from deft.tree_decoding import tree_cache as ref_tc
            from deft.tree_decoding.branch_controller import Branch_Controller
            from deft.tree_decoding.generation import branch_func_example as ref_bf
            from deft.tree_decoding.generation import tree_generate as ref_tg
            from deft.tree_decoding.perf_metrics import PerfMetrics

            for name, (fn_name, src) in RUNS.items():
                if names and name not in names:
                    continue
                g = np.load(os.path.join(GOLD_DIR, f"replay_{name}.npz"))
                prompt_len, max_gen_len, pool_size, width, vocab = (int(x) for x in g["config"])
                template = None
                if src is not None and src[0] == "Reasoning":
                    dataset = ref_dl.load_dataset(os.path.join(BASE, src[0], src[1] + ".json"))
                    template = ref_dl.build_trees([next(it for it in dataset if not it.get("incompleted"))])[0]
                elif src is not None:
                    random.seed(0)
                    template = ref_dl.load_prompts(os.path.join(BASE, src[0], src[1] + ".json"))[0]
                    ref_dl.generate_accepted_len_list(max_gen_len=max_gen_len, tree=template)
                # deft_amd's objects under the reference's code
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
                        assert kv_updater.cache_loc.tolist() == list(range(prompt_len))  # init_prompt's slots
                        return torch.log(torch.from_numpy(permutation_scores(0, 1, vocab))), None

                    def forward_tree_decode(self, forward_mode, token_ids, positions, kv_updater, flag, tree_metadata):
                        self.it += 1
                        assert isinstance(tree_metadata, ref_tc.TreeMetadata)  # built by the REFERENCE's builder on deft_amd's tree
                        leaves = sorted(tree.leaves.values(), key=lambda n: n.id)
                        assert token_ids.tolist() == [lf.token_ids[-1] for lf in leaves]
                        assert positions.tolist() == [lf.positions[-1] for lf in leaves]
                        chk(self.it, tree, np.asarray(kv_updater.cache_loc), {a: getattr(tree_metadata, a).numpy() for a in ARRAYS},
                            np.asarray(pool.mem_state))
                        return (torch.log(torch.from_numpy(permutation_scores(self.it, len(leaves), vocab))),), 0.0

                model = StubModel()
                real_free = tree.free

                def free_after_recording():  # tree_generate.py:274 frees the tree: look at its end state first
                    end.update(nodes=len(tree.nodes), leaves=len(tree.leaves), used=int((np.asarray(pool.mem_state) != 0).sum()),
                               tokens=tree.get_tree_token_number(), finished=len(tree.all_finished_seqs))
                    real_free()

                tree.free = free_after_recording
                ctl = Branch_Controller(branching_function=getattr(ref_bf, fn_name))
                prompt_ids = torch.arange(1, prompt_len + 1, dtype=torch.int32).reshape(1, -1)
                with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                    ref_tg.tree_generate(model=model, mode=deft_amd.ForwardMode.TREE_DECODE_FLATTEN, tokenizer=None, prompt_ids=prompt_ids,
                                         max_seq_len=prompt_len + max_gen_len, width=width, depth=0, branch_controller=ctl,
                                         tree_template=template, output_file=None, perf_metrics=PerfMetrics(None))
                assert chk.k == len(g["iter"]) and chk.snaps_seen == len(chk.snaps), (chk.k, len(g["iter"]))
                nodes, leaves_n, used, tokens, finished = (int(x) for x in g["end_state"])
                assert (end["nodes"], end["leaves"], end["used"], end["tokens"], end["finished"]) == (nodes, leaves_n, used, tokens, finished), end
                done.append((name, chk.k))
                if verbose:
                    print(f"{name}: the reference's tree_generate + {fn_name} + from_tree_cache on deft_amd objects: {chk.k} decode steps "
                          f"identical to the reference's own run", flush=True)
    finally:
        torch.cuda.synchronize, torch.Tensor.cuda = real_sync, real_cuda
        if REF in sys.path:
            sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return done


if __name__ == "__main__":
    if not available():
        raise SystemExit("needs /root/reference (build container only)")
    run(sys.argv[1:] or None)
