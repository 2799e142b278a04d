"""Tree-template replay: drive the tree cache, the metadata builder and the attention operators through the
reference's three workloads (few-shot prompting, multi-step reasoning, speculative decoding) without a language
model — the caller of the hot path on the other side of `TreeMetadata`.

What is mirrored, and from where:

  tree templates                  DeFT/deft/data_loader.py:9-132, :181-235   `deft_amd/templates.py`: a template is a flat
                                  node table plus per-iteration event lists (`branch_at[iter] = [(node, children)]`,
                                  `prune_at[iter] = [nodes]`) -- what the branch controller consults each step
  branch_from_tree_template       DeFT/deft/tree_decoding/generation/branch_func_example.py:293-371
  branch_speculative_decoding     branch_func_example.py:374-442   (the reference's mock: all `tree_size` leaves are
                                  kept, the accepted tokens are squeezed into the root)
  branch_few_shot                 branch_func_example.py:12-62     SimpleTree: branch once after prefill, then greedy
  decode loop                     DeFT/deft/tree_decoding/generation/tree_generate.py:89-260

The model forward is replaced by synthetic q / k / v of the model's geometry and synthetic next-token scores (the
branch functions only use the ARG-TOP-K of the logits, never their values), so a replay exercises exactly the
tree-state transitions, page-table updates, per-step metadata builds and attention calls the reference performs, and
reports the reference's metrics (perf_metrics.py:98-116, :195-210): attention latency, decode latency of the replayed
part, TPOT = latency / generated tokens.

The dataset files themselves are not shipped (they belong to the reference repository); `read_reasoning_file(path)` /
`read_speculative_file(path)` read them where the user has them, and `synthetic_*` build templates of the same form.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional

import numpy as np
import torch

from .forward_mode import ForwardMode, InputMetadata, forward_mode_from_cli
from .memory_pool import ReqToTokenPool, TokenToKVPool
from .templates import (TreeTemplate, default_prompt_len, fit_accept_lengths, read_reasoning_file,  # noqa: F401
                        read_speculative_file, synthetic_beam_template, synthetic_few_shot_template,
                        synthetic_reasoning_template, synthetic_speculative_template)
from .tree_cache import TreeCache, TreeMetadata, register_tree_metadata

__all__ = [
    "TreeTemplate", "read_reasoning_file", "read_speculative_file", "fit_accept_lengths", "synthetic_reasoning_template",
    "synthetic_speculative_template", "synthetic_few_shot_template", "synthetic_beam_template", "branch_from_tree_template",
    "branch_speculative_decoding", "branch_few_shot", "ReplayReport", "TemplateReplay",
]


# ---------------------------------------------------------------------------
# branch functions (branch_func_example.py) on deft_amd.TreeCache
# ---------------------------------------------------------------------------
def _scores(logits) -> "np.ndarray":
    """The branch functions only use the arg-top-k of the scores.  numpy on purpose: a torch CPU op on a [32, vocab]
    tensor wakes the whole intra-op thread pool of a 256-core host and costs milliseconds per decode step."""
    if isinstance(logits, torch.Tensor):
        return logits.detach().float().cpu().numpy()
    return np.asarray(logits)


def _topk_ids(logits, row: int, k: int) -> List[int]:
    x = _scores(logits)[row]
    k = min(k, x.shape[-1])
    idx = np.argpartition(-x, k - 1)[:k]
    return idx[np.argsort(-x[idx], kind="stable")].tolist()


def _greedy(logits) -> List[int]:
    return _scores(logits).argmax(axis=1).tolist()


def branch_from_tree_template(tree: TreeCache, iter: int, max_gen_len: int, logits: torch.Tensor,
                              execution_graph: TreeTemplate) -> bool:
    """branch_func_example.py:293-371: a leaf that the template branches at this iteration branches (its children get
    the top-k tokens of its row), a leaf it releases is cut, every other leaf appends its argmax."""
    branch_pairs = dict(execution_graph.branch_at.get(iter, ()))
    prune_nodes = set(execution_graph.prune_at.get(iter, ()))
    stop = 0 in prune_nodes  # the root is released: the whole template has run (:315-319)
    if stop:
        for leaf in tree.leaves.values():
            tree.output_branch(dstnode=leaf)
    leaves = [tree.root] if iter == 0 else list(tree.leaves.values())
    greedy = _greedy(logits)
    for leaf in leaves:
        if leaf.id in branch_pairs:
            width = len(branch_pairs[leaf.id])
            assert width > 0
            q_idx = 0 if iter == 0 else tree.leaf_to_q[leaf.id]
            ids = _topk_ids(logits, q_idx, width)
            for j, child in enumerate(tree.branch(tree.nodes[leaf.id], width)):
                child.append_token(int(ids[j % len(ids)]))
        elif leaf.id in prune_nodes:
            tree.cut(tree.nodes[leaf.id], record_deleted=True)
        else:
            leaf.append_token(int(greedy[tree.leaf_to_q[leaf.id]]))
    if iter == max_gen_len - 1:  # (:365-369)
        for leaf in tree.leaves.values():
            tree.output_branch(dstnode=leaf)
        stop = True
    return stop


def branch_speculative_decoding(tree: TreeCache, iter: int, max_gen_len: int, logits: torch.Tensor,
                                execution_graph: TreeTemplate) -> bool:
    """branch_func_example.py:374-442, the reference's mock of Medusa-style verification: at iter 0 the root
    branches into `tree_size` one-token leaves; afterwards the first `Accept_length[iter]` leaves are merged into the
    root (their KV slots move to the root), every leaf's own KV is released and its positions shift."""
    accepted = execution_graph.accept_lengths
    assert accepted is not None
    if iter == len(accepted):  # (:387-397)
        for leaf in tree.leaves.values():
            tree.output_branch(dstnode=leaf)
        return True
    size = execution_graph.node_num
    if iter == 0:
        ids = _topk_ids(logits, 0, size)
        for j, leaf in enumerate(tree.branch(tree.root, size)):
            leaf.append_token(int(ids[j % len(ids)]))
        return False
    verified = accepted[iter]
    leaves = list(tree.leaves.values())
    assert len(leaves) == size
    before = len(tree.root.kv_indices)
    for i in range(min(verified, len(leaves))):
        tree.merge_nodes(tree.root, leaves[i], pruneB_flag=False)
    diff = len(tree.root.kv_indices) - before
    tree.reset_nodes_KV(leaves, diff)  # (= reset_node_KV leaf by leaf, :430-436; one refcount update for all of them)
    assert diff == min(verified, len(leaves))
    return False


def branch_few_shot(tree: TreeCache, iter: int, max_gen_len: int, logits: torch.Tensor, execution_graph: TreeTemplate) -> bool:
    """branch_func_example.py:12-62 (SimpleTree): branch into `width` leaves after the prefill, then greedy."""
    width = execution_graph.root_width
    if iter + 1 == max_gen_len:  # (:24-33: the last iteration finishes the branches and appends nothing)
        for leaf in tree.leaves.values():
            tree.output_branch(dstnode=leaf)
        return True
    # (ONE conversion of the scores per step, arg-max and top-k taken from it: with GPU probabilities -- tree_generate passes the
    #  [nq, vocab] softmax -- every further _scores() call was another device-to-host copy and host sync of the whole matrix.  ADVICE r5)
    x = _scores(logits)
    if iter == 0:
        ids = _topk_ids(x, 0, width)
        for j, leaf in enumerate(tree.branch(tree.root, width)):
            tok = int(ids[j % len(ids)])
            leaf.append_token(tok, logprob=float(np.log(x[0, tok])))  # (:36-48: the scores are probabilities)
    else:
        greedy = x.argmax(axis=1)
        for leaf in tree.leaves.values():
            row = tree.leaf_to_q[leaf.id]
            leaf.append_token(int(greedy[row]), logprob=float(np.log(x[row, greedy[row]])))
    return False


BRANCH_FUNCS: Dict[str, Callable[..., bool]] = {
    "reasoning": branch_from_tree_template, "speculative_decoding": branch_speculative_decoding,
    "few_shot": branch_few_shot,
}


# ---------------------------------------------------------------------------
# the decode loop (tree_generate.py:89-260) around the attention path
# ---------------------------------------------------------------------------
@dataclass
class ReplayReport:
    task: str
    mode: str
    steps: int = 0
    prompt_len: int = 0
    generated_tokens: int = 0  # TreeCache.get_tree_token_number() - prompt (tree_cache.py:569-584)
    decoded_rows: int = 0      # sum over steps of live leaves = query rows pushed through attention
    attention_ms: float = 0.0  # GPU time of the per-layer append + attention calls, all layers, all steps
    metadata_ms: float = 0.0   # host time of alloc + TreeMetadata.from_tree_cache (+ upload), all steps
    branch_ms: float = 0.0     # host time of the branch function
    kv_io_bytes: int = 0       # the reference's KV-IO counter: kv_len * Hq * D * 4 per layer call (perf_metrics.py:116-118;
    #                            kv_len = node slots / total_kv_len / total_num_tokens, deft_attention.py:88-91, :129-133, :168-171)
    mask_io_bytes: int = 0     # Flatten only: total_kv_len * 8 per layer call (perf_metrics.py:120-122)
    wall_ms: float = 0.0
    per_step: List[Dict[str, float]] = field(default_factory=list)

    def summary(self) -> Dict[str, Any]:
        gen = max(self.generated_tokens, 1)
        return {
            "task": self.task, "mode": self.mode, "steps": self.steps, "prompt_len": self.prompt_len,
            "generated_tokens": self.generated_tokens, "decoded_rows": self.decoded_rows,
            "attention_latency_ms": round(self.attention_ms, 3), "metadata_ms": round(self.metadata_ms, 3),
            "branch_ms": round(self.branch_ms, 3), "wall_ms": round(self.wall_ms, 3),
            "attention_TPOT_ms_per_token": round(self.attention_ms / gen, 5),  # perf_metrics.py:203-210 on attention latency
            "attention_us_per_step": round(self.attention_ms * 1e3 / max(self.steps, 1), 2),
            "KV_IO_TB": round(self.kv_io_bytes / 1e12, 4), "Mask_IO_GB": round(self.mask_io_bytes / 1e9, 4),
            "max_live_leaves": int(max((s["nq"] for s in self.per_step), default=0)),
            "max_tree_kv_tokens": int(max((s["kv_tokens"] for s in self.per_step), default=0)),
        }


class TemplateReplay:
    """One decoding tree replayed step by step.  `attention=False` runs the tree / page-table / metadata work only
    (no GPU needed); with attention every layer of every step goes through DeFTAttention exactly as the model's
    layers would call it (llama2.py:108-113)."""

    def __init__(self, num_heads: int, num_kv_heads: int, head_dim: int, layers: int, mode: str = "flatten",
                 device: str = "cuda", attention: bool = True, seed: int = 0, vocab: int = 4096, session: Optional[bool] = None,
                 capture_after="auto", incremental: bool = True, win_tiles: Optional[int] = None) -> None:
        """`session`: drive the attention path through `deft_amd.DecodeSession` -- the whole decode step (tree advance,
        TreeMetadata, plan, every layer's append + attention) as ONE captured hipGraph per structural epoch of the tree -- instead
        of the reference-shaped eager calls (`tree.alloc()`, `TreeMetadata.from_tree_cache`, `DeFTAttention.forward` per layer).
        None = wherever a session exists (DeFT-Flatten / DeFT-Node, head_dim 128 or 64 as head pairs, attention on).
        `capture_after`: DecodeSession's -- how many steps of an epoch run eagerly before its step is captured.
        `incremental`: DecodeSession's -- window plans (most steps patch the plan instead of rebuilding metadata and plan)."""
        self.capture_after, self.incremental, self.win_tiles = capture_after, incremental, win_tiles
        self.Hq, self.Hkv, self.D, self.layers = num_heads, num_kv_heads, head_dim, layers
        self.mode = mode
        # (node / node_chunk set BLOCK_CONFIG["MAX_BLOCK_LEN"] as the CLI does -- a process-wide setting.  The replay keeps ITS value
        #  and puts it in force for the duration of run() only: two replays of different modes built before either runs no longer
        #  overwrite each other's chunking, and a caller's own setting survives the construction of a replay.  ADVICE r5)
        from .tree_cache import BLOCK_CONFIG

        before = BLOCK_CONFIG["MAX_BLOCK_LEN"]
        self.forward_mode: ForwardMode = forward_mode_from_cli(mode)
        self.max_block_len = BLOCK_CONFIG["MAX_BLOCK_LEN"] if mode in ("node", "node_chunk", "deft_node", "deft_node_chunk") else -1
        BLOCK_CONFIG["MAX_BLOCK_LEN"] = before
        self.device = device
        self.attention = attention
        self.vocab = vocab
        self.rng = np.random.default_rng(seed)
        # synthetic next-token scores: the branch functions use their arg-top-k only.  A table drawn ONCE and read through a
        # rolling window -- drawing nq x vocab fresh numbers per decode step cost more host time than the step itself
        self._score_table = self.rng.random((1024 + 257, vocab), dtype=np.float32)
        self._score_at = 0
        can = attention and mode in ("flatten", "node", "node_chunk") and (head_dim == 128 or (head_dim == 64 and num_kv_heads % 2 == 0))
        self.session = can if session is None else (bool(session) and can)
        # test hook: called after every step's attention with (tree, layer-0 q rows [nq, Hq*D], layer-0 output [nq, Hq*D])
        self.step_hook: Optional[Callable[[TreeCache, torch.Tensor, torch.Tensor], None]] = None
        self.trace_hook: Optional[Callable[..., None]] = None  # see run()
        if attention:
            from .deft_attention import DeFTAttention

            self.attn = [DeFTAttention(num_heads, head_dim, head_dim ** -0.5, num_kv_heads, l) for l in range(layers)]

    def _scores(self, rows: int) -> np.ndarray:
        tab = self._score_table
        if rows > tab.shape[0]:
            return self.rng.random((rows, self.vocab), dtype=np.float32)
        self._score_at = (self._score_at + 257) % (tab.shape[0] - rows + 1)  # (a different window every step)
        return tab[self._score_at : self._score_at + rows]

    def _pools(self, max_tokens: int, max_leaves: int):
        req = ReqToTokenPool(max_leaves + 8, max_tokens + 8, device=self.device)
        pool = TokenToKVPool(max_tokens, torch.float16, self.Hkv, self.D, self.layers if self.attention else 0,
                             device=self.device)
        return req, pool

    def run(self, *args, **kw) -> ReplayReport:
        """`_run` under this replay's own MAX_BLOCK_LEN (the global is restored afterwards, whatever happens)."""
        from .tree_cache import BLOCK_CONFIG

        before = BLOCK_CONFIG["MAX_BLOCK_LEN"]
        BLOCK_CONFIG["MAX_BLOCK_LEN"] = self.max_block_len
        try:
            return self._run(*args, **kw)
        finally:
            BLOCK_CONFIG["MAX_BLOCK_LEN"] = before

    def _run(self, template: TreeTemplate, task: str, prompt_len: int, max_gen_len: int, max_tokens: Optional[int] = None,
             max_leaves: int = 512, max_rows: int = 512, pipelined: bool = False,
             scores_fn: Optional[Callable[[int, int], np.ndarray]] = None) -> ReplayReport:
        """`pipelined=True`: no per-step synchronisation -- the host builds step t+1's tree state and metadata while the
        GPU still runs step t (the path is launch-only; the synthetic scores do not depend on the GPU's output, as
        in a real engine between branch events, where the tree's SHAPE one step ahead is known).  Per-step attention
        times are then not available; `attention_ms` is the GPU span of the whole replay and `wall_ms` what a caller
        would see.

        `scores_fn(iter, rows)`: the next-token scores handed to the branch function at iteration `iter` (default: the
        replay's own random table).  `self.trace_hook(iter, tree, cache_loc, metadata, session)`, when set, is called once per
        decode step after the step's slots and metadata exist and before the branch function runs -- tests/test_replay_golden.py
        compares what it sees with the reference's own loop, step for step."""
        branch = BRANCH_FUNCS[task]
        score = scores_fn if scores_fn is not None else (lambda _it, rows: self._scores(rows))
        if task == "speculative_decoding":
            max_gen_len = min(max_gen_len, len(template.accept_lengths or []) + 1)
        if max_tokens is None:
            budget = template.token_budget()
            if task == "speculative_decoding":
                budget = sum(template.accept_lengths or []) + 2 * template.node_num * 2
            if task == "few_shot":
                budget = template.root_width * max_gen_len
            max_tokens = prompt_len + budget + max_leaves + 1024
        req, pool = self._pools(max_tokens, max_leaves)
        tree = TreeCache(torch.float16, self.Hkv, self.D, self.layers, req, pool, None, True, False)
        rep = ReplayReport(task=task, mode=self.mode, prompt_len=prompt_len)
        dev = self.device
        if self.attention:
            g = torch.Generator(device=dev)
            g.manual_seed(1)
            for l in range(self.layers):  # prefill stand-in: the prompt's KV is whatever the pool holds
                pool._storage[l].normal_(generator=g)
            q_all = torch.randn((self.layers, max_rows, self.Hq * self.D), dtype=torch.float16, device=dev, generator=g)
            k_all = torch.randn((self.layers, max_rows, self.Hkv * self.D), dtype=torch.float16, device=dev, generator=g)
            v_all = torch.randn((self.layers, max_rows, self.Hkv * self.D), dtype=torch.float16, device=dev, generator=g)
        sess = None
        nq_now = [1]
        if self.session:
            from ._lib import check, lib
            from .session import DecodeSession
            from .tree_cache import _ptr

            sess = DecodeSession(tree, self.Hq, self.Hkv, self.D, self.layers,
                                 lambda l: (q_all[l, : nq_now[0]], k_all[l, : nq_now[0]], v_all[l, : nq_now[0]]),
                                 mode="node" if self.mode == "node_chunk" else self.mode,
                                 capture_after=self.capture_after, incremental=self.incremental, win_tiles=self.win_tiles)
            sizes = np.zeros(9, dtype=np.int64)
        t_wall = time.perf_counter()
        tree.init_prompt(torch.arange(1, prompt_len + 1, dtype=torch.int32))
        logits = score(0, 1)
        stop = branch(tree, 0, max_gen_len, logits, template)  # tree_generate.py:188-197
        it = 1
        first_event = last_event = None
        while not stop and it < max_gen_len:
            # ---- prepare: positions, KV slots for this step's tokens, metadata (tree_generate.py:93-131) -------
            t0 = time.perf_counter()
            leaves = sorted(tree.leaves.values(), key=lambda x: x.id)
            nq = len(leaves)
            if nq == 0:
                break
            assert nq <= max_rows, f"{nq} live leaves exceed max_rows={max_rows}"
            tree.leaf_to_q = {leaf.id: i for i, leaf in enumerate(leaves)}
            if sess is not None:
                # ---- the captured step: slots from the host allocator, everything else on the GPU ----------------------------
                nq_now[0] = nq
                # (pipelined: ONE timing event in front of the first step and one behind the last -- two events per step between the
                #  captured graphs cost the loop ~12 us of idle queue per step, tools/step_boundary.py)
                if not pipelined or first_event is None:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                    if pipelined:
                        first_event = e0
                outs = sess.step()
                if not pipelined:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                if self.trace_hook is not None:
                    self.trace_hook(it, tree, None, None, sess)
                last_event = None if pipelined else e1
                if self.step_hook is not None:
                    self.step_hook(tree, q_all[0, :nq], outs[0][:nq])
                t_md = (time.perf_counter() - t0) * 1e3  # (host time of the whole step call: allocator, staging, graph launch)
                t_attn = 0.0
                if not pipelined:
                    e1.synchronize()
                    t_attn = e0.elapsed_time(e1)
                # sizes for the reference's IO counters, from node lengths on the host (no device read)
                mq, bl, mbl = sess.dt.cfg
                check(lib.deft_tree_md_sizes(tree._native, mq, bl, mbl, 0, _ptr(sizes)), "deft_tree_md_sizes")
                kv_tokens, node_kv_n = int(sizes[2]), int(sizes[4])
                t1 = time.perf_counter()
                logits = score(it, nq)
                stop = branch(tree, it, max_gen_len, logits, template)
                t_br = (time.perf_counter() - t1) * 1e3
                rep.per_step.append({"iter": it, "nq": nq, "kv_tokens": kv_tokens, "attention_ms": t_attn,
                                     "metadata_ms": t_md, "branch_ms": t_br})
                if self.forward_mode == ForwardMode.TREE_DECODE_FLATTEN:
                    io_len = kv_tokens
                    rep.mask_io_bytes += io_len * 8 * self.layers
                else:
                    io_len = node_kv_n
                rep.kv_io_bytes += io_len * self.Hq * self.D * 4 * self.layers
                rep.steps += 1
                rep.decoded_rows += nq
                rep.attention_ms += t_attn
                rep.metadata_ms += t_md
                rep.branch_ms += t_br
                it += 1
                continue
            updater = tree.alloc()
            md = None
            if self.forward_mode != ForwardMode.DECODE:
                md = TreeMetadata.from_tree_cache(tree, device=dev)
                register_tree_metadata(md)
                meta = InputMetadata(self.forward_mode, updater, pool)
            else:
                positions = torch.tensor([lf.positions[-1] for lf in leaves], dtype=torch.int64, device=dev)
                meta = InputMetadata.from_tree(tree, req, pool, self.forward_mode, positions, updater)
            t_md = (time.perf_counter() - t0) * 1e3
            if self.trace_hook is not None:
                self.trace_hook(it, tree, updater.cache_loc, md, None)
            # ---- forward: the attention path of every layer ---------------------------------------------------
            t_attn = 0.0
            if self.attention:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if pipelined and first_event is None:
                    first_event = e0
                for l in range(self.layers):
                    o_l = self.attn[l](q_all[l, :nq], k_all[l, :nq], v_all[l, :nq], meta)
                    if l == 0 and self.step_hook is not None:
                        self.step_hook(tree, q_all[0, :nq], o_l)
                e1.record()
                last_event = e1
                if not pipelined:
                    e1.synchronize()
                    t_attn = e0.elapsed_time(e1)
            # ---- branch (tree_generate.py:226-236) ------------------------------------------------------------
            t1 = time.perf_counter()
            logits = score(it, nq)
            kv_tokens = md.total_kv_len if md is not None else sum(len(n.kv_indices) for n in tree.nodes.values())
            stop = branch(tree, it, max_gen_len, logits, template)
            t_br = (time.perf_counter() - t1) * 1e3
            rep.per_step.append({"iter": it, "nq": nq, "kv_tokens": int(kv_tokens), "attention_ms": t_attn,
                                 "metadata_ms": t_md, "branch_ms": t_br})
            if self.forward_mode == ForwardMode.TREE_DECODE_FLATTEN:
                io_len = int(md.total_kv_len)
                rep.mask_io_bytes += io_len * 8 * self.layers
            elif self.forward_mode == ForwardMode.TREE_DECODE_NODE:
                io_len = int(md.node_kv.shape[0])
            else:
                io_len = int(meta.total_num_tokens)
            rep.kv_io_bytes += io_len * self.Hq * self.D * 4 * self.layers
            rep.steps += 1
            rep.decoded_rows += nq
            rep.attention_ms += t_attn
            rep.metadata_ms += t_md
            rep.branch_ms += t_br
            it += 1
        if pipelined and first_event is not None:
            if last_event is None:  # (the session path records nothing per step)
                last_event = torch.cuda.Event(enable_timing=True)
                last_event.record()
            last_event.synchronize()
            rep.attention_ms = first_event.elapsed_time(last_event)
        rep.wall_ms = (time.perf_counter() - t_wall) * 1e3
        rep.generated_tokens = tree.get_tree_token_number() - prompt_len
        self.tree, self.pool, self.req = tree, pool, req  # left for inspection by tests
        self.graph_captures = sess.captures if sess is not None else None
        self.step_kinds = dict(sess.step_kinds) if sess is not None else None  # (DecodeSession: upload / legacy / replan / patch steps)
        return rep
