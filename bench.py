#!/usr/bin/env python3
"""Benchmark of the DeFT tree-attention decode path on MI355X.

A "step" is ONE tree decode step of the hot path over all layers of the model:
for each of the 32 layers, the paged KV append of the new token rows followed by
the DeFT-Flatten (or DeFT-Node) attention operator, each layer on its own KV pool
(working set = layers x tree KV, far beyond the 256 MB Infinity Cache).  Inputs are
resident in HBM when the timed region starts; the host-side metadata build is a
caller of the path, reported separately (`metadata_build_ms`), not timed, and so is its
device-side repack, once per step for all layers (`plan_build_us_per_step`).

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on
rank 0.  For N > 1 launch with torch.distributed.run (one rank per GPU; `--gpus N` outside a
launcher starts one); every rank decodes its own independent tree (weak scaling, no data-path
collective), and BASELINE configs[4] -- the 8 N-tree forest sharded 8 per GPU by `shard_trees`
-- is measured in the same run (`cfg5_sharded_forest`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import deft_amd  # noqa: E402  (fails loudly if libdeft_amd.so is missing)
from deft_amd._lib import check, lib  # noqa: E402
from deft_amd.utils.workloads import GEOMETRY, WORKLOADS, Workload, algorithmic_bytes, build_forest, build_tree  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak


class Bench:
    def __init__(self, w: Workload, layers: int, device: torch.device, seed: int = 0, extra_steps: int = 64):
        self.w = w
        self.Hq, self.Hkv, self.D, _ = GEOMETRY[w.model]
        self.layers = layers
        self.device = device
        # (--mode node_chunk = DeFT-Node with nodes cut into 128-token entries: BLOCK_CONFIG["MAX_BLOCK_LEN"] is set by the mode's
        #  spelling, examples/run_DeFT_llama_paged.py:145-150, and must be in force BEFORE the metadata is built; every other mode
        #  resets it)
        deft_amd.BLOCK_CONFIG["MAX_BLOCK_LEN"] = -1
        mode = deft_amd.forward_mode_from_cli(w.mode)
        t0 = time.perf_counter()
        if w.trees > 1:  # a batch of independent trees in one pool, one operator call per layer
            self.forest, self.pool = build_forest(w, w.trees, layers, str(device))
        else:
            # (room for the advancing-tree loop: every leaf grows by a token per step)
            tree, self.pool = build_tree(w, layers, str(device), extra_slots=256 + extra_steps * max(w.width, 1))
            self.forest = deft_amd.Forest([tree])
        self.tree_build_s = time.perf_counter() - t0
        builds = []
        for _ in range(4):  # the first build pays lazy initialisation; report the median of the rest
            t0 = time.perf_counter()
            self.md = (self.forest.metadata() if w.trees > 1
                       else deft_amd.TreeMetadata.from_tree_cache(self.forest.trees[0]))
            torch.cuda.synchronize(device)
            builds.append((time.perf_counter() - t0) * 1e3)
        self.metadata_build_ms = sorted(builds[1:])[1]
        self.nq = self.md.query_num
        self.n_kv = self.md.total_kv_len
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        st = self.pool._storage
        for l in range(layers):  # N(0,1) fp16, like the reference's kernel script (test_DeFT_kernel.py:52-54)
            st[l].normal_(generator=g)
        self.q = torch.randn((layers, self.nq, self.Hq * self.D), dtype=torch.float16, device=device, generator=g)
        self.k_new = torch.randn((layers, self.nq, self.Hkv * self.D), dtype=torch.float16, device=device, generator=g)
        self.v_new = torch.randn((layers, self.nq, self.Hkv * self.D), dtype=torch.float16, device=device, generator=g)
        leaves = [lf for t in self.forest.trees for lf in sorted(t.leaves.values(), key=lambda n: n.id)]
        loc = torch.tensor([lf.kv_indices[-1] for lf in leaves], dtype=torch.int32, device=device)
        self.updater = deft_amd.KVCacheUpdater(True, self.pool, loc, None, False)
        if w.mode == "seq":  # sequential comparator: every leaf attends to its own full path through the page table
            tree = self.forest.trees[0]
            lens = [len(tree.leaf_path_slots(lf)) for lf in leaves]
            positions = torch.tensor(lens, dtype=torch.int64, device=device) - 1
            self.meta = deft_amd.InputMetadata.from_tree(tree, tree.req_to_token_pool, self.pool, mode, positions, self.updater)
        else:
            self.meta = deft_amd.InputMetadata(mode, self.updater, self.pool)
        self.attn = [deft_amd.DeFTAttention(self.Hq, self.D, self.D ** -0.5, self.Hkv, l) for l in range(layers)]
        self.out = None
        self.graph = None
        self.launch = "eager"

    def step_eager(self):
        o = None
        for l in range(self.layers):
            o = self.attn[l](self.q[l], self.k_new[l], self.v_new[l], self.meta)
        self.out = o

    def prepare(self, use_graph: bool = True):
        deft_amd.register_tree_metadata(self.md)
        self.step_eager()
        self.step_eager()
        torch.cuda.synchronize(self.device)
        if not use_graph:
            return
        try:
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    self.step_eager()
            torch.cuda.current_stream(self.device).wait_stream(side)
            graph.replay()
            torch.cuda.synchronize(self.device)
            self.graph = graph
            self.launch = "hipgraph"
        except Exception as e:  # keep measuring eagerly, say so in the JSON
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            self.graph = None
            self.launch = "eager"
            torch.cuda.synchronize(self.device)

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.step_eager()

    def release(self):
        """Drop the pools and the per-layer inputs (the scalars and the metadata stay): under a shared GPU -- eight gloo ranks on
        one device -- the headline's 6.7 GB must not sit beside the sharded forest's."""
        self.graph = None
        self.forest = self.pool = self.q = self.k_new = self.v_new = self.updater = self.meta = self.attn = self.out = None
        torch.cuda.empty_cache()

    def algorithmic_bytes_per_layer(self) -> int:
        return algorithmic_bytes(self.n_kv, self.nq, self.Hq, self.Hkv, self.D)

    def time_stage1(self, reps: int):
        """Average duration of the dominant kernel (Flatten stage 1), one HIP event pair per launch,
        on the stream the kernel is launched on (torch's current stream)."""
        if self.w.mode != "flatten":
            return None
        md, pool = self.md, self.pool
        NB, P = md.block_q_cnts.shape[0], md.block_q.shape[0]
        ws_bytes = lib.deft_flatten_workspace_bytes(NB, P, self.nq, self.Hq, self.Hkv, self.D)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device)
        from deft_amd.tree_attention import _flatten_plan

        mdl = [md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv, md.block_lens]
        q0 = self.q[0].view(self.nq, self.Hq, self.D)
        plan = _flatten_plan(mdl, NB, P, self.Hq, self.Hkv, (q0.stride(0), q0.stride(1)), pool.get_key_buffer(0).stride(0),
                             stream.cuda_stream)  # once per step
        def launch_all():
            for l in range(self.layers):
                q = self.q[l].view(self.nq, self.Hq, self.D)
                kb, vb = pool.get_key_buffer(l), pool.get_value_buffer(l)
                rc = lib.deft_flatten_stage1_f16(
                    q.data_ptr(), q.stride(0), q.stride(1), kb.data_ptr(), vb.data_ptr(), kb.stride(0), kb.stride(1),
                    md.block_q.data_ptr(), md.block_q_cnts.data_ptr(), md.block_q_offset.data_ptr(),
                    md.block_bitmasks.data_ptr(), md.block_kv.data_ptr(), md.block_lens.data_ptr(),
                    NB, P, self.nq, self.Hq, self.Hkv, self.D, self.D ** -0.5, plan.data_ptr(), ws.data_ptr(), ws_bytes,
                    torch.cuda.current_stream(self.device).cuda_stream)
                check(rc, "deft_flatten_stage1_f16")

        # one launch per layer pool, back to back on one stream (captured in a hipGraph so that host
        # launch latency does not leak into the device-side interval), one event pair around each sweep
        launch_all()
        torch.cuda.synchronize(self.device)
        graph = None
        try:
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    launch_all()
            torch.cuda.current_stream(self.device).wait_stream(side)
        except Exception:
            graph = None
        def sweep():
            if graph is not None:
                graph.replay()
            else:
                launch_all()

        for _ in range(5):  # warm-up sweeps: clocks and TLBs settle within the first few (first sweeps are ~1 us slower)
            sweep()
        sweeps = []
        for r in range(reps):  # `reps` groups of 4 sweeps, back to back inside one event pair each
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(4):
                sweep()
            e1.record(stream)
            torch.cuda.synchronize(self.device)
            sweeps.append(e0.elapsed_time(e1) * 1e3 / (4 * self.layers))  # us per launch
        sweeps.sort()
        return {"mean_us": sum(sweeps) / len(sweeps), "median_us": sweeps[len(sweeps) // 2],
                "launches": len(sweeps) * 4 * self.layers, "launch": "hipgraph" if graph is not None else "eager"}

    def ceiling_us(self, reps: int = 3):
        """What ONE cold launch that only READS this tree's K/V bytes takes on this GPU (deft_probe_stream_read: coalesced 16-byte
        loads, nothing else), over the same rotating layer pools, captured and timed like `time_stage1`: the hardware ceiling a
        stage-1 launch of this size is judged against.  The tree's tokens sit in the pool's lowest slots
        (prompt contiguous, leaf tokens interleaved step by step), so the byte range is the K/V the kernel reads.  Best of
        256 / 512 / 1024 workgroups."""
        slot_bytes = self.pool.kv_data[0].stride(0) * 2
        nbytes = int(self.n_kv) * slot_bytes
        # (the ceiling reads the pool's first n_kv slots: say so only while that IS where the tree's tokens are -- ADVICE r4)
        bk = self.md.block_kv if getattr(self.md, "block_kv", None) is not None else None
        if bk is not None and bk.numel():
            top = int(bk.max().item())
            assert top < int(self.n_kv) + 64 * max(self.w.width, 1) + 256, (top, self.n_kv)
        stream = torch.cuda.current_stream(self.device)
        best = None
        for wgs in (256, 512, 1024):
            def launch_all():
                for l in range(self.layers):
                    check(lib.deft_probe_stream_read(self.pool.kv_data[l].data_ptr(), nbytes, wgs,
                                                     torch.cuda.current_stream(self.device).cuda_stream), "deft_probe_stream_read")
            launch_all()
            torch.cuda.synchronize(self.device)
            graph = None
            try:
                graph = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(stream)
                with torch.cuda.stream(side):
                    with torch.cuda.graph(graph, stream=side):
                        launch_all()
                stream.wait_stream(side)
            except Exception:
                graph = None
            sweep = graph.replay if graph is not None else launch_all
            for _ in range(3):
                sweep()
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(4):
                    sweep()
                e1.record(stream)
                torch.cuda.synchronize(self.device)
                ts.append(e0.elapsed_time(e1) * 1e3 / (4 * self.layers))
            us = sorted(ts)[len(ts) // 2]
            if best is None or us < best[0]:
                best = (us, wgs)
        return {"us": round(best[0], 2), "workgroups": best[1], "kv_bytes": nbytes}

    def time_plan(self, reps: int = 7):
        """Median duration (us) of the per-step plan kernels -- the device-side repack of the metadata every layer of
        a decode step shares -- built once per step by the operators (cached on the metadata tensors, so the hipGraph
        step above does not contain it): HIP events on the launching stream, right behind attention launches."""
        md, pool = self.md, self.pool
        q0 = self.q[0].view(self.nq, self.Hq, self.D)
        kss = pool.get_key_buffer(0).stride(0)
        s = torch.cuda.current_stream(self.device).cuda_stream
        if self.w.mode == "flatten":
            mdl = [md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv, md.block_lens]
            NB, P = md.block_q_cnts.shape[0], md.block_q.shape[0]
            nbytes = lib.deft_flatten_plan_bytes(NB, P, self.Hq, self.Hkv)
            plan = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=self.device)
            build = lambda: check(lib.deft_flatten_build_plan(*[t.data_ptr() for t in mdl], NB, P, self.Hq, self.Hkv, q0.stride(0),
                                                              q0.stride(1), kss, None, 0, 0, plan.data_ptr(), nbytes, s), "plan")
        elif self.w.mode in ("node", "node_chunk"):
            mdl = [md.node_kv, md.node_kv_offset, md.node_kv_len, md.node_q, md.node_q_offset, md.node_q_len]
            NE, P, total = md.node_kv_offset.shape[0], md.node_q.shape[0], md.node_kv.shape[0]
            nbytes = lib.deft_node_plan_bytes(NE, P, total, self.Hq, self.Hkv)
            plan = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=self.device)
            build = lambda: check(lib.deft_node_build_plan(*[t.data_ptr() for t in mdl], NE, P, total, self.Hq, self.Hkv, q0.stride(0),
                                                           q0.stride(1), kss, None, 0, 0, plan.data_ptr(), nbytes, s), "plan")
        else:
            return None
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.step()
            e0.record()
            build()
            e1.record()
            torch.cuda.synchronize(self.device)
            ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts)[len(ts) // 2]

    def cpu_baseline(self, budget_s: float):
        """the PyTorch-CPU sequential attention in fp32 AND in fp16, same tree, same run."""
        from oracle.cpu_baseline import time_cpu_baseline  # the checker/baseline, never the product path

        paths = self.forest.leaf_paths()
        q = self.q[0].view(self.nq, self.Hq, self.D).float().cpu()
        kv = self.pool.kv_data[0].float().cpu()
        f32 = time_cpu_baseline(q, kv, paths, self.layers, budget_s=budget_s * 0.6)
        f16 = time_cpu_baseline(q, kv, paths, self.layers, budget_s=budget_s * 0.4, dtype=torch.float16)
        return f32, f16

    def step_percentiles(self, steps: int):
        """Per-step GPU time, one HIP event pair per step on the launching stream: p10 / p50 / p90 (us)."""
        stream = torch.cuda.current_stream(self.device)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        for _ in range(5):
            self.step()
        ev[0].record(stream)
        for i in range(steps):
            self.step()
            ev[i + 1].record(stream)
        torch.cuda.synchronize(self.device)
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(steps))
        pick = lambda f: round(ts[min(len(ts) - 1, int(f * len(ts)))], 1)
        return {"p10_us": pick(0.10), "p50_us": pick(0.50), "p90_us": pick(0.90), "steps": steps}

    def end_to_end(self, steps: int, graphed: bool, incremental: bool = True, win_tiles=None):
        """The decode loop a runner really executes, with the tree ADVANCING: per step every leaf takes a token, the host
        allocator hands out nq slots (the only thing that crosses PCIe), the device copy of the tree is advanced by a
        kernel, TreeMetadata and the per-step plan are built on the GPU, then the 32 attention layers (fused append +
        stage 1 + merge).  `graphed`: deft_amd.FlattenDecodeSession -- that whole device-side sequence captured ONCE per
        structural epoch of the tree and replayed per step; otherwise the reference-shaped calls (tree.alloc(),
        TreeMetadata.from_tree_cache, DeFTAttention.forward per layer), eager launches.  No host sync inside the loop."""
        if self.w.trees > 1 or self.w.mode != "flatten":
            return None
        tree = self.forest.trees[0]
        mode = deft_amd.forward_mode_from_cli(self.w.mode)
        if graphed:
            sess = deft_amd.FlattenDecodeSession(tree, self.Hq, self.Hkv, self.D, self.layers,
                                                 lambda l: (self.q[l], self.k_new[l], self.v_new[l]), incremental=incremental,
                                                 win_tiles=win_tiles)

            def one():
                for leaf in tree.leaves.values():
                    leaf.append_token(7)
                sess.step()
        else:
            sess = None

            def one():
                for leaf in tree.leaves.values():
                    leaf.append_token(7)
                upd = tree.alloc()
                md = deft_amd.TreeMetadata.from_tree_cache(tree)
                deft_amd.register_tree_metadata(md)
                meta = deft_amd.InputMetadata(mode, upd, self.pool)
                for l in range(self.layers):
                    self.attn[l](self.q[l], self.k_new[l], self.v_new[l], meta)

        for _ in range(4):
            one()
        torch.cuda.synchronize(self.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            one()
        e1.record()
        host_s = time.perf_counter() - t0
        torch.cuda.synchronize(self.device)
        wall_s = time.perf_counter() - t0
        lens = [len(lf.kv_indices) for lf in tree.leaves.values()]
        end_len = sum(lens) / max(len(lens), 1)
        return {"steps": steps, "ms_per_step": round(wall_s / steps * 1e3, 4), "tokens_per_s": round(self.nq / (wall_s / steps), 1),
                # (the tree GROWS in this loop: its leaves averaged this many tokens over the timed steps -- compare with a frozen
                #  step of THAT tree, `frozen_step_at_mean_len` below, not with the headline's shorter one)
                "mean_branch_len": round(end_len - (steps - 1) / 2.0, 1),
                "gpu_ms_per_step": round(e0.elapsed_time(e1) / steps, 4), "host_ms_per_step": round(host_s / steps * 1e3, 4),
                **({"step_kinds": dict(sess.step_kinds), "graph_captures": sess.captures} if sess is not None else {}),
                "launch": ("hipGraphs per structural epoch of the tree (deft_amd.FlattenDecodeSession): window plans -- one patch kernel "
                           "in front of most steps' layers, metadata + plan rebuilt once per window" if incremental else
                           "one hipGraph per structural epoch of the tree (deft_amd.FlattenDecodeSession(incremental=False): metadata + "
                           "plan rebuilt on every step, the round-5 loop)") if graphed
                          else "eager (tree.alloc, TreeMetadata.from_tree_cache, DeFTAttention.forward per layer)"}


def run_level(w: Workload, layers: int, device, gen_len: int, incremental: bool, win_tiles=None):
    """The reference's few-shot experiment AS A RUN (README.md:214-219: a 4000-token prompt, 400 generated tokens per branch; logs in
    DeFT/experiments/few_shot_prompting/few_shot.ipynb): the north-star tree growing from 1 to `gen_len` tokens per branch through
    deft_amd.FlattenDecodeSession, no host sync inside -- every step of it, the epochs' first (eager) steps and graph captures
    included.  `run_hbm_frac` = the algorithmic bytes of all its attention launches / the GPU time of the run / 8 TB/s: what the >= 60 %
    target looks like averaged over the branch lengths a generation run actually passes through."""
    b = Bench(Workload(**{**w.__dict__, "branch_len": 1}), layers, device, seed=11, extra_steps=gen_len + 8)
    tree = b.forest.trees[0]
    sess = deft_amd.FlattenDecodeSession(tree, b.Hq, b.Hkv, b.D, layers, lambda l: (b.q[l], b.k_new[l], b.v_new[l]), incremental=incremental,
                                         win_tiles=win_tiles)
    steps = gen_len - 1
    n_kv, total_bytes = b.n_kv, 0
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        for leaf in tree.leaves.values():
            leaf.append_token(7)
        sess.step()
        n_kv += b.nq
        total_bytes += algorithmic_bytes(n_kv, b.nq, b.Hq, b.Hkv, b.D) * layers
    e1.record()
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    gpu_s = e0.elapsed_time(e1) * 1e-3
    out = {"steps": steps, "branch_len": [1, gen_len], "generated_tokens": steps * b.nq, "wall_ms": round(wall * 1e3, 2),
           "gpu_ms": round(gpu_s * 1e3, 2), "ms_per_step": round(wall / steps * 1e3, 4), "tokens_per_s": round(steps * b.nq / wall, 1),
           "algorithmic_TB": round(total_bytes / 1e12, 4), "run_GBps": round(total_bytes / gpu_s / 1e9, 1),
           "run_hbm_frac": round(total_bytes / gpu_s / 1e9 / HBM_PEAK_GBPS, 4), "step_kinds": dict(sess.step_kinds),
           "graph_captures": sess.captures}
    del sess, b
    return out


def in_situ(b: "Bench", steps: int = 60):
    """Attention timed where the reference times it: INSIDE the model's forward (DeFT/deft/layers/attention/deft_attention.py:117-149
    runs inside model_runner.py:410-424), i.e. with the layer's dense kernels -- weight streams that sweep L2 and the Infinity Cache --
    between two attention calls.  Harness only: one hipGraph of `layers` x [qkv projection (F.linear, 4096 -> 12288 for Llama-2-7B) ->
    DeFTAttention.forward on strided views of its output -> o projection -> MLP up -> MLP down] at nq rows, fp16 synthetic weights, and
    the same graph WITHOUT the attention call; their difference per layer is the attention's cost in situ, next to the back-to-back
    number of the headline.  (Kernel by kernel: `rocprofv3 --kernel-trace --stats -- python bench.py --in-situ-only`,
    profiles/r6_in_situ_kernel_stats.txt.)"""
    import torch.nn.functional as F

    dev, L, nq = b.device, b.layers, b.nq
    H, Hkv_d, inter = b.Hq * b.D, b.Hkv * b.D, 11008 if b.Hq == b.Hkv else 14336
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    mk = lambda o, i: torch.randn((L, o, i), dtype=torch.float16, device=dev, generator=g) * (i ** -0.5)  # noqa: E731
    Wqkv, Wo, Wup, Wdn = mk(H + 2 * Hkv_d, H), mk(H, H), mk(inter, H), mk(H, inter)
    x = torch.randn((nq, H), dtype=torch.float16, device=dev, generator=g)
    sink = torch.zeros((nq, H), dtype=torch.float16, device=dev)

    def forward(with_attention: bool):
        for l in range(L):
            qkv = F.linear(x, Wqkv[l])
            q, k, v = qkv.split([H, Hkv_d, Hkv_d], dim=-1)
            o = b.attn[l](q, k, v, b.meta) if with_attention else q
            y = F.linear(o.reshape(nq, H), Wo[l])
            z = F.linear(F.linear(y, Wup[l]), Wdn[l])
        sink.copy_(z)

    def capture(with_attention: bool):
        forward(with_attention)
        torch.cuda.synchronize(dev)
        gr = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            with torch.cuda.graph(gr, stream=side):
                forward(with_attention)
        torch.cuda.current_stream(dev).wait_stream(side)
        return gr

    def timed(gr):
        for _ in range(5):
            gr.replay()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            gr.replay()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps

    g_with, g_without = capture(True), capture(False)
    reps = [(timed(g_with), timed(g_without)) for _ in range(3)]
    t_with, t_without = min(r[0] for r in reps), min(r[1] for r in reps)
    weights_mb = (Wqkv[0].numel() + Wo[0].numel() + Wup[0].numel() + Wdn[0].numel()) * 2 / 1e6
    return {"what": "attention inside a model-shaped step: per layer qkv projection -> DeFTAttention.forward -> o projection -> MLP up -> MLP "
                    "down (F.linear at nq rows, synthetic fp16 weights), one hipGraph; attention = that graph minus the same graph without "
                    "the attention call",
            "layers": L, "dense_weights_MB_per_layer": round(weights_mb, 1), "steps": steps,
            "ms_per_step_with_attention": round(t_with * 1e3, 4), "ms_per_step_dense_only": round(t_without * 1e3, 4),
            "attention_us_per_layer_in_situ": round((t_with - t_without) * 1e6 / L, 2)}


def _ctl_device(b: "Bench"):
    """Where the control-plane tensors of the bracket live: the GPU under RCCL, host memory under gloo."""
    import torch.distributed as dist

    return b.device if dist.get_backend() == "nccl" else torch.device("cpu")


def run_timed(b: Bench, steps: int, warmup: int, dist_on: bool):
    import torch.distributed as dist

    if dist_on:
        # rehearse the bracket once, untimed: the first barrier / all-reduce of a process group sets up RCCL's connections
        # and loads its kernels lazily (milliseconds -- a visible share of a 20-step timed region)
        from deft_amd.utils.sharding import max_over_ranks

        dist.barrier()
        torch.cuda.synchronize(b.device)
        max_over_ranks(0.0, _ctl_device(b))
        dist.barrier()
    for _ in range(warmup):
        b.step()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(b.device)
    t0 = time.perf_counter()
    for _ in range(steps):
        b.step()
    torch.cuda.synchronize(b.device)
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        from deft_amd.utils.sharding import max_over_ranks

        dt = max_over_ranks(dt, _ctl_device(b))  # control plane only; the data path has no collective
    return dt


def measure_traffic(args, kernel: str):
    """HBM bytes per launch of the dominant kernel, measured NOW: this same command re-run as a child under
    `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (own pass, nothing else enabled), read as the guide prescribes
    (/opt/skills/guides/MI355X_MICROARCH.md, "HBM": FETCH_SIZE counts 64 B per 128 B request for 16-byte-per-lane
    streams on gfx950 -> x2).  None when rocprofv3 is not there."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None
    out = tempfile.mkdtemp(prefix="deft_pmc_", dir="/tmp")
    cmd = [rp, "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable,
           os.path.abspath(__file__), "--workload", args.workload, "--steps", "3", "--warmup", "1", "--child"]
    if args.branch_len is not None:
        cmd += ["--branch-len", str(args.branch_len)]
    if args.layers:
        cmd += ["--layers", str(args.layers)]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=False)
        vals = []
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    if kernel in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE":
                        vals.append(float(row["Counter_Value"]))
        if not vals:
            return None
        return {"bytes": int(sum(vals) / len(vals) * 1024 * 2.0), "launches": len(vals)}
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def gpu_state():
    """Clock / power mode of the GPU the numbers were taken on (state it)."""
    import shutil
    import subprocess

    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        out = subprocess.run([smi, "--showperflevel", "--showpower", "--showclocks", "--json"], capture_output=True, text=True,
                             timeout=20).stdout
        d = json.loads(out)
        c = d.get("card0", next(iter(d.values())))
        keep = {k: v for k, v in c.items() if any(t in k.lower() for t in ("performance level", "sclk", "mclk", "power"))}
        return keep or None
    except Exception:
        return None


def cfg5_line(device, layers_model: str, n_gpus: int, rank: int, dist_on: bool, steps: int, warmup: int, use_graph: bool):
    """BASELINE configs[4]: a batch of 8 N independent 8k-prefix trees (Llama-3-8B, 8 branches x 64 tokens) sharded
    over the N GPUs -- `shard_trees`, 8 trees per GPU at any N (weak scaling), every rank decodes its share as ONE
    Forest call per layer, no data-path collective (RCCL carries the timing barrier and MAX only)."""
    from deft_amd.utils.sharding import cfg5_shard

    w = WORKLOADS["forest_8kx8"]
    mine = cfg5_shard(n_gpus, rank, trees_per_gpu=w.trees)
    wv = Workload(**{**w.__dict__, "trees": len(mine)})
    b = Bench(wv, GEOMETRY[w.model][3], device, seed=100 + rank)
    b.prepare(use_graph=use_graph)
    dt = run_timed(b, steps, warmup, dist_on)
    s1 = b.time_stage1(reps=2)
    algo = b.algorithmic_bytes_per_layer()
    e2e = None
    try:  # the advancing loop of the same share, the batch as ONE tree object (TreeCache.init_forest) through the session
        e2e = forest_end_to_end(b, wv, device, min(steps, 40), dist_on)
    except Exception as e:
        e2e = {"error": f"{type(e).__name__}: {e}"}
        if dist_on:
            raise  # (a rank that skips the loop's barriers would hang the others)
    shares = [mine]
    if dist_on:
        import torch.distributed as dist

        shares = [None] * n_gpus
        dist.all_gather_object(shares, mine)  # (control plane, after the timed regions: what every rank decoded)
    return {"workload": "BASELINE configs[4]: %d independent trees (8192-token prefix x 8 branches x 64 tokens, Llama-3-8B "
                        "DeFT-Flatten) over %d GPU(s), %d per GPU" % (n_gpus * w.trees, n_gpus, len(mine)),
            "end_to_end": e2e,
            "trees_this_rank": mine, "trees_by_rank": shares, "queries_per_gpu": b.nq, "kv_tokens_per_gpu": b.n_kv,
            "tokens_per_s": round(n_gpus * b.nq / (dt / steps), 1), "ms_per_step": round(dt / steps * 1e3, 4),
            "us_per_layer": round(dt / steps * 1e6 / b.layers, 2), "steps": steps,
            "stage1_us": round(s1["mean_us"], 2) if s1 else None,
            "stage1_hbm_frac": round(algo / (s1["mean_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if s1 else None,
            "collectives_in_data_path": 0, "rccl": "control plane only (barrier + MAX of the step time)",
            "dist_backend": (__import__("torch").distributed.get_backend() if dist_on else None)}


def forest_end_to_end(b: "Bench", w: Workload, device, steps: int, dist_on: bool = False):
    """cfg5's decode LOOP on one GPU: the rank's trees as one tree object (a root without tokens), every leaf growing a token per
    step, deft_amd.FlattenDecodeSession (tree advance, TreeMetadata and plan on the GPU, 32 layers, one hipGraph), wall clock
    without host syncs; next to it the host time of `Forest.metadata()` for the same batch -- what the per-step metadata of
    separate trees costs when it is built on the host."""
    from deft_amd.utils.workloads import build_forest_tree

    t0 = time.perf_counter()
    for _ in range(3):
        b.forest.metadata()
    torch.cuda.synchronize(device)
    host_md_ms = (time.perf_counter() - t0) / 3 * 1e3
    layers = b.layers
    q, k_new, v_new = b.q, b.k_new, b.v_new
    del b.graph
    b.graph = None
    tree, pool = build_forest_tree(w, w.trees, layers, str(device), extra_slots=256 + 64 * w.trees * w.width)
    g = torch.Generator(device=device)
    g.manual_seed(5)
    for l in range(layers):
        pool._storage[l].normal_(generator=g)
    sess = deft_amd.FlattenDecodeSession(tree, b.Hq, b.Hkv, b.D, layers, lambda l: (q[l], k_new[l], v_new[l]))

    def one():
        for leaf in tree.leaves.values():
            leaf.append_token(7)
        sess.step()

    for _ in range(4):
        one()
    if dist_on:
        import torch.distributed as dist

        dist.barrier()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        one()
    e1.record()
    torch.cuda.synchronize(device)
    if dist_on:
        dist.barrier()
    wall = time.perf_counter() - t0
    if dist_on:
        from deft_amd.utils.sharding import max_over_ranks

        wall = max_over_ranks(wall, _ctl_device(b))
    return {"what": "the rank's trees as ONE tree object (TreeCache.init_forest) through deft_amd.FlattenDecodeSession, every leaf "
                    "growing a token per step; no host sync inside the loop",
            "steps": steps, "ms_per_step": round(wall / steps * 1e3, 4), "gpu_ms_per_step": round(e0.elapsed_time(e1) / steps, 4),
            "tokens_per_s": round(len(tree.leaves) / (wall / steps), 1), "graph_captures": sess.captures,
            "host_metadata_ms_of_separate_trees": round(host_md_ms, 3)}


def prefill_lines(device, model: str):
    """Causal prefill attention of one prompt (deft_prefill_f16 behind context_attention_fwd), q / k / v = strided views of
    a fused qkv projection; TFLOP/s counts the causal half against the dense fp16 MFMA peak (2.5 PFLOP/s)."""
    Hq, Hkv, D, _ = GEOMETRY[model]
    out = {"kernel": "deft::prefill_kernel<128>", "bound": "mfma", "peak_TFLOPs": 2500.0, "prompts": {}}
    # (the same prompt lengths with twice the heads of half the width -- head_dim 64, prefill_kernel<64> -- as "<S>_head_dim_64")
    for S, (Hq, Hkv, D) in [(4096, (Hq, Hkv, D)), (16384, (Hq, Hkv, D)), (4096, (2 * Hq, 2 * Hkv, D // 2)), (16384, (2 * Hq, 2 * Hkv, D // 2))]:
        qkv = torch.randn((S, (Hq + 2 * Hkv) * D), dtype=torch.float16, device=device)
        q, k, v = (t.view(S, -1, D) for t in qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1))
        o = torch.empty((S, Hq, D), dtype=torch.float16, device=device)
        start = torch.zeros(1, dtype=torch.int32, device=device)
        lens = torch.tensor([S], dtype=torch.int32, device=device)
        for _ in range(3):
            deft_amd.context_attention_fwd(q, k, v, o, start, lens, S)
        n = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            deft_amd.context_attention_fwd(q, k, v, o, start, lens, S)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        flops = 2.0 * 2.0 * (S * (S + 1) / 2) * D * Hq
        out["prompts"][str(S) if D == GEOMETRY[model][2] else f"{S}_head_dim_{D}"] = {
            "us_per_layer": round(us, 1), "TFLOPs": round(flops / us / 1e6, 1), "frac": round(flops / us / 1e6 / 2500.0, 4)}
        del qkv, o
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="northstar_4kx32", choices=sorted(WORKLOADS))
    ap.add_argument("--branch-len", type=int, default=None, help="override tokens per branch (few_shot trees)")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 FETCH_SIZE child pass")
    ap.add_argument("--no-e2e", action="store_true", help="skip the advancing-tree end-to-end loop")
    ap.add_argument("--win-tiles", type=int, default=None, help="end-to-end loop: DecodeSession(win_tiles=), overflow tiles per region of a window plan")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the sharded-forest (BASELINE configs[4]) measurement")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--in-situ-only", action="store_true",
                    help="run the model-shaped step (dense layer kernels between the attention calls, bench.in_situ) and nothing else: for "
                         "`rocprofv3 --kernel-trace --stats`, so that the stage-1 / merge averages are those of launches with neighbours")
    ap.add_argument("--step-only", action="store_true",
                    help="run the timed step graph and nothing else (no stage-1-only sweeps, percentiles, plan timing): for "
                         "`rocprofv3 --kernel-trace --stats`, so that the per-kernel averages are those of the STEP's launches")
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="process-group backend of the N > 1 bracket (barrier + MAX of the step time; the data path has no "
                         "collective).  nccl = RCCL, one rank per GPU.  gloo: control plane over TCP, and ranks beyond the visible "
                         "GPUs share them round-robin -- rehearses every line of the N > 1 path on a ONE-GPU box")
    ap.add_argument("--force-dist", action="store_true",
                    help="set the process group up even at world size 1 (under a launcher: torch.distributed.run --nproc-per-node 1): "
                         "the N > 1 bracket -- init_process_group, barrier, MAX all-reduce on the control device -- on ONE GPU, which "
                         "is how RCCL itself gets exercised where no multi-GPU box is at hand")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)  # the PMC child: timed steps only
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        # `python bench.py --gpus N` outside a launcher: become the launcher (one rank per GPU, RCCL rendezvous on 127.0.0.1)
        import socket
        import subprocess

        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    dist_on = world > 1 or (args.force_dist and "RANK" in os.environ)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: deft_amd has no CPU path")
    shared_gpus = args.dist_backend == "gloo" and world > torch.cuda.device_count()
    device = torch.device("cuda", local_rank % torch.cuda.device_count() if shared_gpus else local_rank)
    torch.cuda.set_device(device)
    if dist_on:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    n_gpus = world if dist_on else 1

    w = WORKLOADS[args.workload]
    if args.branch_len is not None:
        w = Workload(**{**w.__dict__, "branch_len": args.branch_len})
    layers = args.layers or GEOMETRY[w.model][3]

    b = Bench(w, layers, device, seed=rank)
    b.prepare(use_graph=not args.no_graph)
    dt = run_timed(b, args.steps, args.warmup, dist_on)
    ms_per_step = dt / args.steps * 1e3
    tokens_per_s = n_gpus * b.nq / (dt / args.steps)
    if args.child:  # PMC child: the parent only wants the kernels to have run
        if w.mode == "flatten":
            b.time_stage1(reps=1)
        return
    if args.in_situ_only:
        if rank == 0:
            print(json.dumps({"in_situ": in_situ(b), "attention_us_per_layer_back_to_back": round(ms_per_step * 1e3 / layers, 2)}), flush=True)
        return
    if args.step_only:
        if rank == 0:
            print(json.dumps({"metric": "tree_tokens_per_s", "value": round(tokens_per_s, 1), "unit": "tokens/s", "n_gpus": n_gpus,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
                              "attention_latency_us_per_layer": round(ms_per_step * 1e3 / layers, 2),
                              "note": "--step-only: the step graph alone (for kernel-trace statistics)"}), flush=True)
        return

    s1 = b.time_stage1(reps=3)
    pct = b.step_percentiles(min(args.steps, 200))
    plan_us = b.time_plan()
    algo = b.algorithmic_bytes_per_layer()
    roofline = None
    if s1 is not None:
        achieved = algo / (s1["mean_us"] * 1e-6) / 1e9
        kind = "stage1_np_kernel"
        traffic, traffic_source = None, None
        if rank == 0 and not dist_on and not args.no_traffic:
            t = measure_traffic(args, kind)
            if t:
                traffic = t["bytes"]
                traffic_source = (f"measured in this run: rocprofv3 --pmc FETCH_SIZE --kernel-trace on a child of this command "
                                  f"({t['launches']} launches), x2 gfx950 correction (MI355X_MICROARCH.md, HBM)")
        if traffic is None:
            try:  # fall back to the committed pass of the same command, and say so
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r2a_pmc_fetch_size_northstar_4kx32.json")))
                if w.name == "northstar_4kx32" and w.branch_len == 200:
                    k = [v for n, v in pmc["kernels"].items() if kind in n]
                    traffic = k[0]["hbm_read_bytes_per_launch"] if k else None
                    traffic_source = "profiles/r2a_pmc_fetch_size_northstar_4kx32.json (committed pass, NOT this run)" if traffic else None
            except Exception:
                traffic = None
        ceil = b.ceiling_us()
        roofline = {"bound": "hbm", "kernel": f"deft::{kind}<128> (Flatten stage 1)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": algo, "avg_launch_us": round(s1["mean_us"], 2),
                    "median_launch_us": round(s1["median_us"], 2), "launches_timed": s1["launches"],
                    "timing": ("HIP events around a hipGraph of one launch per layer pool" if s1.get("launch") == "hipgraph"
                               else "HIP events around eager launches") +
                              "; the launches timed are deft_flatten_stage1_f16 = " +
                              ("stage1_np_kernel<128, false, NT> (MHA: the plain instantiation)" if b.Hq == b.Hkv else
                               "stage1_np_kernel<128, false, true, false, false, DYN = true> (GQA: per-chunk cache policy, mirrored item "
                               "order; a GQA launch beyond 16 x CUs items with at most 160 virtual rows runs the plain one)") +
                              " WITHOUT the fused append -- the step's own launches (deft_flatten_decode_append_f16) are the same "
                              "template instantiation with n_new = nq new rows copied into the pool by its first workgroups",
                    # the hardware ceiling of a launch of this size: a bare read of the same K/V bytes (deft_probe_stream_read)
                    "ceiling_us": ceil["us"], "ceiling_workgroups": ceil["workgroups"],
                    "launch_over_ceiling": round(s1["mean_us"] / ceil["us"], 3)}
    step_achieved = algo * layers / (dt / args.steps) / 1e9
    e2e = None
    if rank == 0 and not dist_on and not args.no_e2e and w.trees == 1 and w.mode == "flatten":
        e2e = {"what": "the decode loop with the tree ADVANCING (one token per leaf and step, from the benchmarked tree on): "
                       "host slot allocation + nq slot numbers over PCIe + device tree advance + TreeMetadata and plan built on the "
                       "GPU + 32 x (fused append, stage 1, merge); wall clock over the loop, no host sync inside"}
        for key, graphed, inc in (("graphed", True, True), ("graphed_rebuild_every_step", True, False), ("eager", False, False)):
            try:  # each on a fresh tree (the loop grows it)
                torch.cuda.empty_cache()
                be = Bench(w, layers, device, seed=7)
                be.prepare(use_graph=False)
                e2e[key] = be.end_to_end(min(50, max(10, args.steps // 4)), graphed, incremental=inc, win_tiles=args.win_tiles)
                del be
            except Exception as e:
                e2e[key] = {"error": f"{type(e).__name__}: {e}"}
        try:  # the attention-only replay of a FROZEN step of the tree the loop averaged over: what the loop's steps cost without
            # the per-step work around the attention launches (slot upload, tree advance, TreeMetadata, plan)
            mean_len = e2e.get("graphed", {}).get("mean_branch_len")
            if mean_len and w.kind == "few_shot":
                torch.cuda.empty_cache()
                bf = Bench(Workload(**{**w.__dict__, "branch_len": int(round(mean_len))}), layers, device, seed=7)
                bf.prepare(use_graph=not args.no_graph)
                nf = min(50, max(10, args.steps // 4))
                dtf = run_timed(bf, nf, 5, False)
                fz = dtf / nf * 1e3
                e2e["frozen_step_at_mean_len"] = {"branch_len": int(round(mean_len)), "ms_per_step": round(fz, 4), "steps": nf}
                for key in ("graphed", "graphed_rebuild_every_step", "eager"):
                    if isinstance(e2e.get(key), dict) and "ms_per_step" in e2e[key]:
                        e2e[key]["over_frozen_step_at_mean_len"] = round(e2e[key]["ms_per_step"] / fz, 4)
                del bf
                # A frozen step's time need not be linear in the branch length (the Flatten split packs the branches' tokens into 128-slot
                # blocks across leaf boundaries, and the leaf blocks' groups leave a remainder: 240 -> 254 tokens per branch was +11 % for
                # +4 % of bytes before round 6's last plan rule, profiles/r6_union_len_sweep.txt), so the frozen step AT the mean length can
                # understate what the loop's steps cost frozen: the second reference is the mean of frozen steps at five lengths across
                # the loop's range.
                n_loop = e2e["graphed"]["steps"]
                first = mean_len - (n_loop - 1) / 2.0
                lens = sorted({int(round(first + f * (n_loop - 1))) for f in (0.0, 0.25, 0.5, 0.75, 1.0)})
                per_len = {}
                for bl in lens:
                    torch.cuda.empty_cache()
                    bf = Bench(Workload(**{**w.__dict__, "branch_len": bl}), layers, device, seed=7)
                    bf.prepare(use_graph=not args.no_graph)
                    per_len[str(bl)] = round(run_timed(bf, 30, 5, False) / 30 * 1e3, 4)
                    del bf
                fzm = sum(per_len.values()) / len(per_len)
                e2e["frozen_steps_across_the_loop"] = {
                    "ms_per_step_by_branch_len": per_len, "mean_ms_per_step": round(fzm, 4),
                    "why": "a frozen step need not be linear in the branch length (the Flatten split's leaf blocks are grouped in threes or "
                           "fours and the remainder is a work item of its own: before round 6's last plan rule 240 -> 254 tokens per branch was "
                           "+11 % for +4 % of bytes, profiles/r6_union_len_sweep.txt), so the one frozen step at the loop's MEAN length can be "
                           "cheaper than the loop's steps are frozen; `over_frozen_steps_across_the_loop` is the loop against the mean of these "
                           "five, `over_frozen_step_at_mean_len` (kept for continuity with rounds 4-5) against the one"}
                for key in ("graphed", "graphed_rebuild_every_step", "eager"):
                    if isinstance(e2e.get(key), dict) and "ms_per_step" in e2e[key]:
                        e2e[key]["over_frozen_steps_across_the_loop"] = round(e2e[key]["ms_per_step"] / fzm, 4)
        except Exception as e:
            e2e["frozen_step_at_mean_len"] = {"error": f"{type(e).__name__}: {e}"}

    situ = None
    if rank == 0 and not dist_on and not args.no_extras:
        try:  # attention timed where the reference times it -- between the dense kernels of a layer
            torch.cuda.empty_cache()
            situ = in_situ(b)
            situ["attention_us_per_layer_back_to_back"] = round(ms_per_step * 1e3 / layers, 2)
            situ["in_situ_over_back_to_back"] = round(situ["attention_us_per_layer_in_situ"] / max(ms_per_step * 1e3 / layers, 1e-9), 3)
        except Exception as e:
            situ = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()

    few_shot_run = None
    if rank == 0 and not dist_on and not args.no_e2e and not args.no_extras and w.kind == "few_shot" and w.trees == 1 and w.mode == "flatten":
        # the run-level fraction of the reference's few-shot workload, and where along the branch length it is lost
        few_shot_run = {"what": "the tree of the headline growing from 1 to 400 tokens per branch (the reference's few-shot experiment, "
                                "README.md:214-219) through deft_amd.FlattenDecodeSession: algorithmic bytes of every attention launch "
                                "of the run / GPU time of the run / 8 TB/s; `by_branch_len`: a frozen step at that length"}
        del b.graph
        b.graph = None
        for key, inc in (("window_plans", True), ("rebuild_every_step", False)):
            try:
                torch.cuda.empty_cache()
                few_shot_run[key] = run_level(w, layers, device, 400, inc)
            except Exception as e:
                few_shot_run[key] = {"error": f"{type(e).__name__}: {e}"}
        sweep = {}
        for L in (1, 25, 50, 100, 150, 200, 300, 400):
            try:
                torch.cuda.empty_cache()
                bl = Bench(Workload(**{**w.__dict__, "branch_len": L}), layers, device, seed=3)
                bl.prepare(use_graph=not args.no_graph)
                dtl = run_timed(bl, 100, 10, False)
                al = bl.algorithmic_bytes_per_layer()
                s1l = bl.time_stage1(reps=1)
                sweep[str(L)] = {"us_per_layer": round(dtl / 100 * 1e6 / layers, 2),
                                 "step_hbm_frac": round(al * layers / (dtl / 100) / 1e9 / HBM_PEAK_GBPS, 4),
                                 "stage1_hbm_frac": round(al / (s1l["mean_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if s1l else None}
                del bl
            except Exception as e:
                sweep[str(L)] = {"error": f"{type(e).__name__}: {e}"}
        few_shot_run["by_branch_len"] = sweep

    extras = {}
    if not args.no_extras and rank == 0 and not dist_on:
        variants = []
        if w.kind == "few_shot" and args.branch_len is None:
            variants += [(f"{w.name}_len1", Workload(**{**w.__dict__, "branch_len": 1})),
                         (f"{w.name}_len400", Workload(**{**w.__dict__, "branch_len": 400}))]
        for name in ("fewshot_1kx32", "medusa64_node", "medusa64_tree_node", "medusa64_tree_flatten", "tot50_4k", "gqa_4kx32",
                     "forest_8kx8_single", "northstar_4kx32_node", "northstar_4kx32_node_chunk", "northstar_4kx32_seq",
                     "fewshot_1kx32_seq", "northstar_4kx32_d64"):
            if name != w.name:
                variants.append((name, WORKLOADS[name]))
        del b.graph
        b.graph = None
        for name, wv in variants:
            try:
                torch.cuda.empty_cache()
                bv = Bench(wv, GEOMETRY[wv.model][3], device, seed=1)
                bv.prepare(use_graph=not args.no_graph)
                n = max(100, args.steps // 2)  # >= 100 timed steps
                dtv = run_timed(bv, n, max(5, args.warmup // 2), False)
                s1v = bv.time_stage1(reps=1)
                cv = bv.ceiling_us(reps=1)
                av = bv.algorithmic_bytes_per_layer()
                extras[name] = {
                    "model": wv.model, "mode": wv.mode, "trees": wv.trees, "nq": bv.nq, "kv_tokens": bv.n_kv, "steps": n,
                    "us_per_step": round(dtv / n * 1e6, 1), "us_per_layer": round(dtv / n * 1e6 / bv.layers, 2),
                    "tokens_per_s": round(bv.nq / (dtv / n), 1),
                    "step_GBps": round(av * bv.layers / (dtv / n) / 1e9, 1),
                    "step_hbm_frac": round(av * bv.layers / (dtv / n) / 1e9 / HBM_PEAK_GBPS, 4),
                    "stage1_us": round(s1v["mean_us"], 2) if s1v else None,
                    "stage1_hbm_frac": round(av / (s1v["mean_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if s1v else None,
                    "algorithmic_MB": round(av / 1e6, 2), "ceiling_us": cv["us"], "ceiling_workgroups": cv["workgroups"],
                    "metadata_build_ms": round(bv.metadata_build_ms, 3), "launch": bv.launch,
                    "plan_build_us_per_step": round(pv, 1) if (pv := bv.time_plan(5)) is not None else None,
                }
                del bv
            except Exception as e:  # an extra must never take the headline down
                extras[name] = {"error": f"{type(e).__name__}: {e}"}

    prefill = None
    if not args.no_extras and rank == 0 and not dist_on:
        # the causal prefill in front of the decode path (TTFT of the prompt), same geometry; MFMA-bound
        try:
            torch.cuda.empty_cache()
            prefill = prefill_lines(device, w.model)
        except Exception as e:
            prefill = {"error": f"{type(e).__name__}: {e}"}

    cfg5 = None
    if not args.no_cfg5:
        try:
            del b.graph
            b.graph = None
            if dist_on:
                b.release()  # (nothing below reads the headline's pools when ranks > 1: no CPU baseline, no extras)
            torch.cuda.empty_cache()
            cfg5 = cfg5_line(device, w.model, n_gpus, rank, dist_on, max(50, args.steps // 4), max(5, args.warmup // 4),
                             not args.no_graph)
        except Exception as e:
            cfg5 = {"error": f"{type(e).__name__}: {e}"}
            if dist_on:
                raise

    cpu = None
    if rank == 0 and not dist_on and not args.no_cpu_baseline:
        c, c16 = b.cpu_baseline(args.cpu_budget_s)
        cpu = {"value": round(c["tokens_per_s"], 4), "unit": "tokens/s", "cores": c["cores"], "kind": "port",
               "sample": f"1 of {layers} layer-steps of the same tree ({b.nq} leaves, {b.n_kv} unique KV tokens), "
                         f"PyTorch SDPA {c['dtype']} per leaf incl. page-table gather, {c['cores']} threads (best of 8/16/32/all), "
                         f"{c['warmups']} warm-ups, best of {c['reps']} timed repetitions (BASELINE.md section 3), "
                         f"x{layers} layers extrapolated",
               "ms_per_layer_step": round(c["seconds_per_layer_step"] * 1e3, 2), "host_cores": c["host_cores"],
               "fp16": {"value": round(c16["tokens_per_s"], 4), "ms_per_layer_step": round(c16["seconds_per_layer_step"] * 1e3, 2),
                        "cores": c16["cores"], "warmups": c16["warmups"], "reps": c16["reps"]}}

    if rank == 0:
        Hq, Hkv, D, _ = GEOMETRY[w.model]
        line = {
            "metric": "tree_tokens_per_s", "value": round(tokens_per_s, 1), "unit": "tokens/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{w.model} DeFT-{w.mode}, {w.kind} tree: {w.prefix}-token shared prefix x {w.width} "
                                   f"branches x {w.branch_len} tokens, paged KV, {layers} layers "
                                   f"(Hq={Hq}, Hkv={Hkv}, D={D}); {w.trees} independent tree(s) per GPU",
                       "name": w.name, "queries": b.nq, "unique_kv_tokens": b.n_kv, "layers": layers,
                       "blocks": int(b.md.block_q_cnts.shape[0]), "partial_rows": int(b.md.block_q.shape[0]),
                       "launch": b.launch,
                       "timed_region": "attention-only replay of ONE frozen decode step (hipGraph of 32 x (fused append, stage 1, "
                                       "merge)); metadata / plan of that step are built before it; `end_to_end` below is the "
                                       "advancing-tree loop with them inside"},
            "value_definition": "queries per step x n_gpus / (wall clock of the K timed steps / K): the steps are replayed back to "
                                "back from one hipGraph between two device synchronisations (the driver's contract); "
                                "`step_time_percentiles` brackets EVERY step with its own HIP event pair, which adds ~10 us of "
                                "event work per step -- the wall-clock mean is the headline, the percentiles show the spread",
            "attention_latency_us_per_step": round(ms_per_step * 1e3, 1),
            "attention_latency_us_per_layer": round(ms_per_step * 1e3 / layers, 2),
            "step_time_percentiles": pct,
            "step_algorithmic_GBps": round(step_achieved, 1),
            "step_hbm_frac": round(step_achieved / HBM_PEAK_GBPS, 4),
            "metadata_build_ms": round(b.metadata_build_ms, 3),
            "plan_build_us_per_step": round(plan_us, 1) if plan_us is not None else None,
            "end_to_end": e2e, "few_shot_run": few_shot_run, "in_situ": situ, "gpu_state": gpu_state(),
            "dist": ({"backend": args.dist_backend, "world_size": world, "ranks_share_gpus": bool(shared_gpus),
                      "visible_gpus": torch.cuda.device_count()} if dist_on else None),
            "roofline": roofline, "cpu_baseline": cpu, "cfg5_sharded_forest": cfg5, "prefill": prefill,
            "other_workloads": extras,
        }
        print(json.dumps(line), flush=True)
    if dist_on:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
