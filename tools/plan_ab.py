#!/usr/bin/env python3
"""Plan build time and an output checksum per workload, for same-box A/B of two builds of the library:

    for lib in deft_amd/lib/libdeft_amd_prev.so deft_amd/lib/libdeft_amd.so; do DEFT_AMD_LIB=$(realpath $lib) python tools/plan_ab.py; done

A plan change that is meant to be a pure speed-up must leave every checksum as it was (the partial rows, and so the
output bits, are a function of the plan)."""
import hashlib, json, os, sys
import torch
sys.path.insert(0, os.getcwd())
import deft_amd
from deft_amd._lib import lib, check
from deft_amd.tree_attention import _flatten_plan
from deft_amd.utils.workloads import GEOMETRY, WORKLOADS, build_forest, build_tree

dev = torch.device("cuda", 0)
names = sys.argv[1:] or ["northstar_4kx32", "fewshot_1kx32", "tot50_4k", "gqa_4kx32", "forest_8kx8", "medusa64_node"]
for name in names:
    w = WORKLOADS[name.split(":")[0]]
    if name.endswith(":node"):  # a Flatten workload's tree through the Node operator
        from deft_amd.utils.workloads import Workload
        w = Workload(**{**w.__dict__, "mode": "node"})
    Hq, Hkv, D, _ = GEOMETRY[w.model]
    if w.trees > 1:
        forest, pool = build_forest(w, w.trees, 1, str(dev))
    else:
        tree, pool = build_tree(w, 1, str(dev))
        forest = deft_amd.Forest([tree])
    md = forest.metadata() if w.trees > 1 else deft_amd.TreeMetadata.from_tree_cache(forest.trees[0])
    g = torch.Generator(device=dev); g.manual_seed(1)
    pool._storage[0].normal_(generator=g)
    nq = md.query_num
    q = torch.randn((nq, Hq, D), dtype=torch.float16, device=dev, generator=g)
    kb, vb = pool.get_key_buffer(0), pool.get_value_buffer(0)
    o = torch.empty_like(q)
    if w.mode == "node":
        def call(m): deft_amd.tree_attention_fwd(q, kb, vb, o, m.node_kv, m.node_kv_offset, m.node_kv_len, m.node_q, m.node_q_offset, m.node_q_len)
    else:
        def call(m): deft_amd.tree_attention_subtree_fwd(q, kb, vb, o, m.block_len, m.block_q, m.block_q_cnts, m.block_q_offset, m.block_bitmasks, m.block_kv, m.block_lens)
    call(md); call(md); torch.cuda.synchronize()
    t_att = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(md); e1.record(); torch.cuda.synchronize(); t_att.append(e0.elapsed_time(e1) * 1e3)
    t_plan, t_warm = [], []
    if w.mode != "node":
        NB, P = md.block_q_cnts.shape[0], md.block_q.shape[0]
        mdl = [md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv, md.block_lens]
        nbytes = lib.deft_flatten_plan_bytes(NB, P, Hq, Hkv)
        plan = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(8):  # the plan kernels alone, HIP events on the launching stream
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(4): call(md)  # keep the GPU busy (and its clocks up) right up to the plan kernels, as in a decode step
            e0.record()
            check(lib.deft_flatten_build_plan(*[t.data_ptr() for t in mdl], NB, P, Hq, Hkv, q.stride(0), q.stride(1), kb.stride(0),
                                              None, 0, 0, plan.data_ptr(), nbytes, s), "deft_flatten_build_plan")
            e1.record()
            # ... and once more right behind it: the same kernels with their code and the metadata WARM in the caches (what a lone
            # workgroup pays for cold instruction lines and cold tables is the difference)
            check(lib.deft_flatten_build_plan(*[t.data_ptr() for t in mdl], NB, P, Hq, Hkv, q.stride(0), q.stride(1), kb.stride(0),
                                              None, 0, 0, plan.data_ptr(), nbytes, s), "deft_flatten_build_plan")
            e2 = torch.cuda.Event(enable_timing=True); e2.record()
            torch.cuda.synchronize(); t_plan.append(e0.elapsed_time(e1) * 1e3); t_warm.append(e1.elapsed_time(e2) * 1e3)
    else:
        nd = [md.node_kv, md.node_kv_offset, md.node_kv_len, md.node_q, md.node_q_offset, md.node_q_len]
        NE, Pn, total_kv = md.node_kv_offset.shape[0], md.node_q.shape[0], md.node_kv.shape[0]
        nbytes = lib.deft_node_plan_bytes(NE, Pn, total_kv, Hq, Hkv)
        plan = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(4): call(md)
            e0.record()
            check(lib.deft_node_build_plan(*[t.data_ptr() for t in nd], NE, Pn, total_kv, Hq, Hkv, q.stride(0), q.stride(1), kb.stride(0),
                                           None, 0, 0, plan.data_ptr(), nbytes, s), "deft_node_build_plan")
            e1.record()
            check(lib.deft_node_build_plan(*[t.data_ptr() for t in nd], NE, Pn, total_kv, Hq, Hkv, q.stride(0), q.stride(1), kb.stride(0),
                                           None, 0, 0, plan.data_ptr(), nbytes, s), "deft_node_build_plan")
            e2 = torch.cuda.Event(enable_timing=True); e2.record()
            torch.cuda.synchronize(); t_plan.append(e0.elapsed_time(e1) * 1e3); t_warm.append(e1.elapsed_time(e2) * 1e3)
    print(json.dumps({"workload": name, "mode": w.mode, "out_sha": hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:16],
                      "plan_us": round(sorted(t_plan)[len(t_plan) // 2], 1) if t_plan else None,
                      "plan_us_warm": round(sorted(t_warm)[len(t_warm) // 2], 1) if t_warm else None,
                      "attention_us_cached_plan": round(sorted(t_att)[2], 1)}))
