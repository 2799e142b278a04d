#!/usr/bin/env python3
"""Randomised check of the captured decode loop: random multi-level trees (or batches of trees as one tree object), random cuts
and branches between runs of decode steps; deft_amd.DecodeSession (hipGraphs per structural epoch) against the eager path
(tree.alloc + TreeMetadata.from_tree_cache + DeFTAttention) at every step: pool bytes and page tables bit for bit; outputs BIT FOR
BIT for legacy sessions (incremental=False), within 1e-3 + 2^-10 |ref| (one fp16 step of the value: both sides are ROUNDED results) for window-plan sessions (incremental=True, csrc/window.h:
the same keys in another partition).
   tools/fuzz_session.py [seconds] [seed] [incremental: 0 | 1 | mix (default)]"""
import os, sys, time, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import deft_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
inc_arg = sys.argv[3] if len(sys.argv) > 3 else "mix"
t_end = time.time() + budget
runs = steps = captures = sd_runs = inc_runs = 0
kinds = {"upload": 0, "legacy": 0, "replan": 0, "patch": 0}
worst = 0.0


def agree(out, ref, inc, tag):
    global worst
    if not inc:
        if not torch.equal(out, ref) and os.environ.get("FUZZ_VERBOSE"):
            e = (out.float() - ref.float()).abs()
            print("MISMATCH (legacy session, bit for bit)", tag, "max_q_len", sess.max_q_len, "max err", float(e.max()),
                  "rows", e.view(e.shape[0], -1).amax(dim=1).nonzero().flatten().tolist(), flush=True)
        assert torch.equal(out, ref), tag
        return
    err = (out.float() - ref.float()).abs()
    worst = max(worst, float(err.max()) if err.numel() else 0.0)
    ok = bool((err <= 1e-3 + ref.float().abs() * 2.0 ** -10).all())
    if not ok and os.environ.get("FUZZ_VERBOSE"):
        bad = (err > 1e-3 + ref.float().abs() * 2.0 ** -10)
        rows = bad.view(bad.shape[0], -1).any(dim=1).nonzero().flatten().tolist()
        print("MISMATCH", tag, "max_q_len", sess.max_q_len, "inc", sess.incremental, "rows", rows, "of", bad.shape[0], "W", sess.W, "kinds", sess.step_kinds, "max err per bad row", [round(float(err[r].max()), 5) for r in rows[:8]], flush=True)
    assert ok, (tag, float(err.max()))


while time.time() < t_end:
    Hq, Hkv = rng.choice([(32, 32), (32, 8), (8, 2), (4, 4), (16, 1)])
    # (head_dim 64 -- head pairs on the tile-parallel kernel -- where the geometry allows it: an even number of KV heads)
    D, layers = (64 if Hkv % 2 == 0 and rng.random() < 0.25 else 128), rng.choice([1, 2])
    mode = rng.choice(["flatten", "flatten", "node", "node_chunk"])
    # (--mode node_chunk = DeFT-Node with MAX_BLOCK_LEN = 128, examples/run_DeFT_llama_paged.py:145-150: the metadata cuts every node
    #  into 128-token entries, the Node plan folds them again -- round 5)
    deft_amd.BLOCK_CONFIG["MAX_BLOCK_LEN"] = 128 if mode == "node_chunk" else -1
    smode = "node" if mode == "node_chunk" else mode
    size = 1 << 17 if Hkv <= 8 else 1 << 16
    g = torch.Generator(device="cuda").manual_seed(rng.randint(0, 10 ** 6))
    kv_init = torch.randn((layers, size, 2, Hkv, D), dtype=torch.float16, device="cuda", generator=g)
    forest = rng.random() < 0.3
    prompts = [rng.choice([1, 5, 127, 128, 129, 300, 1000, 3000]) for _ in range(rng.randint(2, 4) if forest else 1)]
    widths = [rng.randint(1, 5) for _ in range(rng.randint(1, 3))]
    lens = [rng.choice([1, 2, 7, 40, 130]) for _ in widths]
    trees = []
    for _ in range(2):
        req = deft_amd.ReqToTokenPool(512, 8192, device="cuda")
        pool = deft_amd.TokenToKVPool(size, torch.float16, Hkv, D, layers, device="cuda")
        tree = deft_amd.TreeCache(torch.float16, Hkv, D, layers, req, pool, None, True, False)
        if forest:
            tree.init_forest([torch.arange(n, dtype=torch.int32) for n in prompts])
        else:
            tree.init_prompt(torch.arange(prompts[0], dtype=torch.int32))
        pool._storage.copy_(kv_init)
        trees.append((tree, pool))
    (te, pe), (ts, ps) = trees
    cap = 512
    q = torch.randn((layers, cap, Hq * D), dtype=torch.float16, device="cuda", generator=g)
    k = torch.randn((layers, cap, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    v = torch.randn((layers, cap, Hkv * D), dtype=torch.float16, device="cuda", generator=g)
    nq_now = [1]
    inc = {"0": False, "1": True}.get(inc_arg, rng.random() < 0.6)
    # (round 6: other max_q_len than the default 32 now and then -- query chunks of 1 / 7 / 16 rows (not more than 32: the operators' query tile, as the reference's BLOCK_M): the reference's knob,
    #  tree_cache.py:619; a window plan's overflow regions follow the chunks)
    # (window-plan sessions only: with one query per chunk a query has dozens of partial rows, and the cooperative merge's slices depend on
    #  the row CAPACITY, which differs between a session and the eager call -- the last bit of a few outputs: legacy sessions keep 32 and
    #  their bit-for-bit comparison)
    mq = rng.choice([32, 32, 32, 32, 16, 7, 1]) if inc else 32
    sess = deft_amd.DecodeSession(ts, Hq, Hkv, D, layers, lambda l: (q[l, : nq_now[0]], k[l, : nq_now[0]], v[l, : nq_now[0]]), mode=smode,
                                  incremental=inc, win_tiles=rng.choice([None, None, 1, 2, 3]) if inc else None, max_q_len=mq)
    sess.debug = bool(os.environ.get("FUZZ_VERBOSE"))
    attn = [deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, l) for l in range(layers)]
    fmode = deft_amd.forward_mode_from_cli(mode)  # (sets BLOCK_CONFIG["MAX_BLOCK_LEN"] for the node modes the same way)

    def both(nsteps):
        global steps
        # Half of the runs WITHOUT a synchronisation between the session's steps: the host runs ahead of the GPU, and the session's
        # step t+1 head (its own stream) overlaps step t's layers -- the outputs of every step are cloned in stream order and
        # compared at the end of the run, pool bytes and page tables then too.
        if nsteps > 1 and rng.random() < 0.5:
            refs, outs, n = [], [], 0
            for _ in range(nsteps):
                for leaf in te.leaves.values():
                    leaf.append_token(7)
                upd = te.alloc()
                md = deft_amd.TreeMetadata.from_tree_cache(te, max_q_len=mq)
                deft_amd.register_tree_metadata(md)
                n = md.query_num
                refs.append([attn[l](q[l, :n], k[l, :n], v[l, :n], deft_amd.InputMetadata(fmode, upd, pe)).clone() for l in range(layers)])
            torch.cuda.synchronize()
            nq_now[0] = n
            for _ in range(nsteps):
                for leaf in ts.leaves.values():
                    leaf.append_token(7)
                out = sess.step()
                outs.append([out[l][:n].clone() for l in range(layers)])
            torch.cuda.synchronize()
            for i in range(nsteps):
                for l in range(layers):
                    agree(outs[i][l], refs[i][l], inc, ("lagged", runs, steps + i, l, mode, Hq, Hkv, prompts, widths, lens))
            assert torch.equal(pe._storage, ps._storage), ("lagged", runs, steps)
            assert torch.equal(te.req_to_token_pool.req_to_token, ts.req_to_token_pool.req_to_token), ("lagged", runs, steps)
            steps += nsteps
            return
        for _ in range(nsteps):
            for tree in (te, ts):
                for leaf in tree.leaves.values():
                    leaf.append_token(7)
            upd = te.alloc()
            md = deft_amd.TreeMetadata.from_tree_cache(te, max_q_len=mq)
            deft_amd.register_tree_metadata(md)
            n = md.query_num
            nq_now[0] = n
            ref = [attn[l](q[l, :n], k[l, :n], v[l, :n], deft_amd.InputMetadata(fmode, upd, pe)) for l in range(layers)]
            out = sess.step()
            torch.cuda.synchronize()
            for l in range(layers):
                agree(out[l][:n], ref[l], inc, (runs, steps, l, mode, Hq, Hkv, prompts, widths, lens))
            assert torch.equal(pe._storage, ps._storage), (runs, steps)
            assert torch.equal(te.req_to_token_pool.req_to_token, ts.req_to_token_pool.req_to_token), (runs, steps)
            steps += 1

    for wd, ln in zip(widths, lens):
        for tree in (te, ts):
            for leaf in sorted(tree.leaves.values(), key=lambda n: n.id):
                if len(tree.leaves) + wd - 1 <= 60:
                    tree.branch(leaf, wd)
        both(ln)
    for _ in range(rng.randint(0, 2)):  # structural changes: cut some leaves, branch one, then decode on
        for tree in (te, ts):
            lv = sorted(tree.leaves.values(), key=lambda n: n.id)
            r2 = random.Random(runs)
            for leaf in r2.sample(lv, min(len(lv) - 1, r2.randint(0, 2))):
                tree.cut(leaf)
            lv = sorted(tree.leaves.values(), key=lambda n: n.id)
            tree.branch(lv[0], r2.randint(2, 3))
        both(rng.choice([3, 20, 140]))
    # speculative-decoding steps (branch_func_example.py:420-437): some leaves' slots are squeezed into an inner node -- the root,
    # or the parent of the first leaf -- and every leaf's KV is released, between single decode steps.  Absorbed by the epoch
    # (the native tree's journal, replayed by the step's first kernel) after the first one, which finds the node without room.
    if rng.random() < 0.5:
        cap_before = sess.captures
        sd_steps = rng.choice([3, 12, 40])
        for it in range(sd_steps):
            for tree in (te, ts):
                r2 = random.Random(runs * 1000 + it)  # (the same draws for both trees)
                lv = sorted(tree.leaves.values(), key=lambda n: n.id)
                target = tree.root if (r2.random() < 0.7 or lv[0].parent is None) else lv[0].parent
                if len(target.kv_indices) == 0 and target is not tree.root:
                    target = tree.root
                before = len(target.kv_indices)
                for leaf in r2.sample(lv, min(len(lv), r2.randint(0, 4))):
                    tree.merge_nodes(target, leaf, pruneB_flag=False)
                # (round 5: now and then the reset is left out -- tree_cache.py:300-325 allows a merge without it; the leaves keep
                #  their slots, take another at the next step, and a later merge hands the target a slot it already holds: the
                #  state that corrupted the device copy's slot list before the upper-bound insertion)
                if r2.random() < 0.8:
                    tree.reset_nodes_KV(lv, len(target.kv_indices) - before)
            both(1)
        assert sess.captures - cap_before <= (3 if not inc else 9) + sd_steps // 200, (sess.captures - cap_before, sd_steps)  # not one epoch per step
        sd_runs += 1
    captures += sess.captures
    assert sess.device_errors() == 0, ("device error flags", sess.device_errors(), runs)
    inc_runs += int(inc)
    for kk in kinds:
        kinds[kk] += sess.step_kinds[kk]
    deft_amd.unregister_tree_metadata()
    deft_amd.BLOCK_CONFIG["MAX_BLOCK_LEN"] = -1
    runs += 1
print(f"session fuzz ok: {runs} trees ({sd_runs} with speculative-decoding merge / reset steps, {inc_runs} on window plans), {steps} steps "
      f"equal to the eager path (legacy sessions bit for bit; window plans worst |err| {worst:.2e}), {captures} graph captures, step kinds {kinds}")
