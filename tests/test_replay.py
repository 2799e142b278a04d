"""Template replay (deft_amd/replay.py): the template model is pinned bit-exactly on what the reference's own
data_loader derives (tests/golden/templates.json, tools/gen_golden_templates.py); the decode loop is exercised on CPU
(tree / page table / metadata only) and on the GPU with every step's attention checked against fp64 truth."""
import json
import os

import numpy as np
import pytest
import torch

import deft_amd
from deft_amd import replay as rp

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "templates.json")))


@pytest.mark.parametrize("name", sorted(GOLD["reasoning"]))
def test_template_records_match_reference(name):
    g = GOLD["reasoning"][name]
    tree = rp.TreeTemplate.from_node_table(g["data"])
    assert tree.node_num == g["node_num"] and tree.max_depth == g["max_depth"] and tree.max_width == g["max_width"]
    assert {str(k): {str(p): c for p, c in v.items()} for k, v in tree.branch_record.items()} == g["branch_record"]
    assert {str(k): v for k, v in tree.prune_record.items()} == g["prune_record"]
    assert {str(k): v for k, v in tree.width_per_depth.items()} == g["width_per_depth"]


def test_load_trees_and_prompts_from_files(tmp_path):
    g = GOLD["reasoning"]["docmergeToT"]
    p = tmp_path / "toy.json"
    p.write_text(json.dumps([{"incompleted": True, "prompt": "x", "data": g["data"]},
                             {"incompleted": False, "prompt": "y", "data": g["data"]}]))
    trees = rp.read_reasoning_file(str(p))
    assert len(trees) == 1 and trees[0].prompt == "y" and trees[0].node_num == g["node_num"]  # incompleted skipped
    sd = GOLD["speculative"]["tree_size64"]
    p2 = tmp_path / "sd.json"
    p2.write_text(json.dumps({"Tree_ID": 1, "Tree_Structure": [], "Token_Tree_size": sd["Token_Tree_size"],
                              "Records": [{"prompt": "q", "Accept_length": sd["Accept_length_0"]}]}))
    t = rp.read_speculative_file(str(p2))[0]
    assert t.node_num == sd["node_num"] == 64 and t.accept_lengths == sd["Accept_length_0"]
    import random
    rp.fit_accept_lengths(t, 50, random.Random(0))
    assert sum(t.accept_lengths) == 50 and t.accept_lengths[:3] == sd["Accept_length_0"][:3]
    with pytest.raises(NotImplementedError):
        rp.read_reasoning_file("x.csv")
    # what tools/replay.py does when it is given a template file and no --prompt-len (it read a `.root` that
    # TreeTemplate does not have): the root's token count of a reasoning FILE; BASELINE's shapes otherwise
    root_tokens = int(trees[0].value[0])
    assert rp.default_prompt_len(trees[0], "reasoning", from_file=True) == (root_tokens if root_tokens > 0 else 4096)
    assert rp.default_prompt_len(trees[0], "reasoning", from_file=False) == 4096
    assert rp.default_prompt_len(t, "speculative_decoding", from_file=True) == 1016
    assert rp.default_prompt_len(rp.synthetic_few_shot_template(4), "few_shot") == 4096


def _cpu_replay(task, template, prompt_len, max_gen_len, mode="flatten"):
    r = rp.TemplateReplay(8, 2, 128, layers=1, mode=mode, device="cpu", attention=False)
    return r, r.run(template, task, prompt_len, max_gen_len)


def test_reasoning_replay_follows_the_template_and_frees_everything():
    tpl = rp.synthetic_reasoning_template(widths=(3, 2), lens=(5, 4))
    r, rep = _cpu_replay("reasoning", tpl, prompt_len=40, max_gen_len=64)
    # 3 nodes x 5 tokens, then 6 x 4: live leaves 3 for 5 steps ... the replay stops when the root is released
    nqs = [int(s["nq"]) for s in rep.per_step]
    assert nqs[:4] == [3, 3, 3, 3] and max(nqs) == 6 and rep.steps == len(nqs)
    assert rep.generated_tokens == 3 * 5 + 6 * 4
    assert len(r.tree.leaves) == 0 and len(r.tree.nodes) == 0  # every node cut, root included
    assert int((r.pool.mem_state != 0).sum()) == 0  # all KV slots back in the pool


def test_beam_template_has_the_shipped_files_shape():
    """width 10 per level, the kept node branching again (dataset/generation/Reasoning/*: 31 / 61 / 91 / 101 lifetime nodes)."""
    tpl = rp.synthetic_beam_template(width=10, depth=6, seg_len=8)
    assert tpl.node_num == 61 and tpl.max_width == 10 and tpl.max_depth == 6
    assert sorted(tpl.branch_record) == [0, 8, 16, 24, 32, 40]  # a branch every 8 steps
    r, rep = _cpu_replay("reasoning", tpl, prompt_len=30, max_gen_len=1000)
    assert rep.steps == 6 * 8


def test_reference_template_replay_runs_to_completion():
    g = GOLD["reasoning"]["docmergeToT"]
    tpl = rp.TreeTemplate.from_node_table(g["data"])
    r, rep = _cpu_replay("reasoning", tpl, prompt_len=int(tpl.value[0]), max_gen_len=100000)
    assert rep.generated_tokens == int(tpl.value[1:].sum())
    assert len(r.tree.nodes) == 0 and int((r.pool.mem_state != 0).sum()) == 0
    assert 1 <= rep.summary()["max_live_leaves"] <= tpl.node_num  # leaves of several depths are live at once


def test_speculative_replay_squeezes_accepted_tokens_into_the_root():
    tpl = rp.synthetic_speculative_template(tree_size=8, steps=6, accept=(1, 3), seed=3)
    r, rep = _cpu_replay("speculative_decoding", tpl, prompt_len=30, max_gen_len=100, mode="node")
    acc = tpl.accept_lengths
    assert rep.steps == len(acc)  # iterations 1 .. len(acc); the branch function stops at iter == len(acc) (:383-394)
    assert len(r.tree.root.kv_indices) == 30 + sum(acc[1:])  # branch_func_example.py:436-440
    assert all(int(s["nq"]) == 8 for s in rep.per_step)


def test_few_shot_replay_counts():
    tpl = rp.synthetic_few_shot_template(width=5)
    r, rep = _cpu_replay("few_shot", tpl, prompt_len=16, max_gen_len=7)
    assert rep.steps == 6 and rep.decoded_rows == 30 and rep.generated_tokens == 5 * 6  # (the last iteration appends nothing, :24-33)
    md = deft_amd.TreeMetadata.from_tree_cache(r.tree, device="cpu")
    assert md.query_num == 5 and md.total_kv_len == 16 + 5 * 6


@pytest.mark.gpu
@pytest.mark.parametrize("session", [True, False])
@pytest.mark.parametrize("task,mode", [("reasoning", "flatten"), ("reasoning", "node"), ("reasoning", "node_chunk"), ("reasoning", "seq"),
                                       ("speculative_decoding", "node"), ("speculative_decoding", "flatten"), ("few_shot", "flatten")])
def test_replay_attention_matches_truth_every_step(task, mode, session):
    """Run a small template with attention on the GPU and check, at every step, the output of layer 0 against fp64
    per-leaf attention over the leaf's page-table row (what the tree holds at that step) -- through the captured session
    (one hipGraph per structural epoch; speculative-decoding steps stay inside their epoch) and through the eager calls."""
    if mode == "seq" and session:
        pytest.skip("the sequential comparator has no session")
    Hq, Hkv, D = 8, 2, 128
    tpl = {"reasoning": rp.synthetic_reasoning_template(widths=(3, 2), lens=(6, 5)),
           "speculative_decoding": rp.synthetic_speculative_template(tree_size=12, steps=9, accept=(1, 3), seed=1),
           "few_shot": rp.synthetic_few_shot_template(width=6)}[task]
    r = rp.TemplateReplay(Hq, Hkv, D, layers=2, mode=mode, device="cuda", attention=True, session=session)
    assert r.session == session
    seen = {"steps": 0}

    def checked(tree, q, out):
        torch.cuda.synchronize()
        leaves = sorted(tree.leaves.values(), key=lambda n: n.id)
        kv = tree.token_to_kv_pool.kv_data[0].float().cpu().numpy().astype(np.float64)
        qn = q.view(-1, Hq, D).float().cpu().numpy().astype(np.float64)
        on = out.view(-1, Hq, D).float().cpu().numpy()
        assert on.shape[0] == len(leaves)
        for i, lf in enumerate(leaves):
            slots = tree.leaf_path_slots(lf)
            for hq in range(Hq):
                kh = hq // (Hq // Hkv)
                s = kv[slots, 0, kh] @ qn[i, hq] / np.sqrt(D)
                p = np.exp(s - s.max())
                ref = (p / p.sum()) @ kv[slots, 1, kh]
                assert np.abs(on[i, hq] - ref).max() < 5e-4
        seen["steps"] += 1

    r.step_hook = checked
    rep = r.run(tpl, task, prompt_len=200, max_gen_len=12)
    assert seen["steps"] == rep.steps > 3 and rep.attention_ms > 0
    if session and task == "speculative_decoding":
        # epochs: the branch into leaves; the first merge into a root without room.  Not one per step (window plans: a replan graph
        # and a patch graph per epoch).
        assert r.graph_captures <= 4 and rep.steps >= 8


@pytest.mark.gpu
def test_random_replays_match_truth_every_step():
    """A short, seeded run of tools/fuzz_replay.py: random templates / geometries / modes, every step of layer 0 against
    fp64 attention over the leaf's page-table row."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_replay.py"), "12", "7"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
