#!/usr/bin/env python3
"""Replay the reference's three workloads through the attention path and print its latency table per mode
(attention-only: the model forward is synthetic, see deft_amd/replay.py).

  tools/replay.py --task reasoning --modes flatten node seq                      # synthetic ToT 4k prompt, 7x128 -> 42x64
  tools/replay.py --task reasoning --template .../Reasoning/sorting128ToT.json  # a file of the reference's dataset
  tools/replay.py --task reasoning --golden-template docmergeToT --max-gen-len 100000   # the same template from tests/golden/templates.json
  tools/replay.py --task speculative_decoding --modes node flatten seq --tree-size 64
  tools/replay.py --task few_shot --width 32 --prompt-len 4096 --max-gen-len 200
"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deft_amd import replay as rp
from deft_amd.utils.workloads import GEOMETRY

ap = argparse.ArgumentParser()
ap.add_argument("--task", default="reasoning", choices=sorted(rp.BRANCH_FUNCS))
ap.add_argument("--modes", nargs="+", default=["flatten", "node", "seq"])
ap.add_argument("--model", default="llama2-7b", choices=sorted(GEOMETRY))
ap.add_argument("--layers", type=int, default=None)
ap.add_argument("--template", default=None, help="a dataset/generation/** file of the reference repository")
ap.add_argument("--golden-template", default=None, metavar="NAME",
                help="reasoning: the node table of the reference's own template NAME (docmergeToT, keywordToT, set128ToT, sorting128ToT: "
                     "the first complete tree of that dataset file) as recorded in tests/golden/templates.json -- the file itself "
                     "belongs to the reference repository and is not on the GPU box")
ap.add_argument("--tree-index", type=int, default=0)
ap.add_argument("--prompt-len", type=int, default=None)
ap.add_argument("--max-gen-len", type=int, default=400)
ap.add_argument("--width", type=int, default=32)
ap.add_argument("--tree-size", type=int, default=64)
ap.add_argument("--sd-steps", type=int, default=100)
ap.add_argument("--pipelined", action="store_true", help="no per-step sync: host runs ahead of the GPU")
ap.add_argument("--no-warmup", action="store_true", help="do not run the first mode once untimed before the table")
ap.add_argument("--host-metadata", action="store_true", help="TreeMetadata by the host builder + one upload per step (round-1 path)")
ap.add_argument("--eager", action="store_true", help="the reference-shaped eager calls per step instead of deft_amd.DecodeSession (one hipGraph per structural epoch)")
ap.add_argument("--beam", default=None, metavar="WIDTH,DEPTH,LEN",
                help="reasoning: the shipped templates' shape instead of the ToT-50 tree -- the kept node branches into WIDTH candidates "
                     "every LEN steps, DEPTH levels (deft_amd.templates.synthetic_beam_template)")
ap.add_argument("--legacy", action="store_true", help="DecodeSession(incremental=False): metadata and plan rebuilt on every step (the round-5 loop)")
ap.add_argument("--win-tiles", type=int, default=None, help="DecodeSession(win_tiles=): overflow tiles per query chunk of a window plan")
ap.add_argument("--capture-after", default="auto", help="DecodeSession(capture_after=): 1, 2, ... or auto")
ap.add_argument("--reps", type=int, default=1, help="timed replays per mode; the table keeps the one with the least attention time and lists all "
                "(a 50 ms replay is doubled by ONE 50 ms stall of the box's driver: profiles/r6_slow_run_hunt.txt)")
ap.add_argument("--out", default=None)
a = ap.parse_args()
if a.host_metadata:
    import deft_amd.tree_cache as _tc
    _tc.DEVICE_METADATA = False
Hq, Hkv, D, L = GEOMETRY[a.model]
L = a.layers or L


def template():
    if a.task == "reasoning":
        if a.golden_template:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "templates.json")))
            return rp.TreeTemplate.from_node_table(gold["reasoning"][a.golden_template]["data"])
        if a.template:
            return rp.read_reasoning_file(a.template)[a.tree_index]
        if a.beam:
            return rp.synthetic_beam_template(*[int(x) for x in a.beam.split(",")])
        return rp.synthetic_reasoning_template()
    if a.task == "speculative_decoding":
        if a.template:
            t = rp.read_speculative_file(a.template)[a.tree_index]
            return t
        return rp.synthetic_speculative_template(a.tree_size, a.sd_steps)
    return rp.synthetic_few_shot_template(a.width)


rows = []
# The first replay of a process is ~200 us per step slower than any later one (module loads, pinned staging buffers, the
# driver's first allocations of every workspace size the growing tree asks for): run the first mode once, untimed.
# ... and the first replay of every MODE pays that mode's own first launches (the kernels of its plan and operator: ~100 ms once, three
# times a 100-step speculative-decoding replay -- profiles/r6_slow_run_hunt.txt): every mode runs once untimed, then the table.
n_warm = 0 if a.no_warmup else len(a.modes)
a_modes = (list(a.modes) if n_warm else []) + [m for m in a.modes for _ in range(max(a.reps, 1))]
best = {}
for idx, mode in enumerate(a_modes):
    tpl = template()
    prompt_len = a.prompt_len or rp.default_prompt_len(tpl, a.task, from_file=bool(a.template or a.golden_template))
    r = rp.TemplateReplay(Hq, Hkv, D, L, mode=mode, device="cuda", attention=True, session=False if a.eager else None,
                          capture_after=a.capture_after if a.capture_after == "auto" else int(a.capture_after), incremental=not a.legacy, win_tiles=a.win_tiles)
    rep = r.run(tpl, a.task, prompt_len, a.max_gen_len, max_rows=max(512, a.width, a.tree_size), pipelined=a.pipelined)
    s = rep.summary(); s["model"] = a.model; s["layers"] = L
    s["path"] = "session (one hipGraph per structural epoch)" if r.session else "eager calls"
    s["graph_captures"] = r.graph_captures; s["step_kinds"] = getattr(r, "step_kinds", None); s["pipelined"] = bool(a.pipelined); s["capture_after"] = a.capture_after
    s["wall_over_attention"] = round(s["wall_ms"] / max(s["attention_latency_ms"], 1e-9), 3)
    if idx < n_warm:
        del r
        torch.cuda.empty_cache()
        continue
    if a.reps > 1:  # (the least attention time of the mode's replays stands in the table; every repetition's time is listed with it)
        prev = best.get(mode)
        times = (prev["reps_attention_ms"] if prev else []) + [s["attention_latency_ms"]]
        if prev is None or s["attention_latency_ms"] < prev["attention_latency_ms"]:
            best[mode] = s
        best[mode]["reps_attention_ms"] = times
        if len(times) == a.reps:
            rows.append(best[mode])
            print(json.dumps(best[mode]), flush=True)
    else:
        rows.append(s)
        print(json.dumps(s), flush=True)
    del r
    torch.cuda.empty_cache()
base = next((x for x in rows if x["mode"] == "seq"), None)
print(f"\n{'mode':10s} {'steps':>6s} {'gen tok':>8s} {'attn ms':>10s} {'us/step':>9s} {'TPOT ms':>9s} {'md ms/step':>10s} {'KV-IO TB':>9s} {'vs seq':>7s}")
for s in rows:
    sp = f"{base['attention_latency_ms'] / s['attention_latency_ms']:.2f}x" if base else "-"
    print(f"{s['mode']:10s} {s['steps']:6d} {s['generated_tokens']:8d} {s['attention_latency_ms']:10.2f} {s['attention_us_per_step']:9.1f} "
          f"{s['attention_TPOT_ms_per_token']:9.4f} {s['metadata_ms'] / max(s['steps'], 1):10.3f} {s['KV_IO_TB']:9.3f} {sp:>7s}")
if a.out:
    json.dump(rows, open(a.out, "w"), indent=1)
