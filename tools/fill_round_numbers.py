#!/usr/bin/env python3
"""(Re)write DESIGN.md's end-of-round paragraph from its template (tools/end_of_round_paragraph.md, placeholders <TAG>_*) and the
files tools/end_of_round.sh wrote:   tools/fill_round_numbers.py r4z [dir = gpurun_out/r4z]     (prints what it filled).
DESIGN.md holds either the markers <TAG>_PARAGRAPH / <TAG>_CEILTABLE (first fill) or an earlier fill of them between the
<!-- ... --> comments this tool leaves behind (refill)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
d = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", tag)
P = tag.upper() + "_"


def last_json(path):
    txt = open(path).read().strip()
    try:
        return json.loads(txt)
    except json.JSONDecodeError:
        return json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])


b = last_json(os.path.join(d, f"{tag}_bench_default.json"))
ow = b["other_workloads"]
wl = lambda n: f"{ow[n]['us_per_layer']}" + (f" ({ow[n]['stage1_hbm_frac']})" if ow[n].get("stage1_hbm_frac") else "")
e = b["end_to_end"]
c5 = b["cfg5_sharded_forest"]
pf = b["prefill"]["prompts"]
stats = open(os.path.join(d, f"{tag}_kernel_stats.txt")).read().splitlines()
avg = lambda key: next((l.split("|")[3].strip() for l in stats if key in l), "?")


def replay(name, mode=None):
    rows = last_json(os.path.join(d, f"{tag}_replay_{name}.json"))
    rows = rows if isinstance(rows, list) else [rows]
    r = next(x for x in rows if mode is None or x["mode"] == mode)
    return r


def per_step(r):
    return round(r["wall_ms"] / r["steps"], 2)


two = last_json(os.path.join(d, f"{tag}_bench_2rank_gloo_one_gpu.json"))


def opt_json(name):
    try:
        return last_json(os.path.join(d, f"{tag}_{name}.json"))
    except Exception:
        return None


eight, rccl = opt_json("bench_8rank_gloo_one_gpu"), opt_json("bench_rccl_world_size_1")
vals = {
    "VALUE": f"{b['value'] / 1e3:.2f}", "MS": b["ms_per_step"], "LAYER": b["attention_latency_us_per_layer"],
    "S1RP": avg("stage1_np_kernel"), "S1": b["roofline"]["avg_launch_us"], "MG": avg("merge_kernel"), "FRAC": b["roofline"]["frac"],
    "STEP": b["step_hbm_frac"], "TRAFFIC": f"{b['roofline']['traffic'] / 1e6:.2f}" if b["roofline"].get("traffic") else "n/a",
    "E2EE": e["eager"]["ms_per_step"], "E2E": e["graphed"]["ms_per_step"], "RATIO": e["graphed"]["over_frozen_step_at_mean_len"],
    "FROZEN": e["frozen_step_at_mean_len"]["ms_per_step"],
    "LEN1": wl("northstar_4kx32_len1"), "LEN400": wl("northstar_4kx32_len400"), "1K": wl("fewshot_1kx32"),
    "MEDUSA": wl("medusa64_node"), "TOTR": None, "TOT": wl("tot50_4k"), "GQA": wl("gqa_4kx32"), "F1": wl("forest_8kx8_single"),
    "CFG5E2E": c5["end_to_end"]["ms_per_step"], "CFG5": f"{c5['us_per_layer']} ({c5.get('stage1_hbm_frac')})",
    "NODE": wl("northstar_4kx32_node"), "NCHUNK": wl("northstar_4kx32_node_chunk") if "northstar_4kx32_node_chunk" in ow else "n/a",
    "MTN": wl("medusa64_tree_node") if "medusa64_tree_node" in ow else "n/a",
    "MTF": wl("medusa64_tree_flatten") if "medusa64_tree_flatten" in ow else "n/a", "SEQ": wl("northstar_4kx32_seq"), "D64": wl("northstar_4kx32_d64"),
    "PF4": pf["4096"]["TFLOPs"], "PF16": pf["16384"]["TFLOPs"],
    "PFD4": (pf.get("4096_head_dim_64") or {}).get("TFLOPs", "?"), "PFD16": (pf.get("16384_head_dim_64") or {}).get("TFLOPs", "?"),
    "CPU16": b["cpu_baseline"]["fp16"]["value"], "CPU": b["cpu_baseline"]["value"],
    "SDS": per_step(replay("speculative_64", "flatten")), "SDE": per_step(replay("speculative_64_eager", "flatten")),
    "SD": per_step(replay("speculative_64_pipelined", "flatten")),
    "FS": per_step(replay("few_shot_4kx32", "flatten")), "2R": two["ms_per_step"],
}
vals["TOTR"] = per_step(replay("reasoning_tot50", "flatten"))
# round 6: window plans against the rebuild-every-step loop, the run-level fraction, attention in situ
def opt(fn, default="n/a"):
    try:
        return fn()
    except Exception:
        return default


fr = b.get("few_shot_run") or {}
situ = b.get("in_situ") or {}
vals.update({
    "ACROSSL": opt(lambda: e["graphed_rebuild_every_step"]["over_frozen_steps_across_the_loop"]),
    "ACROSSF": opt(lambda: e["frozen_steps_across_the_loop"]["mean_ms_per_step"]),
    "ACROSS": opt(lambda: e["graphed"]["over_frozen_steps_across_the_loop"]),
    "E2EL": opt(lambda: e["graphed_rebuild_every_step"]["ms_per_step"]), "RATIOL": opt(lambda: e["graphed_rebuild_every_step"]["over_frozen_step_at_mean_len"]),
    "RUNFRACL": opt(lambda: fr["rebuild_every_step"]["run_hbm_frac"]), "RUNFRAC": opt(lambda: fr["window_plans"]["run_hbm_frac"]),
    "RUNMS": opt(lambda: fr["window_plans"]["ms_per_step"]),
    "SWEEP": opt(lambda: ", ".join(f"{v['step_hbm_frac']} at {k}" for k, v in fr["by_branch_len"].items())),
    "SITUB": opt(lambda: situ["attention_us_per_layer_back_to_back"]), "SITU": opt(lambda: situ["attention_us_per_layer_in_situ"]),
    "SDL": opt(lambda: per_step(replay("speculative_64_pipelined_rebuild", "flatten"))),
    "SDNL": opt(lambda: per_step(replay("speculative_64_pipelined_rebuild", "node"))), "SDN": opt(lambda: per_step(replay("speculative_64_pipelined", "node"))),
    "SDSL": opt(lambda: per_step(replay("speculative_64_rebuild", "flatten"))),
    "TOTPL": opt(lambda: per_step(replay("reasoning_tot50_pipelined_rebuild", "flatten"))), "TOTP": opt(lambda: per_step(replay("reasoning_tot50_pipelined", "flatten"))),
    "TOT3PL": opt(lambda: per_step(replay("reasoning_tot50_llama3_pipelined_rebuild", "flatten"))),
    "TOT3P": opt(lambda: per_step(replay("reasoning_tot50_llama3_pipelined", "flatten"))),
    "FSPL": opt(lambda: per_step(replay("few_shot_4kx32_pipelined_rebuild", "flatten"))), "FSP": opt(lambda: per_step(replay("few_shot_4kx32_pipelined", "flatten"))),
})
rf = b["roofline"]
vals.update({"CEIL": rf.get("ceiling_us", "n/a"), "OVERCEIL": rf.get("launch_over_ceiling", "n/a"),
             "8R": eight["ms_per_step"] if eight else "n/a", "RCCL": rccl["ms_per_step"] if rccl else "n/a"})
try:
    bm, bmp = replay("reasoning_beam10x8", "flatten"), replay("reasoning_beam10x8_pipelined", "flatten")
    vals.update({"BEAM": per_step(bm), "BEAMR": bm["wall_over_attention"], "BEAMP": per_step(bmp), "BEAMPR": bmp["wall_over_attention"]})
except Exception:
    vals.update({"BEAM": "n/a", "BEAMR": "n/a", "BEAMP": "n/a", "BEAMPR": "n/a"})
# the small-launch table of section 4b: B_algo, ceiling, stage 1, layer
rows = []
names = [("medusa64_node", "Medusa-64 as the reference mocks it, DeFT-Node (configs[2])"),
         ("medusa64_tree_node", "the Medusa token tree itself, DeFT-Node (configs[2] read literally)"),
         ("medusa64_tree_flatten", "the Medusa token tree itself, DeFT-Flatten"), ("tot50_4k", "ToT-50, Llama-3-8B (configs[3])"),
         ("forest_8kx8_single", "one 8k x 8 tree of configs[4]"), ("gqa_4kx32", "north-star tree on Llama-3-8B (GQA 4k x 32)"),
         ("northstar_4kx32_len1", "north-star tree at branch length 1"), ("fewshot_1kx32", "1k x 32 x 200 (configs[1])")]
for key, label in names:
    o = ow.get(key) or {}
    rows.append(f"| {label} | {o.get('algorithmic_MB', '?')} | {o.get('ceiling_us', '?')} | {o.get('stage1_us') or '(Node plan: no stage-1-only entry point)'} | {o.get('us_per_layer', '?')} |")
rows.append(f"| north-star 4k x 32 x 200 (headline) | {rf['algorithmic_bytes_per_launch'] / 1e6:.2f} | {rf.get('ceiling_us', '?')} | {rf['avg_launch_us']} | {b['attention_latency_us_per_layer']} |")
ceil_table = "\n".join(rows)
path = os.path.join(ROOT, "DESIGN.md")
txt = open(path).read()
tpl = os.path.join(ROOT, "tools", "end_of_round_paragraph.md")
def put(marker, body):
    """Replace `marker` -- or what an earlier run put in its place, between the comment pair -- with body."""
    global txt
    open_, close_ = f"<!-- {marker} -->", f"<!-- /{marker} -->"
    block = f"{open_}\n{body.rstrip()}\n{close_}"
    if open_ in txt:
        a, z = txt.index(open_), txt.index(close_) + len(close_)
        txt = txt[:a] + block + txt[z:]
    else:
        txt = txt.replace(marker, block, 1)


put(P + "CEILTABLE", ceil_table)
if os.path.exists(tpl):
    put(P + "PARAGRAPH", open(tpl).read())
# longest keys first: R3Z_E2EE before R3Z_E2E, R3Z_S1RP before R3Z_S1, R3Z_TOTR before R3Z_TOT, R3Z_CFG5E2E before R3Z_CFG5, ...
for k in sorted(vals, key=len, reverse=True):
    n = txt.count(P + k)
    txt = re.sub(re.escape(P + k) + r"(?![A-Z0-9])", str(vals[k]), txt)
    print(f"{P + k:14s} -> {vals[k]}   ({n} place(s))")
left = sorted(set(re.findall(re.escape(P) + r"[A-Z0-9]+", txt)))
open(path, "w").write(txt)
print("unfilled:", left or "none")
