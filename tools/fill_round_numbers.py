#!/usr/bin/env python3
"""(Re)write DESIGN.md's end-of-round paragraph from its template (tools/end_of_round_paragraph.md, placeholders R3Z_*) and the
files tools/end_of_round.sh wrote:   tools/fill_round_numbers.py r3z [dir = gpurun_out/r3z]     (prints what it filled)"""
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
vals = {
    "VALUE": f"{b['value'] / 1e3:.2f}", "MS": b["ms_per_step"], "LAYER": b["attention_latency_us_per_layer"],
    "S1RP": avg("stage1_np_kernel"), "S1": b["roofline"]["avg_launch_us"], "MG": avg("merge_kernel"), "FRAC": b["roofline"]["frac"],
    "STEP": b["step_hbm_frac"], "TRAFFIC": f"{b['roofline']['traffic'] / 1e6:.2f}" if b["roofline"].get("traffic") else "n/a",
    "E2EE": e["eager"]["ms_per_step"], "E2E": e["graphed"]["ms_per_step"], "RATIO": e["graphed"]["over_frozen_step_at_mean_len"],
    "FROZEN": e["frozen_step_at_mean_len"]["ms_per_step"],
    "LEN1": wl("northstar_4kx32_len1"), "LEN400": wl("northstar_4kx32_len400"), "1K": wl("fewshot_1kx32"),
    "MEDUSA": wl("medusa64_node"), "TOTR": None, "TOT": wl("tot50_4k"), "GQA": wl("gqa_4kx32"), "F1": wl("forest_8kx8_single"),
    "CFG5E2E": c5["end_to_end"]["ms_per_step"], "CFG5": f"{c5['us_per_layer']} ({c5.get('stage1_hbm_frac')})",
    "NODE": wl("northstar_4kx32_node"), "SEQ": wl("northstar_4kx32_seq"), "D64": wl("northstar_4kx32_d64"),
    "PF4": pf["4096"]["TFLOPs"], "PF16": pf["16384"]["TFLOPs"],
    "CPU16": b["cpu_baseline"]["fp16"]["value"], "CPU": b["cpu_baseline"]["value"],
    "SDS": per_step(replay("speculative_64", "flatten")), "SDE": per_step(replay("speculative_64_eager", "flatten")),
    "SD": per_step(replay("speculative_64_pipelined", "flatten")),
    "FS": per_step(replay("few_shot_4kx32", "flatten")), "2R": two["ms_per_step"],
}
vals["TOTR"] = per_step(replay("reasoning_tot50", "flatten"))
path = os.path.join(ROOT, "DESIGN.md")
txt = open(path).read()
tpl = os.path.join(ROOT, "tools", "end_of_round_paragraph.md")
if os.path.exists(tpl):  # the paragraph between the two headings is replaced by the template, then filled
    t = open(tpl).read()
    head = t.split("**", 2)[1]  # "End of round N"
    a = txt.index("**" + head + "**")
    b = txt.index("**End of round", a + 4)
    txt = txt[:a] + t + txt[b:]
# longest keys first: R3Z_E2EE before R3Z_E2E, R3Z_S1RP before R3Z_S1, R3Z_TOTR before R3Z_TOT, R3Z_CFG5E2E before R3Z_CFG5, ...
for k in sorted(vals, key=len, reverse=True):
    n = txt.count(P + k)
    txt = re.sub(re.escape(P + k) + r"(?![A-Z0-9])", str(vals[k]), txt)
    print(f"{P + k:14s} -> {vals[k]}   ({n} place(s))")
left = sorted(set(re.findall(re.escape(P) + r"[A-Z0-9]+", txt)))
open(path, "w").write(txt)
print("unfilled:", left or "none")
