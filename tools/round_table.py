"""Rewrites DESIGN.md's round 3 -> round 4 table (between the R34_TABLE markers) from profiles/r4z_bench_default.json and the
driver's BENCH_r03.json numbers: python tools/round_table.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(open(os.path.join(ROOT, "profiles", "r4z_bench_default.json")).read().strip().splitlines()[-1])
ow = d["other_workloads"]
R3 = {"northstar": 34.85, "fewshot_1kx32": 26.36, "medusa64_node": 12.85, "tot50_4k": 22.32, "gqa_4kx32": 20.07,
      "forest_8kx8_single": 15.26, "cfg5": 56.13, "northstar_4kx32_d64": 24.6, "northstar_4kx32_node": 35.28}  # BENCH_r03.json
rows = [("north-star 4k x 32 x 200, Llama-2-7B (headline)", R3["northstar"], d["attention_latency_us_per_layer"]),
        ("1k x 32 x 200 (configs[1])", R3["fewshot_1kx32"], ow["fewshot_1kx32"]["us_per_layer"]),
        ("Medusa-64, DeFT-Node (configs[2])", R3["medusa64_node"], ow["medusa64_node"]["us_per_layer"]),
        ("ToT-50, Llama-3-8B (configs[3])", R3["tot50_4k"], ow["tot50_4k"]["us_per_layer"]),
        ("8 trees of 8k x 8 as one batch, Llama-3-8B (configs[4] per GPU)", R3["cfg5"], d["cfg5_sharded_forest"]["us_per_layer"]),
        ("one 8k x 8 tree", R3["forest_8kx8_single"], ow["forest_8kx8_single"]["us_per_layer"]),
        ("north-star tree on Llama-3-8B (GQA 4k x 32)", R3["gqa_4kx32"], ow["gqa_4kx32"]["us_per_layer"]),
        ("north-star tree at head_dim 64", R3["northstar_4kx32_d64"], ow["northstar_4kx32_d64"]["us_per_layer"]),
        ("north-star tree through DeFT-Node", R3["northstar_4kx32_node"], ow["northstar_4kx32_node"]["us_per_layer"])]
t = ("**Round 3 → round 4**, µs per layer of the captured 32-layer step (round 3: the driver's `BENCH_r03.json`; round 4: "
     "`profiles/r4z_bench_default.json`, another box of the pool -- box to box ±0.3):\n\n| workload | round 3 | round 4 |\n|---|---|---|\n")
for n, a, b in rows:
    t += "| %s | %.2f | %.2f |\n" % (n, a, b)
t += "\n"
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
s = re.sub(r"(<!-- R34_TABLE_BEGIN -->\n).*?(<!-- R34_TABLE_END -->\n)", lambda m: m.group(1) + t + m.group(2), s, flags=re.S)
open(p, "w").write(s)
print(t)
