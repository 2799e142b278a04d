#!/usr/bin/env python3
"""(Round 6: DESIGN.md's table is now a hand-merged pair of such tables -- rounds 3-5 and 5-6, two boxes; this tool still PRINTS the table of one run,
but only writes it with --write.)
Rewrites DESIGN.md's round-over-round table (between the ROUNDS_TABLE markers) from a SAME-BOX run of tools/ab_rounds.sh -- every
round's own tree (prev/r3, prev/r4, the working tree) running its own bench.py on its own library, in turn, on one GPU box:

    tools/ab_rounds.sh "<workloads>" prev/r3 prev/r4 . > profiles/r5z_ab_rounds.txt ;  python tools/round_table.py profiles/r5z_ab_rounds.txt

(Round 4's table compared the driver's box of one round with the builder's box of the next: off by the box-to-box spread, VERDICT r4
weak #9.)  Cells: mean of the repetitions, us per layer of the captured 32-layer step."""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
LABEL = {"northstar_4kx32": "north-star 4k x 32 x 200, Llama-2-7B (headline)", "fewshot_1kx32": "1k x 32 x 200 (configs[1])",
         "medusa64_node": "Medusa-64 as the reference mocks it, DeFT-Node (configs[2])", "tot50_4k": "ToT-50, Llama-3-8B (configs[3])",
         "forest_8kx8": "8 trees of 8k x 8 as one batch, Llama-3-8B (configs[4] per GPU)", "forest_8kx8_single": "one 8k x 8 tree",
         "gqa_4kx32": "north-star tree on Llama-3-8B (GQA 4k x 32)", "northstar_4kx32_d64": "north-star tree at head_dim 64",
         "northstar_4kx32_node": "north-star tree through DeFT-Node", "northstar_4kx32_seq": "north-star tree, sequential comparator"}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
trees = []
for line in open(src):
    m = re.match(r"(\S+)\s+rep (\d+)\s+(\S+)\s+([\d.]+)\s+(\S+)", line)
    if not m:
        continue
    tree, _, wl, us, _ = m.groups()
    if tree not in trees:
        trees.append(tree)
    vals[wl][tree].append(float(us))
name = {"prev/r3": "round 3", "prev/r4": "round 4", "prev/r5": "round 5", ".": "this tree"}
t = ("**Rounds on ONE box** (`%s`: `tools/ab_rounds.sh`, every round's own tree and library in turn, mean of the repetitions; "
     "process to process ±0.3 µs), µs per layer of the captured 32-layer step:\n\n| workload | %s |\n|---|%s\n"
     % (os.path.relpath(src, ROOT), " | ".join(name.get(x, x) for x in trees), "---|" * len(trees)))
for wl, label in LABEL.items():
    if wl in vals:
        t += "| %s | %s |\n" % (label, " | ".join("%.2f" % (sum(vals[wl][x]) / len(vals[wl][x])) if vals[wl][x] else "-" for x in trees))
t += "\n"
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
s = re.sub(r"(<!-- ROUNDS_TABLE_BEGIN -->\n).*?(<!-- ROUNDS_TABLE_END -->\n)", lambda m: m.group(1) + t + m.group(2), s, flags=re.S)
if "--write" in sys.argv:
    open(p, "w").write(s)
print(t)
