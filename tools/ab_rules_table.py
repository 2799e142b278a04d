"""Table of a tools/ab_rules.sh log: one row per workload, one column per library, us per layer of every repetition."""
import collections
import re
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
libs = []
for line in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+rep (\d)\s+(\S+)\s+([\d.]+)\s+(\S+)", line)
    if not m:
        continue
    lib = m.group(1).replace("libdeft_amd_rules_", "").replace("libdeft_amd.so", "SHIPPED").replace(".so", "")
    if lib not in libs:
        libs.append(lib)
    rows[m.group(3)][lib].append(float(m.group(4)))
print("%-22s" % "us per layer" + "".join("%-14s" % x for x in libs))
for wl, d in rows.items():
    print("%-22s" % wl + "".join("%-14s" % "/".join("%.2f" % v for v in d.get(x, [])) for x in libs))
