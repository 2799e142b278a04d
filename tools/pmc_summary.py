#!/usr/bin/env python3
"""Summarise a `rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv` pass: per kernel, launches and
the mean counter value; FETCH_SIZE / WRITE_SIZE (KiB) are converted to bytes with the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (x2 for 16-byte-per-lane streams; verified on tools/probes/gather_bw).

  tools/pmc_summary.py <rocprof output dir> [--correction 2.0] > profiles/<name>.json"""
import argparse, csv, glob, json, os, sys
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--correction", type=float, default=2.0)
ap.add_argument("--note", default="")
a = ap.parse_args()
files = glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True)
if not files:
    sys.exit(f"no *counter_collection.csv under {a.dir}")
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"note": a.note or "rocprofv3 --pmc (own pass: --kernel-trace only). FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 "
                         "they count 64 B per 128 B request for 16 B/lane streams, hence the x%g correction." % a.correction,
       "correction": a.correction, "kernels": {}}
for k, counters in sorted(acc.items()):
    e = {}
    for c, vals in counters.items():
        e["launches"] = len(vals)
        e[f"{c}_avg"] = round(sum(vals) / len(vals), 1)
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            key = "hbm_read_bytes_per_launch" if c == "FETCH_SIZE" else "hbm_write_bytes_per_launch"
            e[key] = int(sum(vals) / len(vals) * 1024 * a.correction)
    out["kernels"][k[:120]] = e
print(json.dumps(out, indent=1))
