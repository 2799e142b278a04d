#!/usr/bin/env python3
"""Dump the per-kernel summary of a rocprofv3 results.db (--kernel-trace --stats) as text."""
import glob, sqlite3, sys
path = sys.argv[1]
dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
for db in dbs:
    con = sqlite3.connect(db)
    print(f"# {db}\n# name | calls | total_us | avg_us | pct")
    for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{name[:110]} | {calls} | {total:.1f} | {avg:.3f} | {pct:.2f}")
