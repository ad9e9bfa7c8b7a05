#!/usr/bin/env python3
"""per-kernel totals from a rocprofv3 results database (rocpd sqlite): python scripts/kernel_times.py gpurun_out/x/y_results.db [substring-of-kernel-for-sequence-dump]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
agg = collections.defaultdict(list)
for n, s, e in rows:
    agg[n.split("(")[0][:64]].append(e - s)
print(f"{'kernel':64s} {'calls':>6s} {'total us':>10s} {'avg us':>9s} {'max us':>9s}")
for n, v in sorted(agg.items(), key=lambda x: -sum(x[1])):
    print(f"{n:64s} {len(v):6d} {sum(v) / 1e3:10.1f} {sum(v) / len(v) / 1e3:9.2f} {max(v) / 1e3:9.2f}")
if len(sys.argv) > 2:
    seq = [(s, e) for n, s, e in rows if sys.argv[2] in n][-int(sys.argv[3]) if len(sys.argv) > 3 else -12:]
    print("last launches, us:", [round((e - s) / 1e3, 1) for s, e in seq])
    print("gaps, us:", [round((seq[i + 1][0] - seq[i][1]) / 1e3, 1) for i in range(len(seq) - 1)])
