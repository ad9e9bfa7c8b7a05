#!/usr/bin/env python3
"""how much kernels of different streams overlapped in a rocprofv3 trace (rocpd sqlite)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select start, end, queue_id, stream_id from kernels order by start"))
events = sorted([(s, 1) for s, e, q, st in rows] + [(e, -1) for s, e, q, st in rows])
busy = depth_time = 0
depth, last = 0, events[0][0]
hist = {}
for t, d in events:
    if depth > 0:
        busy += t - last
        hist[depth] = hist.get(depth, 0) + (t - last)
    last = t
    depth += d
total = sum(e - s for s, e, q, st in rows)
span = rows[-1][1] - rows[0][0]
print(f"kernels {len(rows)}, queues {len(set(r[2] for r in rows))}, streams {len(set(r[3] for r in rows))}")
print(f"span {span / 1e6:.1f} ms, sum of kernel durations {total / 1e6:.1f} ms, GPU busy (union) {busy / 1e6:.1f} ms, mean concurrency while busy {total / busy:.2f}")
print("time at concurrency depth (ms):", {k: round(v / 1e6, 1) for k, v in sorted(hist.items())})
