#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace csv (*_kernel_trace.csv) and prints, for the last traced step-sized window, which kernels ran when
(per queue / stream) and how much of the wall time had 1, 2, 3+ kernels in flight: the check that the region's stage DAG overlaps.
    python tools/trace_overlap.py <kernel_trace.csv> [n_last_kernels]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
print("queues:", sorted({r.get("Queue_Id", "?") for r in rows}), "streams:", sorted({r.get("Stream_Id", "?") for r in rows}))
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e6:9.3f} {e / 1e6:9.3f} {(e - s) / 1e6:8.3f} ms  q={r.get('Queue_Id', '?'):>3} st={r.get('Stream_Id', '?'):>3}  {r['Kernel_Name'][:60]}")
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth, last, hist = 0, ev[0][0], defaultdict(int)
for t, d in ev:
    hist[min(depth, 3)] += t - last
    last = t
    depth += d
tot = sum(hist.values())
print("time with k kernels in flight:", {k: f"{v / 1e6:.2f} ms ({100 * v / tot:.0f} %)" for k, v in sorted(hist.items())})
