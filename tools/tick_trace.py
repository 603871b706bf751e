#!/usr/bin/env python3
"""One streaming tick out of a rocprofv3 kernel trace of `mmgen_region_terrain_demo --bench`:
   tools/tick_trace.py <rocprofv3 output dir> [index of the region call, default 20]
Region calls are told apart by their k_heightfield<true> launch (the region path's gathered variant; the per-stage mirror at the end of the
demo uses <false>).  Prints the period of the first 60 calls and every kernel / runtime copy between call i and call i + 1."""
import csv, glob, os, sys

f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=lambda p: -os.path.getsize(p))[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k1 = [i for i, r in enumerate(rows) if "k_heightfield<true>" in r["Kernel_Name"]]
print(len(k1), "region calls; ms between them:", [round((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e6, 2) for a, b in zip(k1, k1[1:])][:60])
a, b = k1[which], k1[which + 1]
t0 = int(rows[a]["Start_Timestamp"])
seg = rows[a:b]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6
print(f"call {which}: period {(int(rows[b]['Start_Timestamp']) - t0) / 1e6:.3f} ms, {len(seg)} kernels / copies, sum of their durations {busy:.3f} ms")
for r in seg:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"  {s:7.3f} {e:7.3f} {e - s:6.3f}  {r['Kernel_Name'].replace('void ', '').split('(')[0][:44]}")
