#!/usr/bin/env python3
"""One steady-state step out of a rocprofv3 kernel trace: tools/trace_one_step.py <kernel_trace.csv> [step index from the end, default 2]
Prints every mm:: kernel (and the runtime's fills / copies) between two consecutive k_heightfield launches, times relative to the first."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k1 = [i for i, r in enumerate(rows) if "k_heightfield" in r["Kernel_Name"]]
a, b = k1[-back - 1], k1[-back]
t0 = int(rows[a]["Start_Timestamp"])
end_prev = max(int(r["End_Timestamp"]) for r in rows[:a]) if a else t0
print(f"step period {(int(rows[b]['Start_Timestamp']) - t0) / 1e6:.3f} ms; previous step's last kernel ended {(t0 - end_prev) / 1e6:.3f} ms before this k_heightfield")
for r in rows[a:b]:
    n = r["Kernel_Name"]
    if not (n.startswith("mm::") or n.startswith("void mm::") or "rocclr" in n): continue
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{s:8.3f} {e:8.3f} {e - s:7.3f}  q={r.get('Queue_Id', '?'):>2}  {n.replace('void ', '').split('(')[0]}")
