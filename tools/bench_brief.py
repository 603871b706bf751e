#!/usr/bin/env python3
"""Developer aid: run bench.py (arguments passed through) and print the headline, the serial pass and the per-kernel milliseconds only."""
import json, subprocess, sys
r = subprocess.run([sys.executable, "bench.py", "--cpu-side", "0"] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in r.stdout.splitlines() if l.startswith('{"metric')]
if not line:
    print(r.stdout[-2000:], r.stderr[-2000:]); sys.exit(1)
j = json.loads(line[-1])
print("ms_per_step", j["ms_per_step"], "median", j.get("ms_per_step_median"), "value", j["value"])
for k in ("serial_pass", "kernels_ms_sum"):
    if k in j: print(k, j[k])
km = j.get("kernels_ms") or {}
print("  ".join(f"{k}={v:.3f}" if isinstance(v, (int, float)) else f"{k}={v}" for k, v in sorted(km.items(), key=lambda kv: -kv[1] if isinstance(kv[1], (int, float)) else 0)))
print("roofline", json.dumps(j.get("roofline"))[:600])
