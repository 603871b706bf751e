#!/usr/bin/env python3
"""Experiment (GPU box): whole-tile steps of R regions in flight at once, one host thread + one stream each - does the chip take the
latency-bound kernels of one step under the issue-bound kernels of another?    python tools/two_regions.py [regions] [steps]"""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
pkg = importlib.import_module("mega-minecraft_amd")
gens = [pkg.MMGen(0) for _ in range(R)]
streams = [torch.cuda.Stream() for _ in range(R)]
nx, nz = 64, 128
def run(i, steps):
    with torch.cuda.stream(streams[i]):
        for s in range(steps):
            gens[i].generate_region(-32 + 1000 * i, -64, nx, nz)
        streams[i].synchronize()
for i in range(R): run(i, 2)          # warm-up, one after the other
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(i, K)) for i in range(R)]
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{R} region(s) in flight: {R * K * nx * nz / dt:.0f} chunks/s, {1000 * dt / (R * K):.3f} ms per step")
