#!/usr/bin/env python3
"""Experiment (GPU box): whole-tile steps of R regions in flight at once, one stream each, ONE host thread (nothing in a step blocks the
host any more) - does the chip take the latency-bound kernels of one step under the issue-bound kernels of another?
    python tools/two_regions.py [regions] [steps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
pkg = importlib.import_module("mega-minecraft_amd")
gens = [pkg.MMGen(0) for _ in range(R)]
streams = [torch.cuda.Stream() for _ in range(R)]
nx, nz = 64, 128
outs = [None] * R
def step(i):
    with torch.cuda.stream(streams[i]):
        outs[i] = gens[i].generate_region(-32, -64, nx, nz)
for i in range(R):
    step(i); step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(K):
    step(s % R)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{R} region(s) in flight: {K * nx * nz / dt:.0f} chunks/s, {1000 * dt / K:.3f} ms per step")
if R > 1:
    assert all(torch.equal(outs[0]["blocks"], o["blocks"]) for o in outs[1:]), "regions differ"
