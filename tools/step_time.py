#!/usr/bin/env python3
"""Developer aid: ms per step of the bench tile for a given set of stage flags (1 erosion, 2 features, 4 decorators), DAG schedule.
   usage: tools/step_time.py [flags ...]   e.g. tools/step_time.py 7 6     (what does the erosion branch cost the step?)"""
import importlib, os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("mega-minecraft_amd")
gen = pkg.MMGen(0)
nx, nz, cx0, cz0 = 64, 128, -32, -64
mask = (ctypes.c_uint8 * ((nx + 6) * (nz + 6)))(*([2] * ((nx + 6) * (nz + 6))))
for rep in range(2):
    for flags in [int(a) for a in sys.argv[1:]] or [7]:
        def step():
            gen.region_begin(cx0, cz0, nx, nz, flags, mask if flags & 2 else None)
            return gen.region_finish(nx, nz)
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 24
        for _ in range(K): out = step()
        torch.cuda.synchronize()
        print(f"flags {flags}: {1000 * (time.perf_counter() - t0) / K:.3f} ms per step", flush=True)
