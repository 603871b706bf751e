#!/usr/bin/env python3
"""GPU box: the HIP path's 65 536-chunk world [-128, 128)^2 (one region) against the ORACLE's per-chunk digests (tests/golden/make_world_digests.py ->
tests/golden/world_digests.npz), every chunk.  Prints the number of differing chunks and the first few positions; exit code 1 on any.
    python tools/check_world_digests.py [digests.npz]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("mega-minecraft_amd")
d = importlib.import_module("mega-minecraft_amd.distributed")
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "world_digests.npz")
cx0, cz0, gold = d.load_world_digests(path)
nz, nx = gold.shape
gen = pkg.MMGen(0)
got = d.chunk_digests(gen.generate_region(cx0, cz0, nx, nz)["blocks"], torch).cpu().numpy().reshape(nz, nx)
bad = (got != gold)
print(f"library {pkg.LIB_PATH}: world ({cx0},{cz0}) {nx}x{nz} = {nx * nz} chunks, {int(bad.sum())} chunks differ from the oracle's digests ({path})")
for z, x in list(zip(*bad.nonzero()))[:16]:
    print(f"  chunk ({cx0 + x},{cz0 + z})")
sys.exit(1 if bad.any() else 0)
