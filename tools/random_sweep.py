#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): random chunk coordinates through the config-2 pipeline and random small regions through the full
pipeline, HIP path vs CPU oracle, bit for bit - far from the origin, across the border of the pruning domain (|block| = 32 768), lazily
and fully built rings.  tests/test_gpu_random.py runs a 20 s slice of it in the GPU suite (seed = hash of the library); this is the long form - run it on the
FINAL library and keep the log (profiles/):
    python tools/random_sweep.py [seed] [chunks] [regions]"""
import hashlib, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_binding import Oracle

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nchunks = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nregions = int(sys.argv[3]) if len(sys.argv) > 3 else 12
from random_draw import draw
sets, regions = draw(seed, nchunks, nregions)
pkg = importlib.import_module("mega-minecraft_amd")
print(f"random sweep: seed {seed}, {nchunks} chunks, {nregions} regions, lib_sha16 {hashlib.sha256(open(pkg.LIB_PATH, 'rb').read()).hexdigest()[:16]}", flush=True)
gen = pkg.MMGen(0); o = Oracle()
bad = 0
t0 = time.time()
for scale, coords in sets:
    out = gen.generate_chunks_no_erosion(gen.positions(coords))
    pos = o.positions(coords)
    hf, bw = o.heightfields(pos)
    layers = o.fix_backward(o.layers(pos, o.gather_heightfields(pos, hf), bw))
    cave = o.caves(pos, hf, bw)
    blocks = o.fill(pos, hf, bw, layers, cave)
    for name, ref in (("hf", hf), ("bw", bw), ("layers", layers), ("cave", cave), ("blocks", blocks)):
        got = out[name].cpu().numpy().reshape(ref.shape)
        same = np.array_equal(got.view(np.uint8), ref.view(np.uint8))
        if not same:
            bad += 1
            print(f"MISMATCH {name} at scale {scale}: {int((got != ref).sum())} elements, first chunk {coords[int(np.argwhere((got != ref).reshape(len(coords), -1).any(1))[0][0])]}")
    print(f"config-2 pipeline, {len(coords)} random chunks within +-{scale}: {'ok' if not bad else 'BAD'}   ({time.time() - t0:.0f} s)", flush=True)
for cx, cz, nx, nz, lazy in regions:
    ref = o.generate_region(cx, cz, nx, nz, erosion=True, features=True, decorators=True)
    got = gen.generate_region(cx, cz, nx, nz, want=("layers", "cave"), lazy_ring=lazy)
    ok = all(np.array_equal(got[k].cpu().numpy().reshape(ref[k].shape).view(np.uint8), ref[k].view(np.uint8)) for k in ("hf", "layers", "cave", "blocks"))
    ok = ok and got["erosion_passes"] > 0
    bad += 0 if ok else 1
    print(f"full pipeline, region ({cx},{cz}) {nx}x{nz} {'lazy' if lazy else 'full'} ring: {'ok' if ok else 'MISMATCH'}   ({time.time() - t0:.0f} s)", flush=True)
print("random sweep:", "ALL BIT-EXACT" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
