#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): random chunk coordinates through the config-2 pipeline and random small regions through the full
pipeline, HIP path vs CPU oracle, bit for bit - far from the origin, across the border of the pruning domain (|block| = 32 768), lazily
and fully built rings.  Not part of pytest (minutes of CPU time); run on the FINAL library and keep the log (profiles/):
    python tools/random_sweep.py [seed] [chunks] [regions]"""
import hashlib, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_binding import Oracle

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nchunks = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nregions = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rng = np.random.default_rng(seed)
pkg = importlib.import_module("mega-minecraft_amd")
print(f"random sweep: seed {seed}, {nchunks} chunks, {nregions} regions, lib_sha16 {hashlib.sha256(open(pkg.LIB_PATH, 'rb').read()).hexdigest()[:16]}", flush=True)
gen = pkg.MMGen(0); o = Oracle()
bad = 0
t0 = time.time()
for scale in (300, 5000, 60000, 2_000_000):
    coords = [tuple(int(v) for v in rng.integers(-scale, scale, 2)) for _ in range(nchunks // 4)]
    if scale == 5000:                                     # a quarter of these on the border of the pruning domain (chunk +-2048)
        for i in range(0, len(coords), 4):
            side = int(rng.integers(0, 4))
            edge = int(rng.choice([-2049, -2048, 2047, 2048]))
            other = int(rng.integers(-2200, 2200))
            coords[i] = (edge, other) if side < 2 else (other, edge)
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
for i in range(nregions):
    kind = i % 4
    scale = (400, 4000, 50000, 0)[kind]
    if kind == 3:                                         # straddling the border of the pruning domain
        edge = int(rng.choice([-2050, -2049, 2046, 2047]))
        other = int(rng.integers(-2100, 2100))
        cx, cz = (edge, other) if rng.integers(0, 2) else (other, edge)
    else:
        cx, cz = (int(v) for v in rng.integers(-scale, scale, 2))
    nx, nz = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    ref = o.generate_region(cx, cz, nx, nz, erosion=True, features=True, decorators=True)
    lazy = bool(i % 5)                                    # every fifth region with the ring built in full (mask 1)
    got = gen.generate_region(cx, cz, nx, nz, want=("layers", "cave"), lazy_ring=lazy)
    ok = all(np.array_equal(got[k].cpu().numpy().reshape(ref[k].shape).view(np.uint8), ref[k].view(np.uint8)) for k in ("hf", "layers", "cave", "blocks"))
    ok = ok and got["erosion_passes"] > 0
    bad += 0 if ok else 1
    print(f"full pipeline, region ({cx},{cz}) {nx}x{nz} {'lazy' if lazy else 'full'} ring: {'ok' if ok else 'MISMATCH'}   ({time.time() - t0:.0f} s)", flush=True)
print("random sweep:", "ALL BIT-EXACT" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
