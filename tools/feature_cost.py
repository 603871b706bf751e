#!/usr/bin/env python3
"""Per-feature cost picture of k_apply_features on the bench tile: placements per feature (device lists) x measured cost of one
rasteriser evaluation (k_feature_box over a box of the feature's own extent)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mega-minecraft_amd")
gen = pkg.MMGen(0)
t = gen.debug_tables()
nx, nz = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 128
gen.region_begin(-nx // 2, -nz // 2, nx, nz, 7)
b = gen.region_placement_buffers()
cnt = b["counts"].cpu().numpy(); fp = b["fp"].cpu().numpy(); cfp = b["cfp"].cpu().numpy()
gen.region_finish(nx, nz)
sf = np.zeros(21, np.int64); cf = np.zeros(10, np.int64)
for c in range(cnt.shape[0]):
    sf += np.bincount(fp[c, :cnt[c, 0], 0] & 255, minlength=21); cf += np.bincount(cfp[c, :min(cnt[c, 1], 1024), 0] & 255, minlength=10)
names = "NONE SPHERE CORAL KELP ICEBERG ACACIA REDWOOD CYPRESS BIRCH PINE_TREE PINE_SHRUB RAFFLESIA LARGE_JUNGLE SMALL_JUNGLE TINY_JUNGLE MED_PURPLE_MUSH PURPLE_MUSH MED_CRYSTAL CRYSTAL PALM CACTUS".split()
cnames = "NONE T1 T2 CAVE_VINE GLOWSTONE STORMLIGHT CEIL_STORMLIGHT CRYSTAL_PILLAR WARPED_FUNGUS AMBER_FUNGUS".split()
rows = []
for cave, counts, nm, reach, bounds in ((False, sf, names, t["feature_reach"], t["feature_bounds"]), (True, cf, cnames, t["cave_feature_reach"], t["cave_feature_bounds"])):
    for f in range(1, len(nm)):
        r = int(reach[f]); lo, hi = int(bounds[f, 0]), int(bounds[f, 1])
        lh = 12 if cave else 0
        size = (2 * r + 1, hi - lo + 1 + lh, 2 * r + 1)
        fpos = (100, 90, 100)
        nvox = size[0] * size[1] * size[2]
        reps = max(1, int(2e7 // nvox))
        gen.debug_feature_box(cave, f, fpos, lh, (fpos[0] - r, fpos[1] + lo, fpos[2] - r), size)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            gen.debug_feature_box(cave, f, fpos, lh, (fpos[0] - r, fpos[1] + lo, fpos[2] - r), size)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        ns_per_eval = dt / nvox * 1e9
        rows.append((("cave " if cave else "") + nm[f], int(counts[f]), nvox, ns_per_eval, counts[f] * nvox * ns_per_eval * 1e-6))
rows.sort(key=lambda r: -r[4])
print(f"{'feature':22s} {'placements':>10s} {'box voxels':>10s} {'ps/eval':>9s} {'est ms':>8s}")
for r in rows:
    print(f"{r[0]:22s} {r[1]:10d} {r[2]:10d} {r[3]*1000:9.1f} {r[4]:8.3f}")
