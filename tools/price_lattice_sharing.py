#!/usr/bin/env python3
"""Pricing of the "multi-voxel lane" for k_cave_voxels' position warp, without building it: a lane that owns two (or four)
y-adjacent voxels can compute simplex3's lattice part - mod289, the permute chain, the four gradient fetches: ~35 of an octave's 142 VALU
instructions and its 9 LDS reads - once for both ONLY when they lie in the same skewed cell with the same corner ordering, and a wave only
skips those instructions when that holds for EVERY one of its 64 lanes (the instructions issue for the wave if a single lane needs them).
This script measures that probability on the bench tile's geometry, in float32 with the kernel's own expressions:
    fbm3from3<5>(noisePos * 0.8): stack j samples (p * 0.005 * 0.8 + offset_j) * 2^octave, offsets (0, 0, 0), (5923.45, 4129.42, 5790.48), (1765.68, 4704.36, 5692.12)
for waves of 64 lanes = 16 columns x 4 y-groups, a lane owning `per` consecutive y.  Prints, per octave, the fraction of waves in which all lanes
share, and the VALU instructions the variant could save per voxel of the warp (upper bound: no cost charged for the checks or the registers)."""
import numpy as np
f32 = np.float32
rng = np.random.default_rng(7)
OFF = np.array([[0, 0, 0], [5923.45, 4129.42, 5790.48], [1765.68, 4704.36, 5692.12]], f32)


def cell_and_order(v):
    d = (v[..., 0] * f32(1 / 3) + v[..., 1] * f32(1 / 3)) + v[..., 2] * f32(1 / 3)
    i = np.floor(v + d[..., None])
    e = (i[..., 0] * f32(1 / 6) + i[..., 1] * f32(1 / 6)) + i[..., 2] * f32(1 / 6)
    x0 = (v - i) + e[..., None]
    order = (x0[..., 0] < x0[..., 1]).astype(np.int8) + 2 * (x0[..., 1] < x0[..., 2]) + 4 * (x0[..., 2] < x0[..., 0])
    return i, order


for per in (2, 4):
    share = np.zeros(5)
    waves = 0
    for _ in range(400):
        cx, cz, row = rng.integers(-32, 32), rng.integers(-64, 64), rng.integers(0, 16)
        y0 = rng.integers(1, 120 - 4 * per)                       # a wave: 16 columns x 4 lanes x `per` consecutive y, inside the noise band
        x = (16 * cx + np.arange(16)).astype(f32)[:, None, None]
        y = (y0 + per * np.arange(4)[None, :, None] + np.arange(per)[None, None, :]).astype(f32)
        z = f32(16 * cz + row)
        p = np.stack(np.broadcast_arrays(x * f32(0.005), y * f32(0.005), np.full_like(y, z) * f32(0.005)), -1).astype(f32) * f32(0.8)
        for k in range(5):
            ok = True
            for j in range(3):
                v = (p + OFF[j]) * f32(2 ** k)
                i, o = cell_and_order(v.astype(f32))
                same = np.all(i == i[:, :, :1], axis=(2, 3)) & np.all(o == o[:, :, :1], axis=2)
                ok_j = bool(same.all())
                share[k] += ok_j / 3
        waves += 1
    frac = share / waves
    saved = (per - 1) / per * 35 * frac                          # of every `per` evaluations, per - 1 skip the lattice part when the wave shares
    print(f"{per} voxels per lane: waves in which every lane shares cell + ordering, octave 0..4: {np.round(frac, 3).tolist()}")
    print(f"   VALU instructions saved per octave evaluation (of 142): {np.round(saved, 1).tolist()};  per voxel's warp (15 evaluations, ~2 130 VALU): "
          f"{3 * saved.sum():.0f} = {100 * 3 * saved.sum() / 2130:.1f} % of the warp, ~{100 * 3 * saved.sum() / 2130 * 0.62:.1f} % of k_cave_voxels, "
          f"~{100 * 3 * saved.sum() / 2130 * 0.62 * 10.6 / 21.4:.1f} % of the step - before the costs (2 x / 4 x the per-voxel registers at 80 VGPRs x 6 waves, the per-octave check)")
