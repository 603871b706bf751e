"""The coordinates of a randomised parity run, drawn from one seed: shared by tests/test_gpu_random.py (seed = hash of the loaded library)
and tools/random_sweep.py (seed from the command line), so that a seed printed by one replays in the other."""
import numpy as np


def draw(seed, nchunks, nregions):
    """-> (chunk sets [(scale, [(cx, cz), ...])], regions [(cx, cz, nx, nz, lazy_ring)])"""
    rng = np.random.default_rng(seed)
    sets = []
    for scale in (300, 5000, 60000, 2_000_000):
        coords = [tuple(int(v) for v in rng.integers(-scale, scale, 2)) for _ in range(nchunks // 4)]
        if scale == 5000:                                     # a quarter of these on the border of the pruning domain (chunk +-2048)
            for i in range(0, len(coords), 4):
                side = int(rng.integers(0, 4))
                edge = int(rng.choice([-2049, -2048, 2047, 2048]))
                other = int(rng.integers(-2200, 2200))
                coords[i] = (edge, other) if side < 2 else (other, edge)
        sets.append((scale, coords))
    regions = []
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
        regions.append((cx, cz, nx, nz, bool(i % 5)))           # every fifth region with the ring built in full (mask 1)
    return sets, regions
