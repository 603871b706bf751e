"""Randomised parity inside the driver-run suite: coordinates nobody chose.  The seed is derived from the SHA-256 of the library that is
loaded, so every build of libmmgen.so is checked at other places than the one before it, and it is printed, so a failure can be replayed
(`python tools/random_sweep.py <seed> ...` draws the same way from the same seed).  A slice of tools/random_sweep.py sized for ~20 s:
256 random chunks through the config-2 pipeline (heights, weights, layers, cave layers, blocks) at four distances from the origin incl. the
border of the pruning domain, and 12 random regions through the full pipeline (erosion + features + decorators), every fourth straddling
that border, every fifth with the 3-chunk ring built in full - all compared with the CPU oracle bit for bit."""
import hashlib

import numpy as np
import pytest
from random_draw import draw

pytestmark = pytest.mark.gpu


def _seed(mmgen_pkg):
    digest = hashlib.sha256(open(mmgen_pkg.LIB_PATH, "rb").read()).hexdigest()
    return int(digest[:8], 16), digest[:16]


def _same(got, ref):
    return np.array_equal(got.cpu().numpy().reshape(ref.shape).view(np.uint8), ref.view(np.uint8))


def test_random_chunks_config2_pipeline_match_oracle(gen, oracle, mmgen_pkg):
    seed, lib = _seed(mmgen_pkg)
    sets, _ = draw(seed, 256, 12)
    print(f"\nrandom parity: seed {seed} (lib_sha16 {lib}), 256 chunks   [replay: python tools/random_sweep.py {seed} 256 12]")
    for scale, coords in sets:
        out = gen.generate_chunks_no_erosion(gen.positions(coords))
        pos = oracle.positions(coords)
        hf, bw = oracle.heightfields(pos)
        layers = oracle.fix_backward(oracle.layers(pos, oracle.gather_heightfields(pos, hf), bw))
        cave = oracle.caves(pos, hf, bw)
        blocks = oracle.fill(pos, hf, bw, layers, cave)
        for name, ref in (("hf", hf), ("bw", bw), ("layers", layers), ("cave", cave), ("blocks", blocks)):
            got = out[name].cpu().numpy().reshape(ref.shape)
            bad = (got.view(np.uint8) != ref.view(np.uint8)).reshape(len(coords), -1).any(1)
            assert not bad.any(), f"seed {seed}: {name} differs at chunks {[coords[i] for i in np.flatnonzero(bad)[:4]]} (scale {scale})"


def test_random_regions_full_pipeline_match_oracle(gen, oracle, mmgen_pkg):
    seed, lib = _seed(mmgen_pkg)
    _, regions = draw(seed, 256, 12)
    print(f"\nrandom parity: seed {seed} (lib_sha16 {lib}), 12 regions   [replay: python tools/random_sweep.py {seed} 256 12]")
    for cx, cz, nx, nz, lazy in regions:
        ref = oracle.generate_region(cx, cz, nx, nz, erosion=True, features=True, decorators=True)
        got = gen.generate_region(cx, cz, nx, nz, want=("layers", "cave"), lazy_ring=lazy)
        for k in ("hf", "layers", "cave", "blocks"):
            assert _same(got[k], ref[k]), f"seed {seed}: {k} differs in region ({cx}, {cz}) {nx} x {nz}, {'lazy' if lazy else 'full'} ring"
        assert got["erosion_passes"] > 0
    assert len(set(regions)) == 12 and any(not r[4] for r in regions)
