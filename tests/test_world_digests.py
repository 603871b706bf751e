"""tests/golden/world_digests.npz - the CPU oracle's digest of every chunk of the 65 536-chunk world [-128, 128)^2 (BASELINE config 5; it
contains config 4's world and every tile of bench.py's layouts), written by tests/golden/make_world_digests.py on the GPU box's 256 host threads
with no HIP code involved (log: profiles/r06_world_digests_oracle_gen.log).  Here: the file is what it says - the oracle, run again on this
machine on sampled chunks as 1 x 1 regions (a different piece size than the 64 x 64 pieces it was made from), reproduces the digests."""
import importlib
import os
import sys

import numpy as np

from conftest import ROOT, GOLDEN

sys.path.insert(0, GOLDEN)


def test_the_other_golden_worlds_are_the_oracles_too(oracle):
    """world_digests_jungle.npz ([1400, 1656) x [-1240, -984)) and world_digests_border.npz ([1920, 2176) x [-128, 128), across the pruning
    domain's border) and world_digests_edge.npz ([39 999 872, 40 000 128) x [-128, 128): block coordinates of 6.4e8): one sampled chunk of each, regenerated
    here as a 1 x 1 region."""
    d = importlib.import_module("mega-minecraft_amd.distributed")
    gen = importlib.import_module("make_world_digests")
    for name, (cx, cz) in (("world_digests_jungle", (1488, -1110)), ("world_digests_border", (2048, 17)), ("world_digests_edge", (40000000, 5))):
        cx0, cz0, dig = d.load_world_digests(os.path.join(GOLDEN, name + ".npz"))
        blocks = oracle.generate_region(cx, cz, 1, 1, erosion=True, features=True, decorators=True, lean=True)["blocks"]
        assert int(gen.chunk_digests_np(blocks)[0]) == int(dig[cz - cz0, cx - cx0]), (name, cx, cz)


def test_golden_world_digests_are_the_oracles(oracle):
    d = importlib.import_module("mega-minecraft_amd.distributed")
    gen = importlib.import_module("make_world_digests")
    cx0, cz0, dig = d.load_world_digests(os.path.join(GOLDEN, "world_digests.npz"))
    rng = np.random.default_rng(20261003)
    # two random chunks, a world corner, and the chunk the first full-world diff of the device against these digests found wrong (round 6)
    samples = [tuple(int(v) for v in rng.integers(-128, 128, 2)) for _ in range(2)] + [(-128, 127), (-93, -80)]
    for cx, cz in samples:
        blocks = oracle.generate_region(cx, cz, 1, 1, erosion=True, features=True, decorators=True, lean=True)["blocks"]
        assert int(gen.chunk_digests_np(blocks)[0]) == int(dig[cz - cz0, cx - cx0]), (cx, cz)
