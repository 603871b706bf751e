"""what the SCHEDULERS produce, held to the CPU oracle directly.

Three hosts stream the world around a player over the same C ABI and write one line per drawable chunk (host/chunk_digest.hpp):
  * oracle/_ref/ref_terrain_dropin - the REFERENCE's own, unmodified src/terrain/terrain.cpp (its Terrain::tick, queues, zones, action-time
    budget) compiled where it lies against mmgen's Chunk in place of the reference's chunk.hpp / chunk.cu (tests/refdrop/);
  * mmgen_terrain_demo - host/terrain.cpp, the mirror of that scheduler;
  * mmgen_region_terrain_demo - host/region_terrain.cpp, the region-batched streaming scheduler.
Every drawable chunk's block digest must equal the ORACLE's digest of that chunk (tests/golden/world_digests.npz, made by
tests/golden/make_world_digests.py with no HIP code involved), and the vertex / index bytes of 36 chunks around the player must equal the oracle's
Chunk::createVBOs restatement (oracle/mmo_mesh.cpp) run on the oracle's own blocks."""
import importlib
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, GOLDEN

pytestmark = pytest.mark.gpu

K_WORD = np.uint64(0x9E3779B97F4A7C15)


def word_digest(buf):
    w = np.frombuffer(np.ascontiguousarray(buf).tobytes(), dtype=np.uint64)
    with np.errstate(over="ignore"):
        return int((w * ((2 * np.arange(w.size, dtype=np.uint64) + 1) * K_WORD)).sum(dtype=np.uint64))


def read_digests(path):
    out = {}
    for line in open(path):
        cx, cz, b, nv, v, i = line.split()
        out[(int(cx), int(cz))] = (int(b, 16), int(nv), int(v, 16), int(i, 16))
    return out


@pytest.fixture(scope="module")
def world():
    d = importlib.import_module("mega-minecraft_amd.distributed")
    return d, d.load_world_digests(os.path.join(GOLDEN, "world_digests.npz"))


@pytest.fixture(scope="module")
def oracle_meshes(oracle):
    """Blocks + meshes of the 6 x 6 chunks around (0, 0) from the oracle alone: an 8 x 8 region (one chunk of neighbours on every side),
    Chunk::createVBOs' restatement on each inner chunk with its four neighbours."""
    cx0, cz0, n = -4, -4, 8
    blocks = oracle.generate_region(cx0, cz0, n, n, erosion=True, features=True, decorators=True, lean=True)["blocks"].reshape(n, n, 98304)
    out = {}
    for z in range(1, n - 1):
        for x in range(1, n - 1):
            nb = [blocks[z + 1, x], blocks[z, x + 1], blocks[z - 1, x], blocks[z, x - 1]]          # N (+z), E (+x), S (-z), W (-x)
            verts, idx = oracle.create_vbos(blocks[z, x], nb, (cx0 + x) * 16, (cz0 + z) * 16)
            out[(cx0 + x, cz0 + z)] = (word_digest(blocks[z, x]), len(verts), word_digest(verts), word_digest(idx))
    return out


def check_against_oracle(lines, world, oracle_meshes, player, what):
    d, w = world
    assert len(lines) >= 33 * 33, (what, len(lines))
    bad = []
    for dz in range(-16, 17):
        for dx in range(-16, 17):
            c = (player[0] + dx, player[1] + dz)
            assert c in lines, f"{what}: chunk {c} is not drawable"
            gold = d.golden_tile_digests(w, c[0], c[1], 1, 1)
            assert gold is not None
            if (int(gold[0]) & 0xFFFFFFFFFFFFFFFF) != lines[c][0]:
                bad.append(c)
    assert not bad, f"{what}: {len(bad)} of 1089 drawable chunks differ from the oracle's blocks, first {bad[:5]}"
    meshed = [c for c in oracle_meshes if c in lines]
    assert len(meshed) >= 32
    for c in meshed:
        assert lines[c] == oracle_meshes[c], f"{what}: chunk {c}: (blocks, vertex count, vertex bytes, index bytes) {lines[c]} vs the oracle's {oracle_meshes[c]}"


def test_reference_terrain_tick_drives_mmgen_chunks(mmgen_pkg, world, oracle_meshes, tmp_path):
    """The reference's unmodified Terrain (terrain.cpp:84-960) over mmgen's Chunk: all 1 089 chunks it makes drawable carry the oracle's
    blocks, 36 of them the oracle's mesh bytes.  The binary is built by tests/refdrop/Makefile where /root/reference exists and travels
    under oracle/_ref/; without it (a checkout that never saw the reference) the test is skipped."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_terrain_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_terrain_dropin not built (needs /root/reference: make -C tests/refdrop)")
    out = tmp_path / "ref.txt"
    r = subprocess.run([exe, str(out), "0", "0"], capture_output=True, text=True, timeout=1500)
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    check_against_oracle(read_digests(out), world, oracle_meshes, (0, 0), "reference Terrain over mmgen Chunk")


def test_action_time_mirror_chunks_equal_oracle(mmgen_pkg, world, oracle_meshes, tmp_path):
    exe = os.path.join(os.path.dirname(mmgen_pkg.LIB_PATH), "mmgen_terrain_demo")
    out = tmp_path / "mirror.txt"
    r = subprocess.run([exe, "0", "0", str(out)], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    check_against_oracle(read_digests(out), world, oracle_meshes, (0, 0), "host/terrain.cpp")


def test_region_terrain_chunks_equal_oracle(mmgen_pkg, world, oracle_meshes, tmp_path):
    """RegionTerrain after its two legs (player (-13, 5), then (0, 0)): everything drawable around the second position - with one lane and
    with several (streaming over several GPUs; rehearsed with two handles on one GPU where there is only one)."""
    exe = os.path.join(os.path.dirname(mmgen_pkg.LIB_PATH), "mmgen_region_terrain_demo")
    out = tmp_path / "region.txt"
    r = subprocess.run([exe, "-13", "5", str(out)], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "IDENTICAL" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    check_against_oracle(read_digests(out), world, oracle_meshes, (0, 0), "host/region_terrain.cpp")
    # ... and of the same scheduler with several lanes (one per GPU, or two region handles on the one GPU of this box): the strips the lanes
    # generated and the meshes across their borders carry the oracle's blocks and mesh bytes too
    assert " lanes (" in r.stdout and "chunks per lane:" in r.stdout
    check_against_oracle(read_digests(str(out) + ".lanes"), world, oracle_meshes, (0, 0), "host/region_terrain.cpp with lanes")
