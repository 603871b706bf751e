"""GPU parity tests for the feature stages (F1 placements, F2 gather, L2 rasterisers, D1 decorators) and the full region
pipeline (all stages, device resident) against the CPU oracle.  Bit-exact block ids is the bar."""
import os
import sys

import numpy as np
import pytest
from conftest import assert_bit_equal
from oracle_binding import feature_box

pytestmark = pytest.mark.gpu

SURFACE_FEATURES = list(range(1, 21))     # SPHERE .. CACTUS (biome.hpp:117-158)
CAVE_FEATURES = list(range(1, 10))        # TEST_GLOWSTONE_PILLAR .. AMBER_FUNGUS (biome.hpp:162-177)


def np_(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("feature", SURFACE_FEATURES)
def test_surface_feature_rasteriser(gen, oracle, feature):
    """Every surface feature rasterised alone in an empty volume at fixed placements (SURVEY §8c fixture 11): the box of
    claimed voxels and their block ids equal the oracle's for several placement positions (different per-feature RNG streams)."""
    total = 0
    for fpos in [(10, 100, -7), (-1234, 90, 777), (40001, 140, -39999), (5, 70, 5)]:
        box_min = (fpos[0] - 28, fpos[1] - 8, fpos[2] - 28)
        size = (57, 130, 57)
        ref = feature_box(oracle, False, feature, fpos, 0, box_min, size)
        got = gen.debug_feature_box(False, feature, fpos, 0, box_min, size)
        assert_bit_equal(got, ref, f"feature {feature} at {fpos}")
        total += int((ref != 255).sum())
    assert total > 0, f"feature {feature} never placed a block in any probe: probe volume is wrong"


@pytest.mark.parametrize("feature", CAVE_FEATURES)
def test_cave_feature_rasteriser(gen, oracle, feature):
    total = 0
    for fpos, lh in [((3, 40, 9), 12), ((-500, 20, 321), 25), ((7777, 60, -8888), 17)]:
        box_min = (fpos[0] - 16, fpos[1] - 16, fpos[2] - 16)
        size = (33, lh + 40, 33)
        ref = feature_box(oracle, True, feature, fpos, lh, box_min, size)
        got = gen.debug_feature_box(True, feature, fpos, lh, box_min, size)
        assert_bit_equal(got, ref, f"cave feature {feature} at {fpos}")
        total += int((ref != 255).sum())
    assert total > 0


REGIONS = [
    # (cx0, cz0, nx, nz): birch forest, jungle, redwood, crystals, coral reef, origin (tianzi pines), swamp, desert, icebergs, mushrooms
    (3654, -2794, 1, 1), (1488, -1110, 2, 1), (3518, 2777, 1, 1), (2669, -2199, 1, 2), (-1268, -1773, 1, 1), (0, 0, 2, 2),
    (1767, -1044, 1, 1), (3227, 152, 1, 1), (1602, 977, 1, 1), (3946, -3906, 1, 1), (-2105, -2470, 1, 1), (-88, -3971, 1, 1),
    # world edge: block coordinates 6.4e8 (fp32 ulp 64), noise lattice cells beyond the LDS tables' domain
    (40_000_000, -40_000_000, 1, 1),
    # across the border of the pruning domain (|block x|, |z| < 32 768 = chunk 2048, csrc/mm_noise.cuh): pruned and plain kernels in one launch
    (2046, -2050, 4, 3), (-2049, 2047, 2, 2),
]


@pytest.mark.parametrize("region", REGIONS)
def test_full_pipeline_region_matches_oracle(gen, oracle, region):
    """All stages incl. erosion, feature placement + gather, feature rasterisation and decorators, device resident: block ids,
    heightfields, eroded layers and cave layers of the region are bit-exact vs the oracle's region pipeline."""
    cx0, cz0, nx, nz = region
    ref = oracle.generate_region(cx0, cz0, nx, nz, erosion=True, features=True, decorators=True)
    out = gen.generate_region(cx0, cz0, nx, nz, want=("layers", "cave"))
    assert_bit_equal(np_(out["hf"]), ref["hf"], "heightfield")
    assert_bit_equal(np_(out["layers"]), ref["layers"], "eroded layers")
    assert_bit_equal(np_(out["cave"]), ref["cave"], "cave layers")
    got = np_(out["blocks"])
    if not np.array_equal(got, ref["blocks"]):
        bad = np.argwhere(got != ref["blocks"])
        c, i = bad[0]
        raise AssertionError(f"{len(bad)} block ids differ; first: chunk {c} column {i // 384} y {i % 384}: got {got[c, i]} want {ref['blocks'][c, i]}")


def test_lush_voxels_are_evaluated_in_place_when_their_queue_is_full(gen, oracle):
    """Clay / moss voxels of lush caves normally go through a device-wide queue to k_fill_lush; a reservation that does not fit is
    evaluated by the wave that made it (and marks its in-range slots as holes).  With the queue capped at 96 entries most of a region's
    reservations take that path: same blocks."""
    MOSS = 125                                                   # MMB_MOSS: only lush caves make it
    for cx0, cz0 in ((-37, 21), (10, -5), (60, 60), (-80, 33), (5, 90), (-120, -70)):
        nx, nz = 4, 3
        ref = oracle.generate_region(cx0, cz0, nx, nz, erosion=True, features=True, decorators=True)
        if int((ref["blocks"] == MOSS).sum()) > 300:
            break
    else:
        pytest.skip("none of the candidate regions has lush caves")
    try:
        assert gen.lib.mmgen_debug_set_lush_queue_cap(96) == 0
        out = gen.generate_region(cx0, cz0, nx, nz)
        assert_bit_equal(np_(out["blocks"]), ref["blocks"], "blocks with a 96-entry lush queue")
    finally:
        gen.lib.mmgen_debug_set_lush_queue_cap(0)
    out = gen.generate_region(cx0, cz0, nx, nz)
    assert_bit_equal(np_(out["blocks"]), ref["blocks"], "blocks with the full queue")


@pytest.mark.parametrize("flags", [(False, False, False), (True, False, False), (False, True, False), (False, False, True), (True, True, False)])
def test_region_flag_combinations(gen, oracle, flags):
    erosion, features, decorators = flags
    ref = oracle.generate_region(1488, -1110, 1, 1, erosion=erosion, features=features, decorators=decorators)
    out = gen.generate_region(1488, -1110, 1, 1, erosion=erosion, features=features, decorators=decorators)
    assert_bit_equal(np_(out["blocks"]), ref["blocks"], f"blocks flags={flags}")


def test_stage_level_features_match_region(gen, oracle):
    """The per-stage C ABI (placements -> gather -> fill with lists -> decorators) on a 7x7 chunk grid gives the same centre chunk
    as the oracle's region pipeline without erosion (stage-level drop-in path == region fast path)."""
    import torch
    cx, cz = 1488, -1110
    coords = [(cx - 3 + x, cz - 3 + z) for z in range(7) for x in range(7)]
    pos = gen.positions(coords)
    hf, bw, g = gen.generate_heightfields(pos, gathered=True)
    layers = gen.fix_backward_layers(gen.generate_layers(g, bw, pos))
    cave = gen.generate_caves(hf, bw, pos)
    fp, cfp, counts = gen.generate_feature_placements(hf, bw, layers, cave, pos)
    target = torch.tensor([24], dtype=torch.int32, device=gen.device)
    gfp, gcfp, bounds = gen.gather_feature_placements(fp, cfp, counts, target, 7, 7)
    sel = slice(24, 25)
    blocks = gen.fill(hf[sel].contiguous(), bw[sel].contiguous(), layers[sel].contiguous(), cave[sel].contiguous(), pos[sel].contiguous(),
                      gfp, gcfp, bounds)
    gen.place_decorators(blocks, hf[sel].contiguous(), bw[sel].contiguous(), cave[sel].contiguous(), pos[sel].contiguous())
    ref = oracle.generate_region(cx, cz, 1, 1, erosion=False, features=True, decorators=True)
    assert int(counts.sum()) > 0
    assert_bit_equal(np_(blocks), ref["blocks"], "stage-level feature path")


def test_tiled_generation_with_halo_exchange_on_one_gpu(mmgen_pkg, oracle):
    """The multi-GPU tiling path (masks: ring cells owned by a peer are NOT computed locally; placements arrive by exchange) on one
    device: two regions play ranks 0 and 1 of a 2x1 layout, the exchange is done by hand with the product's exchange_plan, and the
    stitched result equals the oracle's region.  (The RCCL transport itself is covered by the driver's multi-GPU bench run and, for the
    protocol, by the gloo tests on CPU.)"""
    import importlib
    import torch
    d = importlib.import_module("mega-minecraft_amd.distributed")
    layout = d.TileLayout(1487, -1111, 2, 1, 2, 2)
    gens = [mmgen_pkg.MMGen(0), mmgen_pkg.MMGen(0)]
    bufs = []
    for r in range(2):
        cx0, cz0, nx, nz = layout.region(r)
        mask = layout.local_mask(r)
        assert 0 in mask
        gens[r].region_begin(cx0, cz0, nx, nz, 7, mask)
        bufs.append(gens[r].region_placement_buffers())
    for r in range(2):
        for peer, (recv_cells, _) in layout.exchange_plan(r).items():
            send_cells = layout.exchange_plan(peer)[r][1]
            ri = torch.tensor(recv_cells, dtype=torch.long, device="cuda"); si = torch.tensor(send_cells, dtype=torch.long, device="cuda")
            for k in ("fp", "cfp", "counts"):
                bufs[r][k][ri] = bufs[peer][k][si]
    tiles = [np_(gens[r].region_finish(2, 2)["blocks"]) for r in range(2)]
    ref = oracle.generate_region(1487, -1111, 4, 2, erosion=True, features=True, decorators=True)["blocks"]
    world = np.zeros_like(ref)
    for r in range(2):
        for z in range(2):
            for x in range(2):
                world[(2 * r + x) + 4 * z] = tiles[r][x + 2 * z]
    assert np.array_equal(world, ref)


def test_gathered_list_overflow_truncates_like_the_oracle(gen, oracle):
    """Maximum sizes: the gathered placement lists are capped at 2 048 surface / 4 096 cave entries (chunk.cu:1555-1601).  The
    two-phase region API lets the test overwrite the per-chunk placement lists of the 7 x 7 ring with synthetic dense ones (60
    icebergs - horizontal reach 40 blocks - and 100 glowstone clusters per chunk: 2 940 and 4 900 gathered entries), identically on
    the HIP path and on the oracle.  Entries beyond the caps come from ring-3 chunks whose icebergs DO reach the centre chunk, so the
    blocks only agree if both sides gather in the same order, truncate at the same entry and rasterise first-match-wins alike."""
    import torch
    from oracle_binding import OracleBackend
    cx0, cz0 = 200, -300
    ob = OracleBackend(oracle.nthreads)
    ob.region_begin(cx0, cz0, 1, 1, 7)
    gen.region_begin(cx0, cz0, 1, 1, 7)
    gb = gen.region_placement_buffers()
    obuf = ob.region_placement_buffers()
    assert (gb["w"], gb["h"], gb["x0"], gb["z0"]) == (7, 7, cx0 - 3, cz0 - 3) == (obuf["w"], obuf["h"], obuf["x0"], obuf["z0"])
    rng = np.random.default_rng(11)
    fp = np.zeros((49, 256, 5), np.int32); cfp = np.zeros((49, 1024, 6), np.int32); counts = np.zeros((49, 2), np.int32)
    for cell in range(49):
        ox, oz = 16 * (cx0 - 3 + cell % 7), 16 * (cz0 - 3 + cell // 7)
        counts[cell] = (60, 100)
        fp[cell, :60, 0] = 4                                           # MMF_ICEBERG
        fp[cell, :60, 1] = ox + rng.integers(0, 16, 60); fp[cell, :60, 2] = 120 + rng.integers(0, 30, 60); fp[cell, :60, 3] = oz + rng.integers(0, 16, 60)
        fp[cell, :60, 4] = rng.integers(0, 2, 60)                      # canReplaceBlocks
        cfp[cell, :100, 0] = 4                                         # MMCF_GLOWSTONE_CLUSTER
        cfp[cell, :100, 1] = ox + rng.integers(0, 16, 100); cfp[cell, :100, 2] = 20 + rng.integers(0, 80, 100); cfp[cell, :100, 3] = oz + rng.integers(0, 16, 100)
        cfp[cell, :100, 4] = 4 + rng.integers(0, 12, 100); cfp[cell, :100, 5] = 1
    for k, a in (("fp", fp), ("cfp", cfp), ("counts", counts)):
        obuf[k].copy_(torch.from_numpy(a))
        gb[k].copy_(torch.from_numpy(a).to(gb[k].device))
    ref = ob.region_finish(1, 1)["blocks"]
    got = np_(gen.region_finish(1, 1)["blocks"])
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} block ids differ"
    plain = oracle.generate_region(cx0, cz0, 1, 1, erosion=True, features=True, decorators=True)["blocks"]
    assert int((ref != plain).sum()) > 500                            # the synthetic icebergs really claim voxels of the centre chunk


def _tiled_world_on_one_gpu(mmgen_pkg, layout):
    """All ranks of `layout` played one after the other on this GPU: region_begin with the peer-owned ring masked out, placement
    lists exchanged by hand along the product's exchange_plan, region_finish; returns the stitched [nz_world * nx_world, 98304] blocks."""
    import torch
    n = layout.world_size
    gens = [mmgen_pkg.MMGen(0) for _ in range(n)]
    bufs = []
    for r in range(n):
        cx0, cz0, nx, nz = layout.region(r)
        gens[r].region_begin(cx0, cz0, nx, nz, 7, layout.local_mask(r))
        bufs.append(gens[r].region_placement_buffers())
    for r in range(n):
        for peer, (recv_cells, _) in layout.exchange_plan(r).items():
            send_cells = layout.exchange_plan(peer)[r][1]
            ri = torch.tensor(recv_cells, dtype=torch.long, device="cuda"); si = torch.tensor(send_cells, dtype=torch.long, device="cuda")
            for k in ("fp", "cfp", "counts"):
                bufs[r][k][ri] = bufs[peer][k][si]
    wx0, wz0 = layout.region(0)[0], layout.region(0)[1]
    regs = [layout.region(r) for r in range(n)]
    W = max(c[0] + c[2] for c in regs) - wx0
    H = max(c[1] + c[3] for c in regs) - wz0
    world = torch.zeros((H, W, 98304), dtype=torch.uint8, device="cuda")
    for r in range(n):
        cx0, cz0, nx, nz = regs[r]
        world[cz0 - wz0:cz0 - wz0 + nz, cx0 - wx0:cx0 - wx0 + nx] = gens[r].region_finish(nx, nz)["blocks"].view(nz, nx, 98304)
        gens[r] = None
    return world.view(H * W, 98304), (wx0, wz0, W, H)


def _world_digests(d):
    import os
    return d.load_world_digests(os.path.join(os.path.dirname(__file__), "golden", "world_digests.npz"))


def _assert_chunks_equal_oracle_digests(d, torch, blocks, cx0, cz0, nx, nz, what):
    """Every chunk of the rectangle against the ORACLE's digest of that chunk (tests/golden/world_digests.npz, tests/golden/make_world_digests.py:
    the CPU oracle over the whole [-128, 128)^2 world, no HIP code involved)."""
    gold = d.golden_tile_digests(_world_digests(d), cx0, cz0, nx, nz)
    assert gold is not None
    got = d.chunk_digests(blocks, torch).cpu().numpy()
    bad = np.nonzero(got != gold)[0]
    assert bad.size == 0, f"{what}: {bad.size} of {nx * nz} chunks differ from the oracle, first {[(cx0 + int(i) % nx, cz0 + int(i) // nx) for i in bad[:8]]}"


def test_config4_world_2x2_tiles_equals_single_region_and_oracle(mmgen_pkg, oracle):
    """BASELINE config 4 at full size: the 4 096-chunk world [-32, 32)^2, all stages, as 2 x 2 tiles of 32 x 32 chunks with the
    placement-ring exchange == the same world generated as ONE region (tiling invariance, every block of 4 096 chunks), EVERY chunk ==
    the CPU oracle's chunk by digest (the golden world), and the 2 x 2 chunks around the four-tile corner == the oracle run live."""
    import importlib
    import torch
    d = importlib.import_module("mega-minecraft_amd.distributed")
    layout = d.TileLayout(-32, -32, 2, 2, 32, 32)
    world, (wx0, wz0, W, H) = _tiled_world_on_one_gpu(mmgen_pkg, layout)
    single = mmgen_pkg.MMGen(0).generate_region(wx0, wz0, W, H)["blocks"]
    assert torch.equal(world, single)
    _assert_chunks_equal_oracle_digests(d, torch, world, wx0, wz0, W, H, "config 4 world")
    ref = oracle.generate_region(-1, -1, 2, 2, erosion=True, features=True, decorators=True)["blocks"]
    got = world.view(H, W, 98304)[31:33, 31:33].reshape(4, 98304).cpu().numpy()
    assert np.array_equal(got, ref)


def test_region_with_more_zones_than_one_relaxation_batch_matches_oracle(gen, oracle):
    """A 120 x 120-chunk region touches 12 x 12 = 144 erosion zones: two batches of the persistent relaxation (96 + 48, MMGEN_EROSION_ZONE_BATCH)
    beside one cave launch.  Chunks whose zone is relaxed by the SECOND batch (zone index 116 of the z-major order) == the CPU oracle, and the
    pass count the region reports covers both batches."""
    out = gen.generate_region(-60, -60, 120, 120)
    ref = oracle.generate_region(30, 40, 2, 2, erosion=True, features=True, decorators=True)["blocks"]
    got = out["blocks"].view(120, 120, 98304)[100:102, 90:92].reshape(4, 98304).cpu().numpy()
    assert np.array_equal(got, ref)
    first = oracle.generate_region(-60, -60, 1, 1, erosion=True, features=True, decorators=True)["blocks"]      # ... and one of the first batch
    assert np.array_equal(out["blocks"][0].cpu().numpy(), first[0])
    assert gen.lib.mmgen_region_last_erosion_passes(gen._region()) >= 24


def test_zone_cache_gives_the_same_regions(mmgen_pkg, gen, oracle):
    """mmgen_region_set_zone_cache (the streaming scheduler's: eroded zones kept across calls, K1 / K2 only on the gathered areas of the
    zones that are new): a walk of thin strips - the regime it exists for - with a cache too small for the walk (evictions), in the DAG and
    in the serial schedule, gives block for block, layer for layer what a region without the cache gives; one strip == the CPU oracle."""
    import torch
    walker = mmgen_pkg.MMGen(0)
    walker.region_set_zone_cache(5)
    strips = [(1480 + k, -1120, 1, 35) for k in range(8)] + [(1470, -1100 + 3 * k, 35, 2) for k in range(4)] + [(1480, -1120, 1, 35), (-3, -3, 7, 7), (1487, -1120, 1, 35)]
    for i, (cx0, cz0, nx, nz) in enumerate(strips):
        walker.region_set_serial(i % 5 == 4)
        got = walker.generate_region(cx0, cz0, nx, nz, want=("layers", "cave"))
        ref = gen.generate_region(cx0, cz0, nx, nz, want=("layers", "cave"))
        for k in ("blocks", "hf", "layers", "cave"):
            assert torch.equal(got[k], ref[k]), (i, k, (cx0, cz0, nx, nz))
    hits, misses = walker.region_zone_cache_stats()
    assert hits > 20 and misses >= 5, (hits, misses)
    cx0, cz0, nx, nz = strips[3]
    want = oracle.generate_region(cx0, cz0, nx, 3, erosion=True, features=True, decorators=True)["blocks"]
    got = walker.generate_region(cx0, cz0, nx, 3)
    assert np.array_equal(got["blocks"].cpu().numpy(), want)
    walker.region_set_zone_cache(0)                            # off again: everything is relaxed anew, same result
    again = walker.generate_region(cx0, cz0, nx, 3)
    assert torch.equal(again["blocks"], got["blocks"]) and walker.region_zone_cache_stats() == (0, 0)


def test_config5_world_8_tiles_equals_single_region(mmgen_pkg):
    """BASELINE config 5 at full size: the 65 536-chunk world [-128, 128)^2 as 4 x 2 tiles of 64 x 128 chunks (the 8-GPU layout,
    played on one GPU) is block-for-block the world generated as one region (6.4 GB of block ids compared on the device), and every chunk
    of it equals the CPU oracle's chunk by digest."""
    import importlib
    import torch
    d = importlib.import_module("mega-minecraft_amd.distributed")
    layout = d.TileLayout(-128, -128, 4, 2, 64, 128)
    world, (wx0, wz0, W, H) = _tiled_world_on_one_gpu(mmgen_pkg, layout)
    assert (W, H) == (256, 256)
    g = mmgen_pkg.MMGen(0)
    single = g.generate_region(wx0, wz0, W, H)["blocks"]
    assert torch.equal(world, single)
    # ... and EVERY one of the 65 536 chunks is the CPU oracle's chunk (digest for digest: the bit-exact block-id diff of the north star at
    # full size), hence every tile has the checksum bench.py derives from the same digests for the first 8-GPU run (tiles_bit_exact)
    _assert_chunks_equal_oracle_digests(d, torch, world, wx0, wz0, W, H, "config 5 world")
    w3 = world.view(H, W, 98304)
    gold_world = _world_digests(d)
    for r in range(8):
        cx0, cz0, nx, nz = layout.region(r)
        tile = w3[cz0 - wz0:cz0 - wz0 + nz, cx0 - wx0:cx0 - wx0 + nx].reshape(nx * nz, 98304)
        assert d.tile_checksum(tile, torch) == d.checksum_of_digests(torch.from_numpy(d.golden_tile_digests(gold_world, cx0, cz0, nx, nz)), torch), r
    # the product-only capacity (MMGEN_CFP_CAP = 1 024 cave placements per chunk) over the whole 65 536-chunk world: far away
    longest = g.region_max_cave_placements()
    assert 0 < longest < 512, longest
    # not a trivially empty world: bedrock floor everywhere, and a healthy mix of block ids
    assert bool((single.view(-1, 384)[:, 0] == 56).all())
    assert int(torch.unique(single[::97]).numel()) > 40


@pytest.mark.parametrize("name", ["world_digests_jungle", "world_digests_border", "world_digests_edge"])
def test_three_more_worlds_equal_the_oracle_chunk_for_chunk(mmgen_pkg, name):
    """Three more 65 536-chunk worlds, each generated as one region and held to the CPU oracle's per-chunk digests (tests/golden/make_world_digests.py on
    the GPU box's host threads, profiles/r06e_world_digests_*): [1400, 1656) x [-1240, -984) - jungle / swamp / mesa country around the chunks
    most tests use - and [1920, 2176) x [-128, 128), which straddles the pruning domain's border at block 32 768 (chunk 2 048): the rows
    beyond it take k_fill_far and the unpruned cave / rasteriser paths, the rows inside the pruned ones, in one launch; and
    [39 999 872, 40 000 128) x [-128, 128): block coordinates of 6.4e8, where float32 steps by 64 - every lattice hash outside its table
    domain, 308 870 columns of the oracle's run in the canonical no-layer case (r06n_world_digests_edge_*)."""
    import importlib
    import os
    import torch
    d = importlib.import_module("mega-minecraft_amd.distributed")
    cx0, cz0, gold = d.load_world_digests(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    nz, nx = gold.shape
    assert (nx, nz) == (256, 256)
    got = d.chunk_digests(mmgen_pkg.MMGen(0).generate_region(cx0, cz0, nx, nz)["blocks"], torch).cpu().numpy().reshape(nz, nx)
    bad = np.argwhere(got != gold)
    assert bad.size == 0, f"{name}: {len(bad)} of 65536 chunks differ from the oracle, first {[(cx0 + int(x), cz0 + int(z)) for z, x in bad[:8]]}"


def test_cpp_chunk_api_matches_region_path(mmgen_pkg):
    """The C++ mirror of the reference's Chunk stage API (host/chunk.hpp: generateHeightfields, gatherHeightfield, generateLayers,
    erodeZone, generateCaves, generateFeaturePlacements, gatherFeaturePlacements, fill) driven like Terrain::tick for one zone gives
    the same blocks as the device-resident region path (the headless driver compares in-process; exit code 0 = identical)."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(mmgen_pkg.LIB_PATH), "mmgen_headless")
    assert os.path.exists(exe), "build it with make -C mega-minecraft_amd/csrc"
    for zone in (("0", "0"), ("1488", "-1116")):
        r = subprocess.run([exe, *zone], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "0 chunks differ" in r.stdout


def test_cpp_terrain_scheduler_streams_a_world(mmgen_pkg):
    """The C++ mirror of the reference's Terrain scheduler (host/terrain.hpp: action-time budget, nine queues, zones, spiral) ticked
    headlessly until all queues drain: all (2*16+1)^2 chunks around the player become DRAWABLE and sampled chunks equal the region
    path bit for bit (canonical raw-padding erosion makes the result independent of the order zones get eroded in)."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(mmgen_pkg.LIB_PATH), "mmgen_terrain_demo")
    assert os.path.exists(exe), "build it with make -C mega-minecraft_amd/csrc"
    r = subprocess.run([exe, "-40", "25"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "1089 drawable" in r.stdout and ", 0 bad" in r.stdout


def test_region_batched_scheduler_equals_action_time_scheduler(mmgen_pkg):
    """RegionTerrain (host/region_terrain.hpp: one device-resident region call per missing rectangle + pool meshing, budget in chunks)
    and Terrain (the drop-in mirror of the reference's action-time loop) stream the world around a player and then around a second
    position: all 1 089 drawable chunks have identical blocks, vertices and indices in both (the demo compares in-process); device resident,
    two lanes == one lane == the mirror, and a walk of one-chunk steps (the strip path: meshes enqueued by their tick, booked by the next)
    ends at the same chunks as a one-tick load."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(mmgen_pkg.LIB_PATH), "mmgen_region_terrain_demo")
    assert os.path.exists(exe), "build it with make -C mega-minecraft_amd/csrc"
    r = subprocess.run([exe, "1487", "-1111"], capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "IDENTICAL" in r.stdout and "1089 drawable chunks compared" in r.stdout
    # the streaming form (device resident, 35-chunk strips whose meshes are booked by the following tick) ends at the one-tick load's chunks
    assert "vs the one-tick load: 0 of 1089 drawable chunks differ" in r.stdout


def _two_process_worker(rank, world, port, layout_args, outdir):
    import importlib
    import os
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("mega-minecraft_amd")
    d = importlib.import_module("mega-minecraft_amd.distributed")
    from host_staged_dist import HostStagedDist
    gen = pkg.MMGen(0)                                   # both ranks share GPU 0
    layout = d.TileLayout(*layout_args)
    ctx = d.TileContext(layout, rank, torch, gen.device)
    shim = HostStagedDist(dist, torch)
    for _ in range(2):                                   # twice: the cached layout / plan path of the second step too
        out = d.generate_tile(gen, layout, rank, 7, dist=shim, torch=torch, ctx=ctx)
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, f"blocks_{rank}.npy"), out["blocks"].cpu().numpy())
    np.save(os.path.join(outdir, f"halo_{rank}.npy"), np.array([out["halo_bytes_received"]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_tiling_on_one_gpu_runs_the_real_exchange(gen, tmp_path):
    """The product's N > 1 path end to end on the device with a real process group: two PROCESSES (gloo group, device tensors staged
    through the host by tests/host_staged_dist.py because RCCL refuses two ranks on one GPU) generate the two 6x4-chunk tiles of a
    12x4 world with distributed.generate_tile - TileContext, mmgen_ring_pack_messages / _unpack_messages on the device, the one-phase
    exchange, the base fill under it, lazy world-border ring cells - and the stitched tiles equal the single region."""
    import socket
    import torch.multiprocessing as mp
    layout_args = (1484, -1112, 2, 1, 6, 4)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_process_worker, args=(2, port, layout_args, str(tmp_path)), nprocs=2, join=True)
    tiles = [np.load(tmp_path / f"blocks_{r}.npy") for r in range(2)]
    halo = [int(np.load(tmp_path / f"halo_{r}.npy")[0]) for r in range(2)]
    single = np_(gen.generate_region(1484, -1112, 12, 4)["blocks"])
    world = np.zeros_like(single)
    for r, t in enumerate(tiles):
        for z in range(4):
            for x in range(6):
                world[(6 * r + x) + 12 * z] = t[x + 6 * z]
    assert np.array_equal(world, single), f"{int((world != single).sum())} block ids differ between the two-process tiling and the single region"
    # 3 x 4 ring cells arrive per rank in one fixed-size message: 2 length words + the default budget of 2 048 payload words per cell
    # (a dense cell would be 29.7 KB)
    assert halo == [12 * (2 + 2048) * 4] * 2, halo


def test_cpp_tiled_world_host_single_tile(mmgen_pkg):
    """The C++ multi-GPU host (host/tiled_world.hpp: TileLayout + ExchangePlan + RCCL grouped send / recv, linked against librccl) on one
    GPU: the tiled path with a single tile gives the same blocks as mmgen_region_generate (the demo compares checksums in-process).
    The N > 1 exchange needs N GPUs; its plan is held to distributed.py's by tests/test_distributed_cpu.py and its device kernels by
    test_ring_wire_format_on_the_device."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(mmgen_pkg.LIB_PATH), "mmgen_tiled_demo")
    assert os.path.exists(exe), "build it with make -C mega-minecraft_amd/csrc"
    r = subprocess.run([exe, "--gpus", "1", "--tile", "12", "12", "--steps", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "single tile == mmgen_region_generate: ok" in r.stdout


# ------------------------------------------------------------------------------------------------ F1 / F2 observed directly
def _fields(fp, cfp):
    """placement records as plain fields (padding bytes of the 20 / 24-byte structs masked out)"""
    f = np.stack([fp[..., 0] & 255, fp[..., 1], fp[..., 2], fp[..., 3], fp[..., 4] & 255], -1)
    c = np.stack([cfp[..., 0] & 255, cfp[..., 1], cfp[..., 2], cfp[..., 3], cfp[..., 4], cfp[..., 5] & 255], -1)
    return f, c


F1_REGIONS = [(1488, -1110, 2, 2), (2669, -2199, 2, 2), (3654, -2794, 2, 1), (-1268, -1773, 1, 2), (3946, -3906, 2, 2), (0, 0, 2, 2),
              (3518, 2777, 1, 1), (1767, -1044, 2, 1), (-2105, -2470, 1, 1), (-88, -3971, 1, 1)]


def test_placement_lists_match_oracle_cell_for_cell(gen, oracle):
    """F1 (chunk.cu:1041-1156): the per-chunk placement lists the device emits — counts, entries and EMISSION ORDER (columns z-major,
    emission order inside a column) — equal the oracle's on every cell of the ring-extended grid of ten regions (jungle, crystals, birch,
    coral reef, mushrooms, origin, redwood, swamp, ...), with erosion on (the surface test reads eroded layer thickness)."""
    from oracle_binding import OracleBackend
    ob = OracleBackend(nthreads=oracle.nthreads)
    seen_surface, seen_cave, total = set(), set(), [0, 0]
    for cx0, cz0, nx, nz in F1_REGIONS:
        ob.region_begin(cx0, cz0, nx, nz, 7)
        want = ob.region_placement_buffers()
        gen.region_begin(cx0, cz0, nx, nz, 7)
        got = gen.region_placement_buffers()
        assert (got["x0"], got["z0"], got["w"], got["h"]) == (want["x0"], want["z0"], want["w"], want["h"])
        gc, wc = np_(got["counts"]), want["counts"].numpy()
        assert np.array_equal(gc, wc), f"placement counts differ in region {(cx0, cz0)}: cells {np.argwhere(gc != wc)[:5].tolist()}"
        gf, gcf = _fields(np_(got["fp"]), np_(got["cfp"]))
        wf, wcf = _fields(want["fp"].numpy(), want["cfp"].numpy())
        for cell in range(gc.shape[0]):
            n0, n1 = int(wc[cell, 0]), min(int(wc[cell, 1]), 1024)
            assert np.array_equal(gf[cell, :n0], wf[cell, :n0]), f"surface placements of cell {cell} in region {(cx0, cz0)}"
            assert np.array_equal(gcf[cell, :n1], wcf[cell, :n1]), f"cave placements of cell {cell} in region {(cx0, cz0)}"
            seen_surface.update(wf[cell, :n0, 0].tolist()); seen_cave.update(wcf[cell, :n1, 0].tolist())
        total[0] += int(wc[:, 0].sum()); total[1] += int(wc[:, 1].sum())
        gen.region_finish(nx, nz)
    assert total[0] > 500 and total[1] > 500, total
    # the sample really exercises the gens: jungle trees, crystals, cave vines, glowstone, a crystal-caves feature
    assert {12, 13, 14} <= seen_surface and {17, 18} & seen_surface and {3, 4} <= seen_cave and {5, 6, 7} & seen_cave, (seen_surface, seen_cave)


def test_gathered_lists_match_reference_order(gen, golden):
    """F2 (chunk.cu:1158-1196 + :1555-1601) through the per-stage ABI: for every chunk of a jungle region the gathered lists are the
    concatenation of the 49 neighbour lists in the reference's offset order (tests/golden/ref_tables.npz: gather_offsets, extracted from
    chunk.cu), NONE-terminated, and the bounds are the union of pos.y + featureHeightBounds over the un-truncated lists."""
    import torch
    ref = golden["ref_tables"]
    nx, nz = 3, 2
    gen.region_begin(1487, -1111, nx, nz, 7)
    b = gen.region_placement_buffers()
    w, h = b["w"], b["h"]
    targets = torch.tensor([(x + 3) + w * (z + 3) for z in range(nz) for x in range(nx)], dtype=torch.int32, device=gen.device)
    gfp, gcfp, bounds = gen.gather_feature_placements(b["fp"], b["cfp"], b["counts"], targets, w, h)
    fp, cfp, cnt = np_(b["fp"]), np_(b["cfp"]), np_(b["counts"])
    gfp, gcfp, bounds = np_(gfp), np_(gcfp), np_(bounds)
    gen.region_finish(nx, nz)
    for k, cell in enumerate(targets.tolist()):
        cx, cz = cell % w, cell // w
        ls, lc = [], []
        for dx, dz in ref["gather_offsets"].tolist():
            n = (cx + dx) + w * (cz + dz)
            ls.append(fp[n, :cnt[n, 0]]); lc.append(cfp[n, :min(cnt[n, 1], 1024)])
        ls, lc = np.concatenate(ls), np.concatenate(lc)
        assert len(ls) < 2048 and len(lc) < 4096 and len(ls) > 20 and len(lc) > 20
        wf, wcf = _fields(ls, lc)
        gf, gcf = _fields(gfp[k], gcfp[k])
        assert np.array_equal(gf[:len(ls)], wf) and gf[len(ls), 0] == 0, f"gathered surface list of chunk {k}"
        assert np.array_equal(gcf[:len(lc)], wcf) and gcf[len(lc), 0] == 0, f"gathered cave list of chunk {k}"
        fb, cfb = ref["feature_bounds"], ref["cave_feature_bounds"]
        want = [int((wf[:, 2] + fb[wf[:, 0], 0]).min()), int((wf[:, 2] + fb[wf[:, 0], 1]).max()),
                int((wcf[:, 2] + cfb[wcf[:, 0], 0]).min()), int((wcf[:, 2] + wcf[:, 4] + cfb[wcf[:, 0], 1]).max())]
        assert bounds[k].tolist() == want, (bounds[k].tolist(), want)


def test_config4_world_placement_counts_stay_below_the_caps(gen):
    """MMGEN_FP_CAP (256) and MMGEN_CFP_CAP (1024) are product-only limits (the reference pushes to unbounded vectors, chunk.cu:1028-1038):
    over the whole config-4 world (4 096 chunks + ring) no chunk comes near them, so no placement is ever dropped."""
    gen.region_max_gathered()                                # (clears the record of earlier tests)
    gen.region_begin(-32, -32, 64, 64, 7)
    c = np_(gen.region_placement_buffers()["counts"])
    gen.region_finish(64, 64)
    assert c.shape[0] == 70 * 70
    assert c[:, 0].max() <= 256 and c[:, 1].max() <= 1024, (c[:, 0].max(), c[:, 1].max())
    assert c[:, 0].max() < 128 and c[:, 1].max() < 512, "within a factor 2 of a cap: raise MMGEN_FP_CAP / MMGEN_CFP_CAP"
    assert c[:, 0].sum() > 10000 and c[:, 1].sum() > 10000
    # ... and the reference's own limits, the truncation of the GATHERED lists (2 048 / 4 096, chunk.cu:1573-1601), are never reached either:
    # with the ring computed in full (mask 1: the lengths Chunk::fill would see) the longest list of the whole world stays below a third of
    # them.  So the reference truncates nothing in a generated world, and the lazily built ring (whose shortened lists could only be
    # truncated differently if the full ones were truncated at all) gives the same blocks - checked on the same world below.
    gs, gc = gen.region_max_gathered()
    assert 100 < gs < 2048 // 3 and 100 < gc < 4096 // 3, (gs, gc)


def test_lazy_ring_equals_full_ring(gen):
    """mmgen_region_generate builds the 3-chunk ring lazily (mask 2: only placements that can reach the rectangle, cave noise only on the
    columns that can produce one).  Blocks, heights and the gathered lists' effect are those of the full ring (mask 1) - on feature-rich
    regions, a region at the pruning-domain border and one far from the origin."""
    import torch
    gen.region_max_gathered()
    for (cx0, cz0, nx, nz) in [(1484, -1112, 6, 5), (-20, 7, 9, 4), (2040, -2050, 8, 8), (60000, -41000, 5, 5)]:
        lazy = gen.generate_region(cx0, cz0, nx, nz, lazy_ring=True)
        ls, lc = gen.region_max_gathered()
        full = gen.generate_region(cx0, cz0, nx, nz, lazy_ring=False)
        fs, fc = gen.region_max_gathered()
        assert torch.equal(lazy["blocks"], full["blocks"]) and torch.equal(lazy["hf"], full["hf"]), (cx0, cz0)
        assert ls <= fs < 2048 and lc <= fc < 4096, (ls, fs, lc, fc)          # the lazy lists are sub-lists; nothing is truncated


def test_shortened_ring_lists_give_the_same_blocks_until_the_reference_truncates(gen, oracle):
    """The lazy ring's contract with dense synthetic lists (the caller provides the ring: mask 0).  Every cell of the 7 x 7 grid around one
    chunk carries icebergs (horizontal reach 40 blocks) and glowstone clusters (reach 8); the SHORTENED variant drops, from the 48 ring
    cells, the clusters that cannot reach the chunk - what a lazy ring leaves out.  While the full gathered lists stay below the reference's
    truncation (2 048 / 4 096 entries, chunk.cu:1573-1601) both variants produce the oracle's blocks; once the full list is longer than the
    limit the reference drops its tail and the shortened list - which no longer reaches the limit - may keep entries the reference loses.
    mmgen_region_max_gathered shows a full-ring caller which case it is in."""
    import torch
    from oracle_binding import OracleBackend
    cx0, cz0 = 200, -300
    ob = OracleBackend(oracle.nthreads)

    def lists(nS, nC, seed):
        rng = np.random.default_rng(seed)
        fp = np.zeros((49, 256, 5), np.int32); cfp = np.zeros((49, 1024, 6), np.int32); counts = np.zeros((49, 2), np.int32)
        for cell in range(49):
            ox, oz = 16 * (cx0 - 3 + cell % 7), 16 * (cz0 - 3 + cell // 7)
            counts[cell] = (nS, nC)
            fp[cell, :nS, 0] = 4                                       # MMF_ICEBERG
            fp[cell, :nS, 1] = ox + rng.integers(0, 16, nS); fp[cell, :nS, 2] = 120 + rng.integers(0, 30, nS); fp[cell, :nS, 3] = oz + rng.integers(0, 16, nS)
            fp[cell, :nS, 4] = rng.integers(0, 2, nS)
            cfp[cell, :nC, 0] = 4                                      # MMCF_GLOWSTONE_CLUSTER
            cfp[cell, :nC, 1] = ox + rng.integers(0, 16, nC); cfp[cell, :nC, 2] = 20 + rng.integers(0, 80, nC); cfp[cell, :nC, 3] = oz + rng.integers(0, 16, nC)
            cfp[cell, :nC, 4] = 4 + rng.integers(0, 12, nC); cfp[cell, :nC, 5] = 1
        return fp, cfp, counts

    def shortened(fp, cfp, counts):
        fp2, cfp2, c2 = fp.copy(), np.zeros_like(cfp), counts.copy()
        x0, z0 = 16 * cx0, 16 * cz0
        for cell in range(49):
            e = cfp[cell, :counts[cell, 1]]
            keep = (e[:, 1] >= x0 - 8) & (e[:, 1] <= x0 + 15 + 8) & (e[:, 3] >= z0 - 8) & (e[:, 3] <= z0 + 15 + 8) if cell != 24 else np.ones(len(e), bool)
            cfp2[cell, :keep.sum()] = e[keep]
            c2[cell, 1] = keep.sum()
        return fp2, cfp2, c2

    def run(backend, fp, cfp, counts):
        backend.region_begin(cx0, cz0, 1, 1, 7, [0] * 24 + [1] + [0] * 24 if backend is gen else None)
        b = backend.region_placement_buffers()
        for k, a in (("fp", fp), ("cfp", cfp), ("counts", counts)):
            b[k].copy_(torch.from_numpy(a).to(b[k].device))
        out = backend.region_finish(1, 1)["blocks"]
        return out if isinstance(out, np.ndarray) else np_(out)

    # below the limits: 30 + 60 per cell -> 1 470 / 2 940 gathered entries
    gen.region_max_gathered()
    full = lists(30, 60, 3)
    short = shortened(*full)
    assert short[2][:, 1].sum() < full[2][:, 1].sum() // 2                 # the shortened lists really are much shorter
    ref = run(ob, *full)
    assert np.array_equal(run(gen, *full), ref)
    assert gen.region_max_gathered() == (1470, 2940)
    assert np.array_equal(run(gen, *short), ref), "a shortened ring changed blocks although the reference truncates nothing"
    # beyond them: 60 + 100 per cell -> 2 940 / 4 900: the device still equals the oracle on the FULL lists (same order, same cut) ...
    full = lists(60, 100, 11)
    ref = run(ob, *full)
    assert np.array_equal(run(gen, *full), ref)
    gs, gc = gen.region_max_gathered()
    assert (gs, gc) == (2940, 4900) and gs > 2048 and gc > 4096            # ... and a full-ring caller can see that the reference's cut applied
    # (the shortened variant of THIS case is the documented limit of the lazy ring: its cave list is below 4 096 again, so the clusters of
    # the last ring cells, which the reference drops, stay in)
    short = shortened(*full)
    assert run(gen, *short).shape == ref.shape and gen.region_max_gathered()[1] < 4096


def test_ring_wire_format_on_the_device(gen):
    """mmgen_ring_header / offsets / pack / unpack == the CPU statement of the wire format (tests/oracle_binding.py), incl. a cell whose
    cave count exceeds the cap, and a pack -> unpack round trip into a second placement grid."""
    import torch
    from oracle_binding import OracleBackend
    cpu = OracleBackend.__new__(OracleBackend); cpu.torch = torch
    gen.region_begin(1487, -1111, 2, 2, 7)
    b = gen.region_placement_buffers()
    bufs_cpu = dict(fp=b["fp"].cpu().clone(), cfp=b["cfp"].cpu().clone(), counts=b["counts"].cpu().clone())
    cells = torch.tensor([27, 3, 40, 41, 12, 63, 0], dtype=torch.int32)
    bufs_cpu["counts"][12, 1] = 1500                         # over the cap: the raw count travels, 1 024 entries do
    b["counts"][12, 1] = 1500
    dcells = cells.to(gen.device)
    hdr = gen.ring_header(b, dcells); off = gen.ring_offsets(hdr)
    chdr = cpu.ring_header(bufs_cpu, cells); coff = cpu.ring_offsets(chdr)
    assert torch.equal(hdr.cpu(), chdr) and torch.equal(off.cpu(), coff)
    total = int(off[-1])
    payload = gen.ring_pack(b, dcells, hdr, off, total)
    assert torch.equal(payload.cpu()[:total], cpu.ring_pack(bufs_cpu, cells, chdr, coff, total)[:total])
    dst = dict(fp=torch.zeros_like(b["fp"]), cfp=torch.zeros_like(b["cfp"]), counts=torch.zeros_like(b["counts"]))
    gen.ring_unpack(dst, dcells, hdr, off, payload)
    for c in cells.tolist():
        n0, n1 = min(int(bufs_cpu["counts"][c, 0]), 256), min(int(bufs_cpu["counts"][c, 1]), 1024)
        assert torch.equal(dst["fp"][c, :n0].cpu(), bufs_cpu["fp"][c, :n0]) and torch.equal(dst["cfp"][c, :n1].cpu(), bufs_cpu["cfp"][c, :n1])
        assert torch.equal(dst["counts"][c].cpu(), bufs_cpu["counts"][c])
    assert int(dst["counts"].cpu().sum()) == int(bufs_cpu["counts"][cells.long()].sum())
    # the one-phase messages (mmgen_ring_pack_messages / _unpack_messages): two peers' cells in one call, word for word the CPU statement,
    # with room to spare and with a budget the second peer's message does not fit
    import importlib as il
    d = il.import_module("mega-minecraft_amd.distributed")
    for wpc in (3000, 1200):
        bounds, slots = d.message_layout([0, 3, 7], wpc)
        slots_cpu = torch.tensor(slots, dtype=torch.int32).reshape(-1, 4)
        slots_dev = slots_cpu.to(gen.device)
        msg_cpu, of_cpu = torch.zeros(bounds[-1], dtype=torch.int32), torch.zeros(1, dtype=torch.int32)
        cpu.ring_pack_messages(bufs_cpu, cells, slots_cpu, None, msg_cpu, of_cpu)
        msg = torch.zeros(bounds[-1], dtype=torch.int32, device=gen.device)
        scratch = torch.zeros(3 * len(cells) + 1, dtype=torch.int32, device=gen.device)
        of = torch.zeros(1, dtype=torch.int32, device=gen.device)
        gen.ring_pack_messages(b, dcells, slots_dev, scratch, msg, of)
        assert torch.equal(msg.cpu(), msg_cpu) and int(of) == int(of_cpu) and (int(of) > 0) == (wpc == 1200)
        dst = dict(fp=torch.zeros_like(b["fp"]), cfp=torch.zeros_like(b["cfp"]), counts=torch.full_like(b["counts"], -1))
        dst_cpu = dict(fp=torch.zeros_like(bufs_cpu["fp"]), cfp=torch.zeros_like(bufs_cpu["cfp"]), counts=torch.full_like(bufs_cpu["counts"], -1))
        of.zero_(); of_cpu.zero_()
        gen.ring_unpack_messages(dst, dcells, slots_dev, scratch, msg, of)
        cpu.ring_unpack_messages(dst_cpu, cells, slots_cpu, None, msg_cpu, of_cpu)
        assert int(of) == int(of_cpu)
        for k in ("fp", "cfp", "counts"):
            assert torch.equal(dst[k].cpu(), dst_cpu[k]), (k, wpc)
    gen.region_finish(2, 2)
    assert gen.region_max_cave_placements() == 1500          # the over-the-cap cell was noticed by finish; acknowledged here


@pytest.mark.parametrize("is_cave", [False, True])
def test_rasterisers_stay_inside_reach(gen, golden, is_cave):
    """The column filters drop a placement whose Chebyshev distance exceeds the library's reach table (a product-only bound; the height
    bounds are the reference's own and are applied by the caller, chunk.cu:1446-1452): hundreds of placements per feature (per-feature
    RNG streams: iceberg radius, branch angles, crystal directions, mushroom splines) rasterised on the device into boxes that extend
    16 blocks beyond the reach and 6 beyond the height bounds never claim a voxel farther away than the reach."""
    t = gen.debug_tables()
    rs = np.random.RandomState(7 + is_cave)
    reach = t["cave_feature_reach" if is_cave else "feature_reach"]
    bounds = golden["ref_tables"]["cave_feature_bounds" if is_cave else "feature_bounds"]
    for feature in (CAVE_FEATURES if is_cave else SURFACE_FEATURES):
        r = int(reach[feature])
        side = 2 * (r + 16) + 1
        n_place = 12 if r > 60 else (60 if r > 20 else 200)
        lo, hi = int(bounds[feature, 0]), int(bounds[feature, 1])
        claimed = 0
        for _ in range(n_place):
            fpos = (int(rs.randint(-5000, 5000)), int(rs.randint(20, 200) if not is_cave else rs.randint(12, 100)), int(rs.randint(-5000, 5000)))
            if not is_cave and feature in (2, 3, 4):                      # coral / kelp / iceberg only generate under water
                fpos = (fpos[0], int(rs.randint(60, 96)), fpos[2])
            lh = int(rs.randint(4, 40)) if is_cave else 0
            y0 = max(fpos[1] + lo - 6, 0)
            y1 = min(fpos[1] + lh + hi + 6, 383)
            if not is_cave and feature == 4:
                y1 = 383                                                  # icebergs are placed relative to sea level
            box_min = (fpos[0] - r - 16, y0, fpos[2] - r - 16)
            size = (side, y1 - y0 + 1, side)
            got = gen.debug_feature_box(is_cave, feature, fpos, lh, box_min, size).reshape(side, side, size[1])   # z, x, y
            zz, xx, yy = np.nonzero(got != 255)
            claimed += len(zz)
            if len(zz) == 0:
                continue
            dx, dz = xx + box_min[0] - fpos[0], zz + box_min[2] - fpos[2]
            assert max(np.abs(dx).max(), np.abs(dz).max()) <= r, f"feature {feature} at {fpos}: claims at distance {max(np.abs(dx).max(), np.abs(dz).max())} > reach {r}"
        assert claimed > 0 or feature in (1, 2), f"feature {feature} never claimed a voxel"


# (surface features, placements per chunk), (cave features, placements per chunk): densities keep a column's candidates below the
# capacities of k_apply_features' buffers (128 placements, 256 pairs per flush): the dense cases below run through many flushes per unit
TIGHT_CASES = [
    (([2] * 20 + [16] + [15] * 2, 70), ([8, 9], 200)),                  # CORAL, PURPLE / MEDIUM_PURPLE_MUSHROOM; WARPED / AMBER_FUNGUS
    (([6], 14), ([4, 7], 150)),                                        # REDWOOD_TREE; GLOWSTONE_CLUSTER, CRYSTAL_PILLAR
    (([8, 9, 10], 90), ([5, 6], 90)),                                  # BIRCH / PINE_TREE / PINE_SHRUB; STORMLIGHT spheres (floor, ceiling)
    (([17, 18], 9), ([3], 250)),                                       # MEDIUM_CRYSTAL, CRYSTAL; CAVE_VINE
    (([16], 4), ([3], 30)),                                            # PURPLE_MUSHROOM alone (stem hull, cap slab); they reach up to 72 blocks
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(TIGHT_CASES)))
def test_tight_extents_lose_nothing(gen, oracle, case):
    """k_apply_features drops (placement, column) pairs and shortens vertical extents with per-placement bounds derived from the
    rasterisers' own early-outs (csrc/mmgen_features.hip: surface_extent / cave_extent, 16 features).  The oracle walks every entry for
    every voxel like the reference does.  Dense synthetic lists of exactly those features (every shape and size the draws produce, over
    the 7 x 7 ring, random heights, layer heights from 0 and canReplaceBlocks) go through both; a bound that is too tight shows up as
    a block the HIP path did not place, one that ignores the reference's own height test as a block it placed in excess."""
    import torch
    from oracle_binding import OracleBackend
    (surf, ns), (cave, nc) = TIGHT_CASES[case]
    total_diff_from_plain = 0
    for (cx0, cz0, seed) in ((200, -300, 5 + 10 * case), (-40, 12, 6 + 10 * case)):
        ob = OracleBackend(oracle.nthreads)
        ob.region_begin(cx0, cz0, 1, 1, 7)
        gen.region_begin(cx0, cz0, 1, 1, 7)
        gb = gen.region_placement_buffers()
        obuf = ob.region_placement_buffers()
        rng = np.random.default_rng(seed)
        fp = np.zeros((49, 256, 5), np.int32); cfp = np.zeros((49, 1024, 6), np.int32); counts = np.zeros((49, 2), np.int32)
        for cell in range(49):
            ox, oz = 16 * (cx0 - 3 + cell % 7), 16 * (cz0 - 3 + cell // 7)
            counts[cell] = (ns, nc)
            fp[cell, :ns, 0] = rng.choice(surf, ns)
            fp[cell, :ns, 1] = ox + rng.integers(0, 16, ns); fp[cell, :ns, 3] = oz + rng.integers(0, 16, ns)
            fp[cell, :ns, 2] = np.where(fp[cell, :ns, 0] == 2, rng.integers(60, 126, ns), rng.integers(90, 190, ns))
            fp[cell, :ns, 4] = rng.integers(0, 2, ns)                          # canReplaceBlocks
            cfp[cell, :nc, 0] = rng.choice(cave, nc)
            cfp[cell, :nc, 1] = ox + rng.integers(0, 16, nc); cfp[cell, :nc, 2] = 5 + rng.integers(0, 120, nc); cfp[cell, :nc, 3] = oz + rng.integers(0, 16, nc)
            cfp[cell, :nc, 4] = rng.integers(0, 30, nc); cfp[cell, :nc, 5] = rng.integers(0, 2, nc)
        for k, a in (("fp", fp), ("cfp", cfp), ("counts", counts)):
            obuf[k].copy_(torch.from_numpy(a))
            gb[k].copy_(torch.from_numpy(a).to(gb[k].device))
        ref = ob.region_finish(1, 1)["blocks"]
        got = np_(gen.region_finish(1, 1)["blocks"])
        assert np.array_equal(got, ref), f"case {case}, region ({cx0},{cz0}): {int((got != ref).sum())} block ids differ"
        plain = oracle.generate_region(cx0, cz0, 1, 1, erosion=True, features=True, decorators=True)["blocks"]
        total_diff_from_plain += int((ref != plain).sum())
    assert total_diff_from_plain > 1000                                       # the synthetic placements really claim voxels


@pytest.mark.gpu
def test_sphere_whose_radius_is_the_rounded_root_of_its_rim_distance(gen, oracle):
    """Found by the first diff of the full 65 536-chunk world against the oracle's digests (round 6; random sweeps of 4 x 65 536 other chunks
    had not met the case): the ceiling stormlight sphere at block (-1478, 53, -1267), layer height 11, draws radius == fl(sqrt(34)), so its
    eight rim voxels (+-3, 0, +-5), (+-5, 0, +-3) pass the rasteriser's `dist > radius` test (featurePlacement.hpp:1202-1208) exactly at
    equality - and cave_extent had dropped their columns because fl(radius^2) < 34.  Chunks (-93, -80) and (-93, -79), all stages."""
    ref = oracle.generate_region(-93, -80, 1, 2, erosion=True, features=True, decorators=True)["blocks"]
    got = np_(gen.generate_region(-93, -80, 1, 2)["blocks"])
    assert int((ref == 93).sum()) > 100                                        # the sphere's cyan crystal is there ...
    assert ref[0, 384 * (7 + 16 * 8) + 64] == 93                               # ... including the rim voxel at (-1481, 64, -1272)
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} block ids differ"


@pytest.mark.gpu
def test_bench_n2_rehearsal_on_one_gpu():
    """bench.py's N > 1 path (parent spawns torch.distributed.run before touching HIP, two ranks, 2 x 1 tile layout, TileContext, two-phase
    ring exchange overlapped with the base fill, max-over-ranks timing, the border spot check against the oracle) rehearsed with both
    ranks on ONE GPU: gloo group + host-staged p2p (tests/host_staged_dist.py) instead of RCCL, which refuses two ranks on one device.
    Small tiles; the line must carry n_gpus = 2, be marked dry_run, and its blocks at the tile border must be bit-exact."""
    import json, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MMGEN_BENCH_ONE_GPU_DRYRUN="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--tile-nx", "24",
                        "--tile-nz", "24", "--cpu-side", "12"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and "dry_run" in line and line["config"]["tiles"] == "2x1"
    assert line["halo_bytes_received_per_step_all_ranks"] > 0
    assert line["parity_spot_check"] == "bit-exact"
    assert "roofline" in line and "cpu_baseline" in line and line["scaling"] == "weak"


# ------------------------------------------------------------------------------------------------ stage DAG == serial schedule
@pytest.mark.gpu
@pytest.mark.parametrize("slices", [0, 2, 5])
def test_stage_dag_equals_serial_schedule(gen, slices):
    """The region's stage DAG (erosion beside the caves, z slices pipelined over four streams, include/mmgen.h mmgen_region_set_serial) and
    the one-stream schedule give identical blocks / heights / layers / cave layers, step after step (back-to-back steps of different
    rectangles exercise the cross-step ordering of the internal streams), with and without the optional middle fill call."""
    import torch
    regs = [(-20, 7, 24, 40), (100, -300, 16, 24), (-20, 7, 24, 40)]
    want = ("layers", "cave")
    try:
        gen.region_set_serial(True)
        ref = [gen.generate_region(*r, want=want) for r in regs]
        torch.cuda.synchronize()
        gen.region_set_serial(False, slices)
        got = [gen.generate_region(*r, want=want) for r in regs]          # enqueued back to back, no synchronisation in between
        torch.cuda.synchronize()
        for r, a, b in zip(regs, ref, got):
            for k in ("blocks", "hf", "layers", "cave"):
                assert torch.equal(a[k], b[k]), f"{k} of region {r} differs between the serial schedule and the DAG with {slices} slices"
        # begin -> fill -> finish (the tiling caller's order), and a begin -> fill whose finish never comes followed by a new begin
        cx0, cz0, nx, nz = regs[0]
        mask = [2] * ((nx + 6) * (nz + 6))
        gen.region_begin(cx0, cz0, nx, nz, 7, mask)
        gen.region_fill(nx, nz)
        gen.region_begin(cx0, cz0, nx, nz, 7, mask)
        gen.region_fill(nx, nz)
        out = gen.region_finish(nx, nz)
        torch.cuda.synchronize()
        assert torch.equal(out["blocks"], ref[0]["blocks"])
    finally:
        gen.region_set_serial(False, 0)


# ------------------------------------------------------------------------------------------------ RCCL on the one GPU there is
def _rccl_loopback_worker(port, outdir):
    import importlib as il
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    pkg = il.import_module("mega-minecraft_amd")
    d = il.import_module("mega-minecraft_amd.distributed")
    gen = pkg.MMGen(0)
    lay = d.TileLayout(1480, -1120, 1, 1, 12, 10)
    ctx = d.TileContext(lay, 0, torch, gen.device, loopback=True)
    outs = [d.generate_tile(gen, lay, 0, 7, dist=dist, torch=torch, ctx=ctx) for _ in range(2)]          # twice: buffers re-used
    ref = gen.generate_region(1480, -1120, 12, 10)
    torch.cuda.synchronize()
    res = dict(equal=[bool(torch.equal(o["blocks"], ref["blocks"])) for o in outs], halo=[int(o["halo_bytes_received"]) for o in outs],
               backend=dist.get_backend())
    import json
    json.dump(res, open(os.path.join(outdir, "loopback.json"), "w"))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_loopback_ships_the_ring_through_the_real_transport(tmp_path):
    """RCCL itself, on the one GPU a box has: a communicator of ONE rank (backend nccl = RCCL), the product's generate_tile /
    exchange_placements with TileContext(loopback=True): the whole 3-chunk ring is packed on the device, sent rank 0 -> rank 0 by
    torch.distributed.batch_isend_irecv (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on device buffers), its local copy wiped
    while the payload is in flight, unpacked; the tile must equal the plain region.  Runs in a child process (its own process group)."""
    import json
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_rccl_loopback_worker, args=(port, str(tmp_path)))
    p.start(); p.join(600)
    assert p.exitcode == 0, f"RCCL loopback worker exit code {p.exitcode}"
    res = json.load(open(tmp_path / "loopback.json"))
    assert res["backend"] == "nccl" and res["equal"] == [True, True] and all(h > 10000 for h in res["halo"]), res


@pytest.mark.gpu
def test_cpp_tiled_world_rccl_loopback(mmgen_pkg):
    """The C++ host over RCCL (host/tiled_world.cpp: ncclGroupStart / ncclSend / ncclRecv on sMain and sComm, the event hand-over between
    them, the base fill while the payload travels) with a one-rank communicator in loopback mode: tile == mmgen_region_generate."""
    import subprocess
    exe = os.path.join(os.path.dirname(mmgen_pkg.LIB_PATH), "mmgen_tiled_demo")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, "--loopback", "--tile", "20", "12", "--steps", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ring shipped rank 0 -> rank 0 over RCCL" in r.stdout and "ok" in r.stdout, r.stdout


@pytest.mark.gpu
def test_cave_placement_cap_overflow_is_loud(mmgen_pkg):
    """The reference's cave placement lists are unbounded (chunk.cu:1028-1038); the library keeps MMGEN_CFP_CAP = 1 024 per chunk.  A list
    beyond the cap (forced here by raising a cell's count between begin and finish, the way a received ring header could) is recorded by
    finish, every later begin / finish fails with MMGEN_ERROR_PLACEMENT_OVERFLOW until mmgen_region_max_cave_placements acknowledges it,
    and a normal world stays far below the cap."""
    import torch
    gen = mmgen_pkg.MMGen(0)                     # its own region handle: the sticky error must not leak into other tests
    cx0, cz0, nx, nz = 1488, -1110, 3, 3
    gen.region_begin(cx0, cz0, nx, nz, 7)
    gen.region_finish(nx, nz)
    normal = gen.region_max_cave_placements()
    assert 0 < normal < 512, normal
    gen.region_begin(cx0, cz0, nx, nz, 7)
    bufs = gen.region_placement_buffers()
    bufs["counts"][4, 1] = 1500                  # a ring cell claims 1 500 cave placements
    gen.region_finish(nx, nz)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="MMGEN_CFP_CAP"):
        gen.region_begin(cx0, cz0, nx, nz, 7)
    assert gen.region_max_cave_placements() == 1500
    out = gen.generate_region(cx0, cz0, nx, nz)  # acknowledged: works again
    # (generate_region builds the ring lazily: its lists are subsets of the full ones)
    assert 0 < gen.region_max_cave_placements() <= normal and out["blocks"].shape[0] == nx * nz


@pytest.mark.gpu
def test_release_frees_the_stream_scratch(gen):
    """mmgen_release(stream): the per-(device, stream) scratch of the per-stage calls does not outlive its stream"""
    import ctypes
    import torch
    s = torch.cuda.Stream()
    pos = gen.positions([(3, 4), (5, 6)])
    with torch.cuda.stream(s):
        hf, bw, _ = gen.generate_heightfields(pos, gathered=True)
        a = gen.generate_caves(hf, bw, pos)
        gen.lib.mmgen_release.argtypes = [ctypes.c_void_p]
        assert gen.lib.mmgen_release(ctypes.c_void_p(s.cuda_stream)) == 0
        assert gen.lib.mmgen_release(ctypes.c_void_p(s.cuda_stream)) == 0        # nothing left: a no-op
        b = gen.generate_caves(hf, bw, pos)                                        # scratch comes back on demand
    s.synchronize()
    assert torch.equal(a, b)
