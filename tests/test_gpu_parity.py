"""GPU parity tests (pytest -m gpu, run on the MI355X box): the HIP path through the C ABI against the CPU oracle on
the same inputs, against the committed golden vectors, and — at BASELINE's full batch size — through size-independent
properties.  Bar: bit-exact for block ids, cave layers and (stronger than the 1e-5 north-star tolerance) heightfields."""
import hashlib

import numpy as np
import pytest
from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu
HEIGHT_TOL = 1e-5      # north_star: heightfield floats within 1e-5 of the reference (we additionally assert bit equality)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def np_(t):
    return t.cpu().numpy()


# ------------------------------------------------------------------------------------------------ device math
def test_device_simplex_matches_real_glm(gen, golden):
    g = golden["glm_probe"]
    assert_bit_equal(gen.debug_probe("simplex2", g["xy"], len(g["xy"]))[:, 0], g["simplex2"], "device simplex2 vs glm")
    assert_bit_equal(gen.debug_probe("simplex3", g["xyz"], len(g["xyz"]))[:, 0], g["simplex3"], "device simplex3 vs glm")
    # the lattice-split form k_cave_voxels uses (gradients shared per wave through LDS) is the same function
    assert_bit_equal(gen.debug_probe("simplex3_split", g["xyz"], len(g["xyz"]))[:, 0], g["simplex3"], "device split simplex3 vs glm")


def test_device_simplex_lattice_hash_shortcuts(gen, oracle):
    """The device replaces glm::mod(i, 289)'s IEEE division by floor((i + 0.5) * fl(1 / 289)) and one fused multiply-add (simplex2, |i| < 2^21)
    and mod289's multiply + subtract by one fused multiply-add (simplex3, |i| < 2^23): exact by the arguments in csrc/mm_noise.cuh (the
    first one is also checked for every integer of its domain on the CPU, tests/test_oracle_math.py).  Held to the oracle (itself held to
    the real glm on 1.6 M points) where those arguments are tight: lattice coordinates around multiples of 289, negative, up to and across
    the domain limits."""
    import ctypes
    f = np.float32
    rs = np.random.RandomState(9)

    def pts(dim, scale, n):
        base = (rs.randint(-scale // 289, scale // 289, (n, dim)) * 289 + rs.randint(-2, 3, (n, dim))).astype(np.float64)
        return np.concatenate([(base + rs.rand(n, dim)).astype(f), ((rs.rand(n, dim) - 0.5) * 2 * scale).astype(f)])

    def ref(fn, a):
        out = np.zeros(len(a), f)
        a = np.ascontiguousarray(a)
        fn(len(a), a.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
        return out

    # small; up to and across 2^21; large; across 2^23; across 2^24 (arguments are ~ 1.4 x the lattice coordinate)
    for scale in (5000, 1_450_000, 3_000_000, 4_000_000, 8_388_000 * 2, 16_777_000 * 2):
        a2 = pts(2, scale, 20000)
        assert_bit_equal(gen.debug_probe("simplex2", a2, len(a2))[:, 0], ref(oracle.lib.mmo_simplex2, a2), f"simplex2 at scale {scale}")
        a3 = pts(3, scale, 20000)
        assert_bit_equal(gen.debug_probe("simplex3", a3, len(a3))[:, 0], ref(oracle.lib.mmo_simplex3, a3), f"simplex3 at scale {scale}")
        assert_bit_equal(gen.debug_probe("simplex3_split", a3, len(a3))[:, 0], ref(oracle.lib.mmo_simplex3, a3), f"split simplex3 at scale {scale}")


def test_device_math_matches_kat(gen, golden):
    k = golden["oracle_kat"]
    n = len(k["trig_in"])
    assert_bit_equal(gen.debug_probe("sin", k["trig_in"], n)[:, 0], k["sin"], "sin")
    assert_bit_equal(gen.debug_probe("cos", k["trig_in"], n)[:, 0], k["cos"], "cos")
    assert_bit_equal(gen.debug_probe("pow", np.stack([k["pow_x"], k["pow_y"]], 1), 1024)[:, 0], k["pow"], "pow")
    assert_bit_equal(gen.debug_probe("atan2", np.stack([k["atan2_y"], k["atan2_x"]], 1), 1024)[:, 0], k["atan2"], "atan2")
    assert_bit_equal(gen.debug_probe("acos", k["acos_x"], 1024)[:, 0], k["acos"], "acos")
    assert_bit_equal(gen.debug_probe("hash", k["hash_in"], 1024)[:, 0].view(np.uint32), k["hash_out"], "hash")
    seeds = k["rng_seeds"].copy()
    assert_bit_equal(gen.debug_probe("rng4_u01", seeds, 512, 4), k["u01_w1"], "u01 (4-arg seed)")
    s3 = seeds.copy(); s3[:, 3] = np.int32(-2**31)
    assert_bit_equal(gen.debug_probe("rng4_u01", s3, 512, 4), k["u01_w0"], "u01 (3-arg seed)")
    assert_bit_equal(gen.debug_probe("fbm2_5", k["fbm2_in"], 1024)[:, 0], k["fbm2_5"], "fbm2<5>")
    assert_bit_equal(gen.debug_probe("fbm3_4", k["fbm3_in"], 1024)[:, 0], k["fbm3_4"], "fbm3<4>")
    assert_bit_equal(gen.debug_probe("rand3from3", k["cells"], 1024, 3), k["rand3from3"], "rand3From3")
    assert_bit_equal(gen.debug_probe("worley2", k["fbm2_in"], 1024, 5), k["worley2"], "worley2")
    assert_bit_equal(gen.debug_probe("worley3", k["fbm3_in"], 1024, 5), k["worley3"], "worley3")
    assert_bit_equal(gen.debug_probe("special_cave_noise", k["fbm3_in"], 1024)[:, 0], k["special_cave_noise"], "specialCaveNoise")
    for b in range(24):
        packed = np.zeros((64, 3), np.float32); packed[:, 0] = np.int32(b).view(np.float32); packed[:, 1:] = k["biome_height_pos"]
        assert_bit_equal(gen.debug_probe("biome_height", packed, 64)[:, 0], k["biome_height"][b], f"getHeight biome {b}")
    packed = np.zeros((2048, 5), np.float32)
    packed[:, :3] = k["cb_vox"].view(np.float32); packed[:, 3] = k["cb_maxheight"]; packed[:, 4] = np.int32(190249401).view(np.float32)
    assert (gen.debug_probe("cave_biome", packed, 2048)[:, 0].astype(np.uint8) == k["cave_biome"]).all()


# ------------------------------------------------------------------------------------------------ stage parity
def _coords(golden):
    return [tuple(c) for c in golden["stages"]["coords"].tolist()]


def test_stages_match_oracle_and_golden(gen, oracle, golden):
    """Every stage output of the HIP path on 32 chunks (all 24 biomes + mixed + negative coordinates) == oracle, == golden."""
    s = golden["stages"]
    coords = _coords(golden)
    out = gen.generate_chunks_no_erosion(gen.positions(coords))
    pos = oracle.positions(coords)
    hf, bw = oracle.heightfields(pos)
    g = oracle.gather_heightfields(pos, hf)
    layers = oracle.fix_backward(oracle.layers(pos, g, bw))
    cave = oracle.caves(pos, hf, bw)
    blocks = oracle.fill(pos, hf, bw, layers, cave)
    assert np.abs(np_(out["hf"]) - hf).max() <= HEIGHT_TOL
    for name, ref in (("hf", hf), ("bw", bw), ("gathered", g), ("layers", layers), ("cave", cave), ("blocks", blocks)):
        got = np_(out[name]).reshape(ref.shape)
        assert_bit_equal(got, ref, f"{name} vs oracle")
        for i in range(len(coords)):
            assert sha(got[i]) == str(s["sha_" + name][i]), f"{name} golden sha, chunk {coords[i]}"
    assert_bit_equal(np_(out["blocks"])[coords.index((0, 0))], s["blocks_0_0"], "golden blocks (0,0)")


def test_far_coordinates_match_oracle(gen, oracle):
    """World-edge coordinates: block coordinates up to 6.4e8 (fp32 ulp 64, several columns share one float) and simplex lattice
    cells beyond 2^24, where the device's LDS noise tables are out of their domain and the arithmetic path takes over lane by lane.
    Every stage output is still bit-exact vs the oracle."""
    coords = [(2_000_000, -2_000_000), (40_000_000, 39_999_999), (-40_000_000, 123), (1_048_576, -1_048_577)]
    out = gen.generate_chunks_no_erosion(gen.positions(coords))
    pos = oracle.positions(coords)
    hf, bw = oracle.heightfields(pos)
    g = oracle.gather_heightfields(pos, hf)
    layers = oracle.fix_backward(oracle.layers(pos, g, bw))
    cave = oracle.caves(pos, hf, bw)
    blocks = oracle.fill(pos, hf, bw, layers, cave)
    for name, ref in (("hf", hf), ("bw", bw), ("gathered", g), ("layers", layers), ("cave", cave), ("blocks", blocks)):
        assert_bit_equal(np_(out[name]).reshape(ref.shape), ref, f"{name} vs oracle at far coordinates")


def test_abi_variants_agree(gen, oracle):
    """mmgen_generate_heightfields (reference kernel shape) == the gathered variant; caller-gathered input == fused gather."""
    coords = [(5, 5), (-9, 2), (300, -300)]
    pos = gen.positions(coords)
    hf1, bw1 = gen.generate_heightfields(pos)
    hf2, bw2, g2 = gen.generate_heightfields(pos, gathered=True)
    assert_bit_equal(np_(hf1), np_(hf2), "hf variants"); assert_bit_equal(np_(bw1), np_(bw2), "bw variants")
    g_host = oracle.gather_heightfields(oracle.positions(coords), np_(hf1))
    assert_bit_equal(np_(g2), g_host, "gathered")


def test_empty_and_single(gen):
    import torch
    pos0 = torch.zeros((0, 2), dtype=torch.int32, device=gen.device)
    out = gen.generate_chunks_no_erosion(pos0)
    assert out["blocks"].shape[0] == 0
    one = gen.generate_chunks_no_erosion(gen.positions([(7, -7)]))
    assert one["blocks"].shape == (1, 98304)


def test_error_behaviour(gen):
    """Null device pointers are rejected with hipErrorInvalidValue (=1), like a failed CUDA call in the reference."""
    assert gen.lib.mmgen_generate_heightfields(None, 4, None, None, None) == 1
    assert gen.lib.mmgen_generate_caves(None, None, None, -1, None, None) == 1
    assert b"invalid" in gen.lib.mmgen_error_string(1).lower()


# ------------------------------------------------------------------------------------------------ full-size properties (config 2: 256 chunks)
def test_full_batch_properties(gen, oracle):
    import torch
    coords = [(x, z) for z in range(16) for x in range(16)]
    pos = gen.positions(coords)
    a = gen.generate_chunks_no_erosion(pos)
    torch.cuda.synchronize()
    # determinism: second run identical
    b = gen.generate_chunks_no_erosion(pos)
    for k in ("hf", "layers", "cave", "blocks"):
        assert torch.equal(a[k], b[k]), k
    # batch-split and permutation invariance (chunks are independent given position)
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(1)).to(gen.device)
    c = gen.generate_chunks_no_erosion(pos[perm][:100])
    assert torch.equal(c["blocks"], a["blocks"][perm][:100])
    assert torch.equal(c["cave"], a["cave"][perm][:100])
    # structure: bedrock floor, air above max(height, sea level), gathered ring == neighbour chunk heights
    blocks = a["blocks"].view(256, 256, 384)
    assert bool((blocks[:, :, 0] == 56).all())
    top = torch.clamp(a["hf"].to(torch.int32), min=128)
    ys = torch.arange(384, device=gen.device).view(1, 1, 384)
    assert bool((blocks[ys.expand(256, 256, 384) > top.unsqueeze(-1)] == 0).all())
    hf = a["hf"].view(16, 16, 16, 16)          # [cz][cx][z][x]
    g = a["gathered"].view(16, 16, 18, 18)
    assert torch.equal(g[3, 4, 1:17, 17], hf[3, 5, :, 0]) and torch.equal(g[3, 4, 0, 1:17], hf[2, 4, 15, :]) and g[3, 4, 17, 17] == hf[4, 5, 0, 0]
    # spot parity with the oracle on 6 random chunks of the batch
    sel = [3, 77, 130, 201, 255, 16]
    sub = [coords[i] for i in sel]
    opos = oracle.positions(sub)
    ohf, obw = oracle.heightfields(opos)
    og = oracle.gather_heightfields(opos, ohf)
    ol = oracle.fix_backward(oracle.layers(opos, og, obw))
    oc = oracle.caves(opos, ohf, obw)
    ob = oracle.fill(opos, ohf, obw, ol, oc)
    assert_bit_equal(np_(a["blocks"])[sel], ob, "blocks sample of the 256-chunk batch")
    assert_bit_equal(np_(a["cave"])[sel], oc, "cave layers sample")


def test_fill_call_larger_than_its_sub_batches(gen, oracle):
    """mmgen_fill cuts a call into sub-batches of 8 192 chunks for its row lists (and into batches of 16 384 for its queue): one call of
    8 192 + 88 chunks must equal the same chunks filled in small calls, on both sides of the cut, and the oracle on a sample."""
    import torch
    rng = np.random.default_rng(5)
    n = 8192 + 88
    coords = [(int(x), int(z)) for x, z in rng.integers(-3000, 3000, (n, 2))]
    pos = gen.positions(coords)
    hf, bw, g = gen.generate_heightfields(pos, gathered=True)
    layers = gen.fix_backward_layers(gen.generate_layers(g, bw, pos))
    cave = gen.generate_caves(hf, bw, pos)
    big = gen.fill(hf, bw, layers, cave, pos)
    torch.cuda.synchronize()
    for lo, hi in ((0, 40), (8192 - 40, 8192 + 40), (n - 24, n)):
        part = gen.fill(hf[lo:hi].contiguous(), bw[lo:hi].contiguous(), layers[lo:hi].contiguous(), cave[lo:hi].contiguous(), pos[lo:hi].contiguous())
        assert torch.equal(part, big[lo:hi]), (lo, hi)
    sel = [0, 8191, 8192, 8193, n - 1]
    opos = oracle.positions([coords[i] for i in sel])
    ohf, obw = oracle.heightfields(opos)
    ol = oracle.fix_backward(oracle.layers(opos, oracle.gather_heightfields(opos, ohf), obw))
    oc = oracle.caves(opos, ohf, obw)
    assert_bit_equal(np_(big)[sel], oracle.fill(opos, ohf, obw, ol, oc), "blocks on both sides of the sub-batch cut")


# ------------------------------------------------------------------------------------------------ config 3: one erosion zone
def _zone_planes_oracle(oracle, zone):
    coords = [(zone[0] - 6 + x, zone[1] - 6 + z) for z in range(24) for x in range(24)]
    pos = oracle.positions(coords)
    hf, bw = oracle.heightfields(pos)
    layers = oracle.layers(pos, oracle.gather_heightfields(pos, hf), bw)
    planes = np.concatenate([layers[:, 12:20, :], hf[:, None, :]], axis=1)            # [576, 9, 256]
    planes = planes.reshape(24, 24, 9, 16, 16).transpose(2, 0, 3, 1, 4).reshape(9, 384 * 384)
    return np.ascontiguousarray(planes), hf, bw, layers


@pytest.mark.parametrize("zone", [(0, 0), (-984, 3072)])      # origin (config 3) and a steep MOUNTAINS zone
def test_erosion_zone_matches_oracle(gen, oracle, zone):
    """BASELINE config 3: one 24x24-chunk erosion zone (576 chunks, 6-chunk padding): K1+K2 on all 576, relaxation to
    convergence; eroded planes, accumulated heights and the pass count are bit-exact vs the oracle."""
    import torch
    planes, hf, bw, layers = _zone_planes_oracle(oracle, zone)
    coords = gen.zone_area_coords(*zone)
    pos = gen.positions(coords)
    ghf, gbw, gg = gen.generate_heightfields(pos, gathered=True)
    glayers = gen.generate_layers(gg, gbw, pos)
    assert_bit_equal(np_(glayers), layers, "raw layers of the zone area")
    packed = gen.pack_zone_planes(glayers, ghf)
    assert_bit_equal(np_(packed)[0, : 9 * 147456].reshape(9, -1), planes, "packed zone planes (copyLayers)")
    ref = planes.copy()
    ref_passes = oracle.erode_zone_planes(ref)
    out, passes, acc = gen.erode_zones(packed, want_acc=True)
    torch.cuda.synchronize()
    got = np_(out)[0, : 9 * 147456].reshape(9, -1)
    assert passes == ref_passes, (passes, ref_passes)
    assert_bit_equal(got, ref, "eroded planes")
    assert (got[8] == planes[8]).all()
    assert (ref[:8] != planes[:8]).any(), "erosion changed nothing: test zone is degenerate"
    # accumulated heights = total lift of the lowest layer's column... (sum of per-layer start changes); must be finite and >= 0 somewhere
    assert np.isfinite(np_(acc)).all()
    # idempotence at the fixed point: a second erosion of the eroded planes converges in exactly 8 passes per... at least 8 passes
    again, p2 = gen.erode_zones(out.clone())
    assert p2 >= 8


def test_erosion_batched_zones_equal_single(gen):
    import torch
    zones = [(0, 0), (12, 0)]
    packs = []
    for z in zones:
        pos = gen.positions(gen.zone_area_coords(*z))
        hf, bw, g = gen.generate_heightfields(pos, gathered=True)
        packs.append(gen.pack_zone_planes(gen.generate_layers(g, bw, pos), hf))
    both = torch.cat(packs, dim=0).contiguous()
    singles = [gen.erode_zones(p.clone())[0] for p in packs]
    batched, _ = gen.erode_zones(both)
    for i in range(2):
        assert torch.equal(batched[i], singles[i][0])


# ------------------------------------------------------------------------------------------------ pins to reference-derived data
def test_device_tables_match_reference_literals(gen, golden):
    """The constant rule tables compiled into libmmgen == the literals of the reference's BiomeUtils::init (tests/golden/ref_tables.npz,
    extracted from biomeFuncs.hpp:725-1256 by tools/extract_ref_tables.py), and the gather order == chunk.cu:1158-1167."""
    from test_oracle_math import check_tables_against_reference
    t = gen.debug_tables()
    check_tables_against_reference(t, golden["ref_tables"], "device")
    assert np.array_equal(t["gather_offsets"], golden["ref_tables"]["gather_offsets"])
    assert (t["feature_reach"][1:] >= 0).all() and t["feature_reach"].max() <= 127      # reach travels in 8 bits (k_apply_features)


def test_device_rng_matches_real_thrust(gen, golden):
    """rng3 / rng4 + u01 on the device == the real rocThrust engine composed like rng.hpp:86-96 (tests/golden/thrust_probe.npz)."""
    t = golden["thrust_probe"]
    xyzw = t["xyzw"].copy()
    assert_bit_equal(gen.debug_probe("rng4_u01", xyzw, len(xyzw), 4), t["u01_4"], "u01 (4-arg seed) vs thrust")
    x3 = xyzw.copy(); x3[:, 3] = np.int32(-2 ** 31)
    assert_bit_equal(gen.debug_probe("rng4_u01", x3, len(x3), 4), t["u01_3"], "u01 (3-arg seed) vs thrust")


def test_stage_calls_on_two_streams_do_not_share_scratch(gen):
    """mmgen_generate_caves keeps per-column scratch inside the library: it is keyed by (device, stream), so the same batch generated
    concurrently on two streams (different positions) gives the results of the single-stream runs."""
    import torch
    pos_a = gen.positions([(x, 7) for x in range(24)])
    pos_b = gen.positions([(-900 + x, -33) for x in range(24)])
    ref = []
    for pos in (pos_a, pos_b):
        hf, bw = gen.generate_heightfields(pos)
        ref.append((hf, bw, gen.generate_caves(hf, bw, pos).clone()))
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = [None, None]
    for _ in range(3):
        for i, (st, pos) in enumerate(((s1, pos_a), (s2, pos_b))):
            with torch.cuda.stream(st):
                outs[i] = gen.generate_caves(ref[i][0], ref[i][1], pos)
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(outs[i], ref[i][2])


def test_erosion_launch_that_fills_the_chip_exactly(gen):
    """16 zones in one call: 48 workgroups per zone = 768 = every slot the occupancy query promises (three per CU).  If the chip held fewer,
    the workgroups that cannot start would wait for slots of the XCD the dispatcher has bound them to while spinning workgroups hold them
    (profiles/LOG.md, round 5) - the call would end with MMGEN_ERROR_EROSION_STALL.  It must simply work, and give each zone what a
    single-zone call gives it."""
    import torch
    zones = [(12 * (i % 4), 12 * (i // 4)) for i in range(16)]
    packs = []
    for z in zones:
        pos = gen.positions(gen.zone_area_coords(*z))
        hf, bw, g = gen.generate_heightfields(pos, gathered=True)
        packs.append(gen.pack_zone_planes(gen.generate_layers(g, bw, pos), hf))
    batched, passes = gen.erode_zones(torch.cat(packs, dim=0).contiguous())
    for i in (0, 7, 15):
        single, p1 = gen.erode_zones(packs[i].clone())
        assert torch.equal(batched[i], single[0]) and p1 <= passes


@pytest.mark.gpu
def test_starved_relaxation_gives_up_and_the_rescue_pass_finishes_its_zones(gen, oracle):
    """The relaxation's workgroups wait for each other on the device; a launch whose groups can never become complete (here: every group
    waits for one workgroup more than it has) must end itself after the timeout - never spin forever - and the rescue pass enqueued behind
    it must then relax the zones it abandoned: same planes, same pass count as a healthy launch, no error (the reference's host loop cannot
    stall either: chunk.cu:682-705).  Per-stage ABI (one zone, two zones) and the region path (a stall inside mmgen_region_generate)."""
    import torch
    pos = gen.positions(gen.zone_area_coords(0, 0))
    hf, bw, g = gen.generate_heightfields(pos, gathered=True)
    packed = gen.pack_zone_planes(gen.generate_layers(g, bw, pos), hf)
    good, passes = gen.erode_zones(packed.clone())
    region_good = gen.generate_region(3, 5, 2, 2)["blocks"].clone()
    stalls0, rescued0 = gen.erosion_stalls()
    gen.debug_erosion_stall(1, 200)
    try:
        again, p2 = gen.erode_zones(packed.clone())
        two, p3 = gen.erode_zones(torch.cat([packed, packed], dim=0).contiguous())
        region_again = gen.generate_region(3, 5, 2, 2)["blocks"].clone()
        torch.cuda.synchronize()
    finally:
        gen.debug_erosion_stall(0, 0)
    assert p2 == passes and torch.equal(again, good)
    assert p3 == passes and torch.equal(two[0], good[0]) and torch.equal(two[1], good[0])
    assert torch.equal(region_again, region_good)
    ref = oracle.generate_region(3, 5, 2, 2, erosion=True, features=True, decorators=True)["blocks"]
    assert np.array_equal(region_again.cpu().numpy(), ref)
    gen.generate_region(3, 5, 1, 1)                      # (a region learns of its stall at its next call)
    stalls1, rescued1 = gen.erosion_stalls()
    assert stalls1 >= stalls0 + 3 and rescued1 >= rescued0 + 3, ((stalls0, rescued0), (stalls1, rescued1))
    # healthy again afterwards, and a healthy launch rescues nothing
    once, p4 = gen.erode_zones(packed.clone())
    assert p4 == passes and torch.equal(once, good) and gen.erosion_stalls() == (stalls1, rescued1)
