"""CPU tests: the oracle's math layer against the REAL glm golden vectors, frozen KATs and libm accuracy bounds."""
import ctypes
import os

import numpy as np
import pytest
from conftest import assert_bit_equal, ROOT
from oracle_binding import _p


def test_simplex_matches_real_glm_golden(oracle, golden):
    g = golden["glm_probe"]
    s2 = np.zeros(len(g["xy"]), np.float32); oracle.lib.mmo_simplex2(len(s2), _p(np.ascontiguousarray(g["xy"])), _p(s2))
    s3 = np.zeros(len(g["xyz"]), np.float32); oracle.lib.mmo_simplex3(len(s3), _p(np.ascontiguousarray(g["xyz"])), _p(s3))
    assert_bit_equal(s2, g["simplex2"], "simplex2 vs glm")
    assert_bit_equal(s3, g["simplex3"], "simplex3 vs glm")


def test_simplex_matches_live_glm_probe_if_present(oracle):
    """In the build container the real-glm probe is rebuilt from /root/reference; compare on fresh random points."""
    path = os.path.join(ROOT, "oracle", "_ref", "libglmprobe.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built here (no /root/reference)")
    ref = ctypes.CDLL(path)
    rng = np.random.default_rng(99)
    for scale in (1.0, 300.0, 2e5):
        xy = (rng.uniform(-1, 1, (50000, 2)) * scale).astype(np.float32)
        xyz = (rng.uniform(-1, 1, (50000, 3)) * scale).astype(np.float32)
        a = np.zeros(50000, np.float32); b = np.zeros(50000, np.float32)
        oracle.lib.mmo_simplex2(50000, _p(xy), _p(a)); ref.ref_simplex2(50000, _p(xy), _p(b)); assert_bit_equal(a, b, "simplex2 live")
        oracle.lib.mmo_simplex3(50000, _p(xyz), _p(a)); ref.ref_simplex3(50000, _p(xyz), _p(b)); assert_bit_equal(a, b, "simplex3 live")


def test_glm_helpers_match_golden(golden):
    """smoothstep / mix / mod / normalize / length of glm restated in numpy fp32 with glm's operation order."""
    g = golden["glm_probe"]
    f = np.float32
    e0, e1, x = g["e0e1x"][:, 0], g["e0e1x"][:, 1], g["e0e1x"][:, 2]
    with np.errstate(all="ignore"):
        t = np.minimum(np.maximum((x - e0) / (e1 - e0), f(0)), f(1))
        ss = t * t * (f(3) - f(2) * t)
    ok = np.isfinite(g["smoothstep"])
    assert_bit_equal(ss[ok], g["smoothstep"][ok], "smoothstep")
    mx = e0 * (f(1) - x) + e1 * x
    assert_bit_equal(mx, g["mix"], "mix")
    a, b = g["ab"][:, 0], g["ab"][:, 1]
    assert_bit_equal(a - b * np.floor(a / b), g["mod"], "mod")
    v = g["v3"]
    d = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2]
    assert_bit_equal(np.sqrt(d), g["length"], "length")
    assert_bit_equal(v * (f(1) / np.sqrt(d))[:, None], g["normalize"], "normalize")


def test_kat_frozen(oracle, golden):
    """The oracle still produces the frozen known-answer vectors (contract drift detector)."""
    k = golden["oracle_kat"]
    L = oracle.lib
    out = np.zeros_like(k["hash_out"]); L.mmo_hash(len(out), _p(np.ascontiguousarray(k["hash_in"])), _p(out)); assert_bit_equal(out, k["hash_out"], "hash")
    seeds = np.ascontiguousarray(k["rng_seeds"])
    for use_w in (0, 1):
        u = np.zeros((len(seeds), 4), np.float32); L.mmo_rng_u01(len(seeds), _p(seeds), use_w, 4, _p(u)); assert_bit_equal(u, k[f"u01_w{use_w}"], "u01")
    x = np.ascontiguousarray(k["trig_in"]); s = np.zeros_like(x); L.mmo_sinf(len(x), _p(x), _p(s)); assert_bit_equal(s, k["sin"], "sin")
    xyz = np.ascontiguousarray(k["fbm3_in"]); sc = np.zeros(len(xyz), np.float32)
    L.mmo_special_cave_noise(len(xyz), _p(xyz), _p(sc)); assert_bit_equal(sc, k["special_cave_noise"], "specialCaveNoise")


def test_hash_and_minstd_known_answers(oracle):
    """hash() of rng.hpp:69-78 and thrust minstd_rand re-derived independently in Python integers."""
    def h(a):
        M = 0xFFFFFFFF
        a = ((a + 0x7ed55d16) + (a << 12)) & M; a = ((a ^ 0xc761c23c) ^ (a >> 19)) & M; a = ((a + 0x165667b1) + (a << 5)) & M
        a = ((a + 0xd3a2646c) ^ (a << 9)) & M; a = ((a + 0xfd7046c5) + (a << 3)) & M; a = ((a ^ 0xb55a4f09) ^ (a >> 16)) & M
        return a
    xs = np.array([0, 1, 12345, 0x80000000, 0xFFFFFFFF, 0x7ed55d16], np.uint32)
    out = np.zeros_like(xs); oracle.lib.mmo_hash(len(xs), _p(xs), _p(out))
    assert [int(v) for v in out] == [h(int(v)) for v in xs]
    # minstd: seed -> x = 48271 x mod (2^31-1); u01 = float(x-1)/2^31 ; 4-argument seeding with negative coordinates
    seeds = np.array([[3, 4, 5, 6], [-7, 200, -9, 190249401]], np.int32)
    u = np.zeros((2, 3), np.float32); oracle.lib.mmo_rng_u01(2, _p(seeds), 1, 3, _p(u))
    for row, (x, y, z, w) in zip(u, seeds.tolist()):
        M = 0xFFFFFFFF
        s = (h((0x80000000 | ((x << 22) & M) | ((y << 11) & M) | (w & M)) & M) ^ h(z & M)) % 2147483647 or 1
        for got in row:
            s = (s * 48271) % 2147483647
            assert np.float32(got) == np.float32(np.float32(s - 1) / np.float32(2147483648.0))


def _ulps(a, b):
    ai = a.view(np.int32).astype(np.int64); bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7fffffff), ai); bi = np.where(bi < 0, -(bi & 0x7fffffff), bi)
    return np.abs(ai - bi)


def test_deterministic_libm_accuracy(oracle):
    """The contract libm is within 2 ulp of the correctly rounded result (numpy fp64 -> fp32) on the argument ranges the
    path uses: sin/cos up to 1e11 (sin-hash dot products), pow on [0,2]^[0.1,4], atan2, acos."""
    rng = np.random.default_rng(5)
    L = oracle.lib
    for scale in (1, 1e3, 1e7, 1e11):
        x = (rng.uniform(-1, 1, 200000) * scale).astype(np.float32)
        s = np.zeros_like(x); c = np.zeros_like(x); L.mmo_sinf(len(x), _p(x), _p(s)); L.mmo_cosf(len(x), _p(x), _p(c))
        assert np.abs(s.astype(np.float64) - np.sin(x.astype(np.float64))).max() < 1.3e-7
        assert np.abs(c.astype(np.float64) - np.cos(x.astype(np.float64))).max() < 1.3e-7
    x = rng.uniform(0, 2, 100000).astype(np.float32); y = rng.uniform(0.1, 4, 100000).astype(np.float32)
    p = np.zeros_like(x); L.mmo_powf(len(x), _p(x), _p(y), _p(p))
    assert _ulps(p, np.power(x.astype(np.float64), y.astype(np.float64)).astype(np.float32)).max() <= 1
    yy = rng.uniform(-5, 5, 100000).astype(np.float32); xx = rng.uniform(-5, 5, 100000).astype(np.float32)
    a = np.zeros_like(yy); L.mmo_atan2f(len(yy), _p(yy), _p(xx), _p(a))
    assert _ulps(a, np.arctan2(yy.astype(np.float64), xx.astype(np.float64)).astype(np.float32)).max() <= 1
    xc = rng.uniform(-1, 1, 100000).astype(np.float32); ac = np.zeros_like(xc); L.mmo_acosf(len(xc), _p(xc), _p(ac))
    assert _ulps(ac, np.arccos(xc.astype(np.float64)).astype(np.float32)).max() <= 1


def test_tables(oracle, golden):
    """Frozen tan(angle of repose) constants are the correctly rounded fp64 values; weights tables match the fixture."""
    k = golden["oracle_kat"]
    mi = np.zeros((20, 4), np.float32); oracle.lib.mmo_tables_material_infos(_p(mi))
    assert_bit_equal(mi, k["material_infos"], "material infos")
    for row, deg in zip(mi[12:], (55, 40, 45, 40, 30, 35, 65, 45)):
        rad = np.float32(deg) * np.float32(0.01745329251994329576923690768489)
        assert row[2] == np.float32(np.tan(np.float64(rad)))
    bm = np.zeros((24, 20), np.float32); oracle.lib.mmo_tables_biome_material_weights(_p(bm))
    assert_bit_equal(bm, k["biome_material_weights"], "biome material weights")
    assert bm[8, 7] == np.float32(3.2) and bm[0, 15] == 0 and bm[23, 12] == 1    # SAVANNA/TERRACOTTA, CORAL_REEF/DIRT, MOUNTAINS/GRAVEL


def test_noise_table_domains():
    """The HIP path serves glm's permute() chain and the corner gradients of simplex noise from LDS tables built per workgroup
    with the same fp32 operations (csrc/mm_noise.cuh).  Exactness needs only that every lookup index is inside the tables:
      * mod289 (x - floor(x * (1/289)) * 289, fp32) of an integer-valued |x| < 2^24 lies in [-1, 289]   (the guard the device checks),
      * glm::mod(x, 289) (x - 289 * floor(x / 289)) of the same lies in [0, 288]                           (simplex2),
      * permute of every integer in [-16, 700) is an integer in [0, 288]: so chained indices stay in [-1, 578], inside the table's [-2, 580)
        (simplex3: no reduction of an index) resp. wrap into [0, 290] of simplex2's 296 gradient entries."""
    f = np.float32
    xs = np.arange(-2 ** 24 + 1, 2 ** 24, dtype=np.int64).astype(f)
    m = (xs - np.floor(xs * (f(1.0) / f(289.0))) * f(289.0)).astype(f)
    assert m.min() >= -1 and m.max() <= 289 and np.array_equal(m, np.floor(m))
    g = (xs - f(289.0) * np.floor(xs / f(289.0))).astype(f)
    assert g.min() >= 0 and g.max() <= 288
    x = np.arange(-16, 700, dtype=f)
    t = ((x * f(34.0)) + f(1.0)) * x
    p = (t - np.floor(t * (f(1.0) / f(289.0))) * f(289.0)).astype(f)
    assert p.min() >= 0 and p.max() <= 288 and np.array_equal(p, np.floor(p))
    # chained index range vs the one table of csrc/mm_noise.cuh (entry v = {gradient of permute(v), 16 * permute(v)}, v in [-2, 580)):
    # z and z + 1 >= -1; p + y + o and b + x + o with p, b in [0, 288], x, y in [-1, 289], o in {0, 1}
    assert -1 + 0 + 0 >= -2 and 288 + 289 + 1 < 580
    # every entry is computed from its own argument: permute(v) for v up to 579 is exact integer arithmetic in fp32 (< 2^24)
    assert (579 * 34 + 1) * 579 < 2 ** 24


def test_simplex2_remainder_without_division():
    """csrc/mm_noise.cuh simplex2_inl: inside its table domain (integer lattice coordinates |i| < 2^21) the device takes glm::mod(i, 289) as
    i - 289 * floor(fl(fl(i + 0.5) * fl(1 / 289))) with one fused multiply-add.  Every integer of the domain (and a margin): the fp32 floor
    is the true floor(i / 289), hence the remainder (a small integer, exact under the single rounding of the fma) is glm's value."""
    f = np.float32
    i = np.arange(-2 ** 21 - 64, 2 ** 21 + 65, dtype=np.int64)
    x = i.astype(f)
    assert np.array_equal(x.astype(np.int64), i)
    q = np.floor(((x + f(0.5)).astype(f) * (f(1.0) / f(289.0))).astype(f))
    assert np.array_equal(q.astype(np.int64), i // 289)
    r = i - 289 * q.astype(np.int64)                                # what the fma returns: exact
    g = (x - f(289.0) * np.floor(x / f(289.0))).astype(f)            # glm::mod(x, 289) in fp32
    assert np.array_equal(r, g.astype(np.int64)) and r.min() == 0 and r.max() == 288


def _oracle_tables(oracle):
    """The oracle's rule tables in the numeric layout of tools/extract_ref_tables.py."""
    t = {}
    br = np.zeros((24, 6), np.uint8); cr = np.zeros((5, 4), np.uint8); gr = np.zeros(24, np.uint8)
    oracle.lib.mmo_tables_rules(_p(br), _p(cr), _p(gr))
    mi = np.zeros((20, 4), np.float32); oracle.lib.mmo_tables_material_infos(_p(mi))
    bm = np.zeros((24, 20), np.float32); oracle.lib.mmo_tables_biome_material_weights(_p(bm))
    fb = np.zeros((21, 2), np.int32); cfb = np.zeros((10, 2), np.int32)
    sg = np.zeros((24, 4, 11), np.float32); cg = np.zeros((5, 3, 9), np.float32)
    dg = np.zeros((24, 7, 10), np.float32); cdg = np.zeros((5, 6, 10), np.float32)
    oracle.lib.mmo_tables_gens(_p(fb), _p(cfb), _p(sg), _p(cg), _p(dg), _p(cdg))
    t.update(biome_rules=br, cave_rules=cr, grass=gr, material_infos=mi, biome_material_weights=bm, feature_bounds=fb,
             cave_feature_bounds=cfb, surf_gens=sg, cave_gens=cg, deco_gens=dg, cave_deco_gens=cdg)
    return t


def check_tables_against_reference(tables, ref, what):
    """Shared by the CPU (oracle) and GPU (device constant tables) tests: every rule table == the literals extracted from the
    reference's BiomeUtils::init (tests/golden/ref_tables.npz, made by tools/extract_ref_tables.py)."""
    for name in ("biome_rules", "grass", "biome_material_weights", "feature_bounds", "cave_feature_bounds", "surf_gens", "cave_gens",
                 "deco_gens", "cave_deco_gens"):
        got, want = np.asarray(tables[name]), ref[name]
        assert got.shape == want.shape, (what, name, got.shape, want.shape)
        assert np.array_equal(got.astype(want.dtype), want), f"{what}: {name} differs from the reference's literals at {np.argwhere(got.astype(want.dtype) != want)[:5]}"
    if "cave_rules" in tables:
        assert np.array_equal(tables["cave_rules"], ref["cave_rules"]), f"{what}: cave rules"
    mi, rmi = np.asarray(tables["material_infos"], np.float32), ref["material_infos"]
    assert np.array_equal(mi[:, [0, 1, 3]], rmi[:, [0, 1, 3]]), f"{what}: material block / thickness / scale"
    assert np.array_equal(mi[:12, 2], rmi[:12, 2]), f"{what}: stratified noise amplitudes"
    # eroded materials: the reference stores tanf(glm::radians(degrees)) (biomeFuncs.hpp:840-845); ours is the correctly rounded tangent
    rad = rmi[12:, 2].astype(np.float32) * np.float32(0.01745329251994329576923690768489)
    assert np.array_equal(mi[12:, 2], np.tan(rad.astype(np.float64)).astype(np.float32)), f"{what}: tan(angle of repose)"


def test_tables_match_reference_literals(oracle, golden):
    check_tables_against_reference(_oracle_tables(oracle), golden["ref_tables"], "oracle")
    assert golden["ref_tables"]["enum_counts"].tolist()[:5] == [24, 5, 20, 21, 10]


def test_minstd_matches_real_thrust(oracle, golden):
    """oracle Rng == the real rocThrust minstd_rand + uniform_real_distribution<float> (tests/golden/thrust_probe.npz, generated by
    executing /opt/rocm/include/thrust host-side): bare engine incl. the seed-0 -> 1 rule and seeds >= m, and the reference's two
    seeding compositions (rng.hpp:86-96) incl. negative / far coordinates."""
    t = golden["thrust_probe"]
    seeds = np.ascontiguousarray(t["seeds"])
    raw = np.zeros((len(seeds), 4), np.uint32); u01 = np.zeros((len(seeds), 4), np.float32)
    oracle.lib.mmo_minstd(len(seeds), _p(seeds), 4, _p(raw), _p(u01))
    assert np.array_equal(raw, t["raw"]) and t["raw"][0, 0] == 48271
    assert_bit_equal(u01, t["u01"][:, :4], "u01 of the bare engine")
    xyzw = np.ascontiguousarray(t["xyzw"])
    for use_w, key in ((0, "u01_3"), (1, "u01_4")):
        out = np.zeros((len(xyzw), 4), np.float32)
        oracle.lib.mmo_rng_u01(len(xyzw), _p(xyzw), use_w, 4, _p(out))
        assert_bit_equal(out, t[key], f"makeSeededRandomEngine {3 + use_w}-arg")
    # third derivation, in Python integers: x <- 48271 x mod (2^31 - 1), u = float32(x - 1) / float32(2^31 - 2 + 1)
    m = 2 ** 31 - 1
    for s, r in zip(seeds[:16].tolist(), t["raw"][:16].tolist()):
        x = s % m or 1
        for want in r:
            x = 48271 * x % m
            assert x == want


def test_simplex_bounds(oracle):
    """cave_biome on the device (csrc/mm_biome.cuh) skips evaluations whose outcome is decided by |fbm2<3>| <= 0.875 * B2 (the supremum computed here is 1.035; the device
    uses 1.16, slack for rounding included).
    The bound is adversarial, not statistical: simplex2 = 130 * sum_k m_k^4 (g_k . x_k) <= 130 * max|g| * sup sum_k (0.5 - r_k^2)_+^4 r_k
    (every gradient aligned with its corner offset).  max|g| comes from the 289-entry gradient table exactly as the device builds it;
    the supremum over the simplex is taken on a 3000 x 3000 grid of the skewed unit cell plus the Lipschitz slack of the grid
    (|d/dr (0.5 - r^2)^4 r| <= 0.0625, three corners, half a cell diagonal).  A sample of the oracle's simplex2 stays below it, too."""
    f = np.float32
    p = np.arange(0, 289, dtype=f)
    t = p * f(0.024390243902439)
    X = f(2) * (t - np.floor(t)) - f(1)
    h = np.abs(X) - f(0.5)
    a = X - np.floor(X + f(0.5))
    nrm = f(1.79284291400159) - f(0.85373472095314) * (a * a + h * h)
    gmax = float((np.sqrt(a.astype(np.float64) ** 2 + h.astype(np.float64) ** 2) * nrm.astype(np.float64)).max())
    n = 3000
    u = (np.arange(n) + 0.5) / n
    U, V = np.meshgrid(u, u, indexing="ij")
    c0 = (3 - np.sqrt(3)) / 6
    tt = (U + V) * c0
    x0, y0 = U - tt, V - tt
    i1x = (x0 > y0).astype(np.float64)
    hk = lambda x, y: np.maximum(0.5 - (x * x + y * y), 0) ** 4 * np.sqrt(x * x + y * y)
    F = hk(x0, y0) + hk(x0 - i1x + c0, y0 - (1 - i1x) + c0) + hk(x0 - 1 + 2 * c0, y0 - 1 + 2 * c0)
    slack = 3 * 0.0625 * (np.sqrt(2) / n) / 2
    bound = 130.0 * gmax * (float(F.max()) + slack)
    assert bound < 1.06, bound
    # the device uses MM_SIMPLEX2_BOUND = 1.16 and only within 32 768 blocks of the origin: a point evaluated up to 0.006 outside its cell (the
    # rounding of the skew sum at arguments up to 2.6e4) adds at most 0.006 * 3 * 0.0625 to the sum (csrc/mm_noise.cuh)
    assert 130.0 * gmax * (float(F.max()) + slack + 0.006 * 3 * 0.0625) < 1.16
    rs = np.random.RandomState(3)
    xy = (rs.rand(2_000_000, 2).astype(f) - f(0.5)) * f(2000.0)
    out = np.zeros(len(xy), f)
    oracle.lib.mmo_simplex2(len(xy), _p(xy), _p(out))
    assert float(np.abs(out).max()) < bound


def test_simplex3_bound(oracle):
    """cave_huge on the device (csrc/mmgen_kernels.hip) leaves its octave loop once the octaves still to come cannot move the result,
    using |simplex3| <= B3 (the supremum computed here is 1.226; the device uses 1.37, slack for rounding included).  Adversarial bound like test_simplex_bounds: simplex3 = 42 * sum_k m_k^4 (g_k . x_k)
    <= 42 * sup|g| * sup sum_k (0.6 - r_k^2)_+^4 r_k.  sup|g|: a gradient is p * (1.79284291400159 - 0.85373472095314 |p|^2), whose length
    s (1.7928... - 0.8537... s^2) is at most 1.00001 whatever p is.  The supremum over the simplex: coarse grid of the unit cell (160^3),
    then 8^3 sub-grids of every coarse cell that could still hold the maximum, plus the Lipschitz slack of the fine grid
    (|d/dr (0.6 - r^2)^4 r| <= 0.1296, four corners, half a cell diagonal).  A sample of the oracle's simplex3 stays below it, too."""
    s_ = np.linspace(0.0, 2.0, 2_000_001)
    gmax = float((s_ * (1.79284291400159 - 0.85373472095314 * s_ * s_)).max())
    assert gmax < 1.00001

    def F(v):
        i = np.floor(v + v.sum(-1, keepdims=True) / 3.0)
        x0 = v - i + i.sum(-1, keepdims=True) / 6.0
        g = (x0[..., [1, 2, 0]] <= x0).astype(np.float64)         # step(x0.yzx, x0.xyz)
        lz = (1.0 - g)[..., [2, 0, 1]]
        i1, i2 = np.minimum(g, lz), np.maximum(g, lz)
        tot = 0.0
        for x in (x0, x0 - i1 + 1 / 6.0, x0 - i2 + 1 / 3.0, x0 - 0.5):
            r2 = (x * x).sum(-1)
            tot = tot + np.maximum(0.6 - r2, 0) ** 4 * np.sqrt(r2)
        return tot

    n, m = 160, 8
    h = 1.0 / n
    u = (np.arange(n) + 0.5) * h
    slack = 4 * 0.1296 * (np.sqrt(3) * h / 2)
    vals, pts = [], []
    for a in u:
        V = np.stack(np.meshgrid([a], u, u, indexing="ij"), -1).reshape(-1, 3)
        vals.append(F(V)); pts.append(V)
    vals, pts = np.concatenate(vals), np.concatenate(pts)
    sel = pts[vals + slack > vals.max()]
    sub = (np.arange(m) + 0.5) / m * h - h / 2
    S = np.stack(np.meshgrid(sub, sub, sub, indexing="ij"), -1).reshape(-1, 3)
    fine = max(float(F((sel[k:k + 2000, None, :] + S[None]).reshape(-1, 3)).max()) for k in range(0, len(sel), 2000))
    bound = 42.0 * 1.00001 * (fine + 4 * 0.1296 * (np.sqrt(3) * h / m / 2))
    assert bound < 1.23, bound
    # the device uses MM_SIMPLEX3_BOUND = 1.37 inside its pruning domain: 0.006 outside the cell adds at most 0.006 * 4 * 0.1296 to the sum
    assert 42.0 * 1.00001 * (fine + 4 * 0.1296 * (np.sqrt(3) * h / m / 2) + 0.006 * 4 * 0.1296) < 1.37
    f = np.float32
    rs = np.random.RandomState(4)
    xyz = (rs.rand(2_000_000, 3).astype(f) - f(0.5)) * f(2000.0)
    out = np.zeros(len(xyz), f)
    oracle.lib.mmo_simplex3(len(xyz), _p(xyz), _p(out))
    assert float(np.abs(out).max()) < bound


def test_worley_search_shortcuts_are_exact():
    """csrc/mm_noise.cuh special_cave_noise, staged path, restated in numpy fp32:
      * the three smallest of the 27 squared distances kept as unsigned integers with med3 / med3 / min updates, in any order, are the
        three smallest floats (squared distances are finite and >= +0: float order == order of the bit patterns);
      * the lower bound of a column of cells, fl(bx + by) with bx in {fl(fx fx), 0, fl(gx gx)}, gx = fl(1 - fx), is never above the squared
        distance the device computes for a cell of that column - so skipping a column whose bound is >= s3 changes nothing."""
    f = np.float32
    rs = np.random.RandomState(5)
    n = 200000
    fx, fy, fz = (rs.rand(n).astype(f) for _ in range(3))
    # edge cases: fractions of exactly 0 and (through rounding of px - floor(px)) 1, cell points of exactly 0 and 1
    fx[:1000] = 0; fy[1000:2000] = 1; fz[2000:3000] = 0
    pts = rs.rand(27, 3, n).astype(f)
    pts[:, :, 3000:3500] = np.round(pts[:, :, 3000:3500])
    gx, gy = (f(1) - fx).astype(f), (f(1) - fy).astype(f)
    bx = {-1: (fx * fx).astype(f), 0: np.zeros(n, f), 1: (gx * gx).astype(f)}
    by = {-1: (fy * fy).astype(f), 0: np.zeros(n, f), 1: (gy * gy).astype(f)}
    d2s = []
    k = 0
    for x in (-1, 0, 1):
        for y in (-1, 0, 1):
            bound = (bx[x] + by[y]).astype(f)
            for z in (-1, 0, 1):
                qx, qy, qz = pts[k]; k += 1
                dx = ((qx if x == 0 else (f(x) + qx).astype(f)) - fx).astype(f)
                dy = ((qy if y == 0 else (f(y) + qy).astype(f)) - fy).astype(f)
                dz = ((qz if z == 0 else (f(z) + qz).astype(f)) - fz).astype(f)
                d2 = (((dx * dx).astype(f) + (dy * dy).astype(f)).astype(f) + (dz * dz).astype(f)).astype(f)
                # the reference's form of the same cell (always adds the offset): identical squared distance
                rx = ((f(x) + qx).astype(f) - fx).astype(f); ry = ((f(y) + qy).astype(f) - fy).astype(f); rz = ((f(z) + qz).astype(f) - fz).astype(f)
                ref = (((rx * rx).astype(f) + (ry * ry).astype(f)).astype(f) + (rz * rz).astype(f)).astype(f)
                assert np.array_equal(d2, ref)
                assert np.all(bound <= d2), (x, y, z)
                d2s.append(d2)
    d2s = np.stack(d2s)                                             # [27][n]
    u = d2s.view(np.uint32).astype(np.int64)
    big = np.int64(0x7f7fffff)
    for order in (np.arange(27), rs.permutation(27), np.arange(27)[::-1]):
        u1 = np.full(n, big); u2 = u1.copy(); u3 = u1.copy()
        med3 = lambda a, b, c: np.minimum(np.maximum(a, b), np.maximum(np.minimum(a, b), c))
        for i in order:
            v = u[i]
            u3 = med3(u2, u3, v); u2 = med3(u1, u2, v); u1 = np.minimum(u1, v)
        srt = np.sort(d2s, axis=0)
        assert np.array_equal(u1.astype(np.uint32).view(f), srt[0]) and np.array_equal(u2.astype(np.uint32).view(f), srt[1])
        assert np.array_equal(u3.astype(np.uint32).view(f), srt[2])
