#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.  Run in the BUILD container only:

    python tests/golden/make_golden.py

Sources of truth:
  * glm_probe.npz   — outputs of the REAL vendored glm 0.9.9.8 (reference external/include/glm), through
                      oracle/_ref/libglmprobe.so (built by oracle/Makefile from the headers where they lie under
                      /root/reference; the reference has no tests or golden vectors of its own).  Pins
                      simplex 2D/3D and the glm helper functions of both the oracle and the HIP kernels.
  * oracle_kat.npz  — known-answer vectors of the CPU oracle (frozen so that any drift of the contract is caught, and so
                      that the GPU box — which has no /root/reference — can check the device math against the same numbers).
  * stages.npz      — stage outputs of the oracle for a set of chunks covering all 24 surface biomes: SHA-256 per array +
                      full arrays for a few chunks.
The fixtures are DATA (inputs and expected outputs); no reference source text is stored.
"""
import ctypes
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_binding import Oracle, _p  # noqa: E402

# chunk coordinates whose 256 columns are 100 % one biome (index = biome id), found by scanning with the oracle
BIOME_CHUNKS = [(-1268, -1773), (-175, 36), (-2, 2569), (1602, 977), (2859, 900), (-1720, 2988), (-860, -570), (2936, 3693),
                (-2105, -2470), (-271, -2278), (-3135, -3526), (3518, 2777), (1767, -1044), (2609, -3227), (3654, -2794),
                (2467, 3337), (1488, -1110), (-88, -3971), (3946, -3906), (2669, -2199), (-3556, -1599), (3227, 152),
                (-24, 2992), (-983, 3072)]
# mixed / transitional chunks (several biomes active), origin, negative coordinates
MIXED_CHUNKS = [(0, 0), (1, 0), (-1, -1), (-3, 7), (100, -250), (-700, 333), (40, 41), (1000, 1000)]
FULL_CHUNKS = [(0, 0), (-175, 36), (2669, -2199)]     # full arrays stored for these


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def positions(rng, n, dims, scales=(1.0, 50.0, 3000.0, 1e6)):
    parts = [(rng.uniform(-1, 1, (n // len(scales), dims)) * s).astype(np.float32) for s in scales]
    return np.ascontiguousarray(np.concatenate(parts))


def make_glm_probe():
    lib_path = os.path.join(ROOT, "oracle", "_ref", "libglmprobe.so")
    if not os.path.exists(lib_path):
        raise SystemExit("oracle/_ref/libglmprobe.so missing: run `make -C oracle` in the build container (needs /root/reference)")
    r = ctypes.CDLL(lib_path)
    rng = np.random.default_rng(20240501)
    xy = positions(rng, 2048, 2)
    xyz = positions(rng, 2048, 3)
    s2 = np.zeros(len(xy), np.float32); r.ref_simplex2(len(xy), _p(xy), _p(s2))
    s3 = np.zeros(len(xyz), np.float32); r.ref_simplex3(len(xyz), _p(xyz), _p(s3))
    e = rng.uniform(-2, 2, (1024, 3)).astype(np.float32)
    ss = np.zeros(1024, np.float32); r.ref_smoothstep(1024, _p(e), _p(ss))
    mx = np.zeros(1024, np.float32); r.ref_mix(1024, _p(e), _p(mx))
    ab = np.ascontiguousarray(np.stack([rng.uniform(-100, 100, 1024), rng.uniform(0.5, 40, 1024)], 1).astype(np.float32))
    md = np.zeros(1024, np.float32); r.ref_mod(1024, _p(ab), _p(md))
    v3 = rng.uniform(-5, 5, (1024, 3)).astype(np.float32)
    nm = np.zeros((1024, 3), np.float32); r.ref_normalize3(1024, _p(v3), _p(nm))
    ln = np.zeros(1024, np.float32); r.ref_length3(1024, _p(v3), _p(ln))
    np.savez_compressed(os.path.join(HERE, "glm_probe.npz"), xy=xy, simplex2=s2, xyz=xyz, simplex3=s3, e0e1x=e, smoothstep=ss, mix=mx,
                        ab=ab, mod=md, v3=v3, normalize=nm, length=ln)
    print("glm_probe.npz", len(xy), len(xyz))


def make_oracle_kat(o):
    L = o.lib
    rng = np.random.default_rng(777)
    out = {}
    hx = rng.integers(0, 2**32, 1024, dtype=np.uint64).astype(np.uint32)
    hx[:4] = [0, 1, 0x80000000, 0xFFFFFFFF]
    hy = np.zeros_like(hx); L.mmo_hash(len(hx), _p(hx), _p(hy))
    out["hash_in"], out["hash_out"] = hx, hy
    seeds = rng.integers(-2**20, 2**20, (512, 4)).astype(np.int32)
    seeds[:4] = [[0, 0, 0, 0], [-1, -1, -1, 7589341], [15, 383, -16, 190249401], [-100000, 5, 100000, 329271348]]
    for use_w in (0, 1):
        u = np.zeros((512, 4), np.float32); L.mmo_rng_u01(512, _p(seeds), use_w, 4, _p(u))
        out[f"u01_w{use_w}"] = u
    out["rng_seeds"] = seeds
    x = np.concatenate([positions(rng, 2048, 1, (1.0, 100.0, 1e5, 1e9)).ravel(), np.float32([0, -0.0, 1e11, -3.5e10])]).astype(np.float32)
    s = np.zeros_like(x); c = np.zeros_like(x); L.mmo_sinf(len(x), _p(x), _p(s)); L.mmo_cosf(len(x), _p(x), _p(c))
    out["trig_in"], out["sin"], out["cos"] = x, s, c
    px = rng.uniform(0, 2, 1024).astype(np.float32); py = rng.choice(np.float32([2.0, 2.4, 0.5, 3.7]), 1024).astype(np.float32)
    pw = np.zeros_like(px); L.mmo_powf(1024, _p(px), _p(py), _p(pw))
    out["pow_x"], out["pow_y"], out["pow"] = px, py, pw
    ay = rng.uniform(-5, 5, 1024).astype(np.float32); ax = rng.uniform(-5, 5, 1024).astype(np.float32)
    at = np.zeros_like(ay); L.mmo_atan2f(1024, _p(ay), _p(ax), _p(at))
    acx = rng.uniform(-1, 1, 1024).astype(np.float32); ac = np.zeros_like(acx); L.mmo_acosf(1024, _p(acx), _p(ac))
    out["atan2_y"], out["atan2_x"], out["atan2"], out["acos_x"], out["acos"] = ay, ax, at, acx, ac
    xy = positions(rng, 1024, 2); xyz = positions(rng, 1024, 3, (1.0, 20.0, 500.0, 20000.0))
    f = np.zeros(len(xy), np.float32); L.mmo_fbm2(len(xy), 5, _p(xy), _p(f)); out["fbm2_in"], out["fbm2_5"] = xy, f
    f3 = np.zeros(len(xyz), np.float32); L.mmo_fbm3(len(xyz), 4, _p(xyz), _p(f3)); out["fbm3_in"], out["fbm3_4"] = xyz, f3
    cells = rng.integers(-30000, 30000, (1024, 3)).astype(np.float32)
    r3 = np.zeros((1024, 3), np.float32); L.mmo_rand3from3(1024, _p(cells), _p(r3)); out["cells"], out["rand3from3"] = cells, r3
    w2 = np.zeros((len(xy), 5), np.float32); L.mmo_worley2(len(xy), _p(xy), _p(w2)); out["worley2"] = w2
    w3 = np.zeros((len(xyz), 5), np.float32); L.mmo_worley3(len(xyz), _p(xyz), _p(w3)); out["worley3"] = w3
    sc = np.zeros(len(xyz), np.float32); L.mmo_special_cave_noise(len(xyz), _p(xyz), _p(sc)); out["special_cave_noise"] = sc
    bpos = (rng.uniform(-1, 1, (64, 2)) * 60000).astype(np.float32)
    bh = np.zeros((24, 64), np.float32)
    for b in range(24):
        L.mmo_biome_height(64, b, _p(bpos), _p(bh[b]))
    out["biome_height_pos"], out["biome_height"] = bpos, bh
    vox = np.stack([rng.integers(-50000, 50000, 2048), rng.integers(1, 200, 2048), rng.integers(-50000, 50000, 2048)], 1).astype(np.int32)
    mh = rng.uniform(70, 230, 2048).astype(np.float32)
    cb = np.zeros(2048, np.uint8); L.mmo_cave_biome(2048, _p(vox), _p(mh), 190249401, _p(cb))
    out["cb_vox"], out["cb_maxheight"], out["cave_biome"] = vox, mh, cb
    obw = rng.choice(np.float32([0, 0, 0, 0.3, 1.0]), 2048).astype(np.float32)
    sg = np.zeros(2048, np.uint8); L.mmo_should_generate_cave(2048, _p(vox), _p(mh), _p(obw), _p(sg))
    out["sg_obw"], out["should_generate_cave"] = obw, sg
    mi = np.zeros((20, 4), np.float32); L.mmo_tables_material_infos(_p(mi)); out["material_infos"] = mi
    bm = np.zeros((24, 20), np.float32); L.mmo_tables_biome_material_weights(_p(bm)); out["biome_material_weights"] = bm
    br = np.zeros((24, 6), np.uint8); cr = np.zeros((5, 4), np.uint8); gr = np.zeros(24, np.uint8)
    L.mmo_tables_rules(_p(br), _p(cr), _p(gr)); out["biome_rules"], out["cave_rules"], out["grass_blocks"] = br, cr, gr
    np.savez_compressed(os.path.join(HERE, "oracle_kat.npz"), **out)
    print("oracle_kat.npz", {k: v.shape for k, v in out.items()})


def make_stages(o):
    coords = BIOME_CHUNKS + MIXED_CHUNKS
    pos = o.positions(coords)
    hf, bw = o.heightfields(pos)
    g = o.gather_heightfields(pos, hf)
    layers = o.fix_backward(o.layers(pos, g, bw))
    cave = o.caves(pos, hf, bw)
    blocks = o.fill(pos, hf, bw, layers, cave)
    out = dict(coords=np.array(coords, np.int32))
    out["sha_hf"] = np.array([sha(a) for a in hf]); out["sha_bw"] = np.array([sha(a) for a in bw])
    out["sha_gathered"] = np.array([sha(a) for a in g]); out["sha_layers"] = np.array([sha(a) for a in layers])
    out["sha_cave"] = np.array([sha(a) for a in cave]); out["sha_blocks"] = np.array([sha(a) for a in blocks])
    out["hf"] = hf    # all heightfields (1 KiB each): the 1e-5 tolerance check needs values, not hashes
    for c in FULL_CHUNKS:
        i = coords.index(c)
        tag = f"{c[0]}_{c[1]}"
        out[f"bw_{tag}"], out[f"layers_{tag}"], out[f"cave_{tag}"], out[f"blocks_{tag}"] = bw[i], layers[i], cave[i], blocks[i]
    ub = o.ub_counters()
    out["ub_counters"] = np.array([ub["no_layer_found"], ub["cave_layer_overflow"], ub["decorator_out_of_range"]], np.int64)
    np.savez_compressed(os.path.join(HERE, "stages.npz"), **out)
    print("stages.npz", len(coords), "chunks", ub)


def make_block_data():
    """Per-block render data and face directions straight from the reference's own block.cpp / enums.hpp (oracle/_ref/libblockprobe.so)."""
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblockprobe.so"))
    n = ref.ref_num_blocks()
    data = np.zeros((n, 13), np.int32); dirs = np.zeros((6, 3), np.int32)
    ref.ref_block_data(data.ctypes.data_as(ctypes.c_void_p)); ref.ref_dir_vecs(dirs.ctypes.data_as(ctypes.c_void_p))
    abi = np.zeros(256, np.int32)
    na = ref.ref_abi_layout(abi.ctypes.data_as(ctypes.c_void_p))
    np.savez_compressed(os.path.join(HERE, "block_data.npz"), block_data=data, dir_vecs=dirs, abi_layout=abi[:na])


if __name__ == "__main__":
    oracle = Oracle()
    make_block_data()
    make_glm_probe()
    make_oracle_kat(oracle)
    make_stages(oracle)
