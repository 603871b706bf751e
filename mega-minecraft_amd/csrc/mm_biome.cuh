// mmgen device "content" layer: rule tables, surface-biome noise / weights / heights, cave-biome selection and the
// per-voxel block pre/post-process rules.  Behavioural spec: src/terrain/biomeFuncs.hpp (tables :725-1256, noise :109-220,
// heights :224-383, block rules :385-707).  Tables are compile-time constants (no upload step: mmgen_init only validates
// the device), so every translation unit folds them into immediates where the index is static.
#pragma once
#include "mm_noise.cuh"
#include "../../include/mmgen_types.h"

namespace mm {

// ---------------------------------------------------------------------------------------------------------
// tables
// ---------------------------------------------------------------------------------------------------------
enum : uint8_t { RI = 0, RP = 1, RN = 2 };   // ignore / multiply by n / multiply by (1 - n)

// biome x {ocean, beach, rocky, magic, temperature, moisture}        (biomeFuncs.hpp:735-762)
__device__ constexpr uint8_t kBiomeRules[MMGEN_NUM_BIOMES][6] = {
    {RP, RN, RP, RP, RI, RI}, {RP, RN, RP, RN, RI, RI}, {RP, RN, RN, RI, RP, RI}, {RP, RN, RN, RP, RN, RI}, {RP, RN, RN, RN, RN, RI},
    {RP, RP, RP, RI, RI, RI}, {RP, RP, RN, RI, RP, RI}, {RP, RP, RN, RI, RN, RI},
    {RN, RI, RP, RP, RP, RP}, {RN, RI, RP, RP, RP, RN}, {RN, RI, RP, RP, RN, RP}, {RN, RI, RP, RP, RN, RN},
    {RN, RI, RP, RN, RP, RP}, {RN, RI, RP, RN, RP, RN}, {RN, RI, RP, RN, RN, RP}, {RN, RI, RP, RN, RN, RN},
    {RN, RI, RN, RP, RP, RP}, {RN, RI, RN, RP, RP, RN}, {RN, RI, RN, RP, RN, RP}, {RN, RI, RN, RP, RN, RN},
    {RN, RI, RN, RN, RP, RP}, {RN, RI, RN, RN, RP, RN}, {RN, RI, RN, RN, RN, RP}, {RN, RI, RN, RN, RN, RN}};

// grass block per biome (biomeFuncs.hpp:786-801; default DIRT, biome.hpp:60-63)
__device__ constexpr uint8_t kGrassBlock[MMGEN_NUM_BIOMES] = {
    MMB_DIRT, MMB_DIRT, MMB_DIRT, MMB_DIRT, MMB_DIRT,
    MMB_DIRT, MMB_JUNGLE_GRASS_BLOCK, MMB_DIRT,
    MMB_SAVANNA_GRASS_BLOCK, MMB_DIRT, MMB_SNOWY_GRASS_BLOCK, MMB_GRASS_BLOCK, MMB_JUNGLE_GRASS_BLOCK, MMB_DIRT, MMB_GRASS_BLOCK, MMB_GRASS_BLOCK,
    MMB_JUNGLE_GRASS_BLOCK, MMB_DIRT, MMB_MYCELIUM, MMB_DIRT, MMB_JUNGLE_GRASS_BLOCK, MMB_DIRT, MMB_GRASS_BLOCK, MMB_GRASS_BLOCK};

// material -> block, thickness, (noise amplitude | tan(angle of repose)), (noise scale | max slope)   (biomeFuncs.hpp:813-847)
// tan(AoR) values: correctly rounded tan of the fp32 radian value (the reference evaluates tanf on the host).
__device__ constexpr uint8_t kMaterialBlock[MMGEN_NUM_MATERIALS] = {
    MMB_BLACKSTONE, MMB_DEEPSLATE, MMB_SLATE, MMB_STONE, MMB_TUFF, MMB_CALCITE, MMB_GRANITE, MMB_TERRACOTTA, MMB_MARBLE, MMB_ANDESITE,
    MMB_RED_SANDSTONE, MMB_SANDSTONE, MMB_GRAVEL, MMB_CLAY, MMB_MUD, MMB_DIRT, MMB_RED_SAND, MMB_SAND, MMB_SMOOTH_SAND, MMB_SNOW};
__device__ constexpr float kMaterialThickness[MMGEN_NUM_MATERIALS] = {
    32.f, 66.f, 6.f, 40.f, 24.f, 20.f, 18.f, 32.f, 28.f, 24.f, 3.0f, 3.5f, 2.5f, 2.7f, 2.3f, 4.2f, 3.5f, 3.8f, 4.5f, 2.5f};
__device__ constexpr float kMaterialAmpOrTan[MMGEN_NUM_MATERIALS] = {
    32.f, 20.f, 24.f, 30.f, 42.f, 30.f, 36.f, 16.f, 56.f, 48.f, 2.0f, 1.5f,
    0x1.6d9b1ap+0f, 0x1.ad9e76p-1f, 0x1p+0f, 0x1.ad9e76p-1f, 0x1.279a74p-1f, 0x1.66819ap-1f, 0x1.127f34p+1f, 0x1p+0f};
__device__ constexpr float kMaterialScaleOrMaxSlope[MMGEN_NUM_MATERIALS] = {
    0.0030f, 0.0045f, 0.0062f, 0.0050f, 0.0060f, 0.0040f, 0.0034f, 0.0020f, 0.0050f, 0.0030f, 0.0035f, 0.0025f,
    1.8f, 1.8f, 1.6f, 1.2f, 1.5f, 1.4f, 4.0f, 1.5f};

// biome x material weight (biomeFuncs.hpp:856-957): base 1 for the 9 always-on stratified materials + DIRT, 0 for the rest, then overrides
struct MatWeights { float w[MMGEN_NUM_BIOMES][MMGEN_NUM_MATERIALS]; };
constexpr MatWeights make_material_weights()
{
    MatWeights t{};
    for (int b = 0; b < MMGEN_NUM_BIOMES; ++b) {
        for (int m = 0; m < MMGEN_NUM_MATERIALS; ++m) t.w[b][m] = 1.f;
        t.w[b][MMM_TERRACOTTA] = 0.f; t.w[b][MMM_RED_SANDSTONE] = 0.f; t.w[b][MMM_SANDSTONE] = 0.f; t.w[b][MMM_GRAVEL] = 0.f;
        t.w[b][MMM_CLAY] = 0.f; t.w[b][MMM_MUD] = 0.f; t.w[b][MMM_RED_SAND] = 0.f; t.w[b][MMM_SAND] = 0.f;
        t.w[b][MMM_SMOOTH_SAND] = 0.f; t.w[b][MMM_SNOW] = 0.f;
    }
    t.w[MMBIO_CORAL_REEF][MMM_DIRT] = 0.0f; t.w[MMBIO_CORAL_REEF][MMM_SAND] = 0.7f; t.w[MMBIO_CORAL_REEF][MMM_SMOOTH_SAND] = 0.8f;
    t.w[MMBIO_ARCHIPELAGO][MMM_GRAVEL] = 0.3f; t.w[MMBIO_ARCHIPELAGO][MMM_DIRT] = 0.0f; t.w[MMBIO_ARCHIPELAGO][MMM_SAND] = 0.8f;
    t.w[MMBIO_WARM_OCEAN][MMM_DIRT] = 0.0f; t.w[MMBIO_WARM_OCEAN][MMM_SAND] = 0.7f;
    t.w[MMBIO_ICEBERGS][MMM_GRAVEL] = 0.5f; t.w[MMBIO_ICEBERGS][MMM_DIRT] = 0.0f;
    t.w[MMBIO_COOL_OCEAN][MMM_GRAVEL] = 0.5f; t.w[MMBIO_COOL_OCEAN][MMM_DIRT] = 0.0f;
    t.w[MMBIO_ROCKY_BEACH][MMM_DIRT] = 0.0f; t.w[MMBIO_ROCKY_BEACH][MMM_GRAVEL] = 1.0f;
    t.w[MMBIO_TROPICAL_BEACH][MMM_DIRT] = 0.0f; t.w[MMBIO_TROPICAL_BEACH][MMM_SMOOTH_SAND] = 1.0f;
    t.w[MMBIO_BEACH][MMM_DIRT] = 0.0f; t.w[MMBIO_BEACH][MMM_SAND] = 1.0f;
    t.w[MMBIO_SAVANNA][MMM_STONE] = 0.6f; t.w[MMBIO_SAVANNA][MMM_TUFF] = 0.15f; t.w[MMBIO_SAVANNA][MMM_CALCITE] = 0.0f;
    t.w[MMBIO_SAVANNA][MMM_GRANITE] = 0.2f; t.w[MMBIO_SAVANNA][MMM_TERRACOTTA] = 3.2f; t.w[MMBIO_SAVANNA][MMM_MARBLE] = 0.0f;
    t.w[MMBIO_MESA][MMM_CLAY] = 0.8f; t.w[MMBIO_MESA][MMM_DIRT] = 0.0f;
    t.w[MMBIO_FROZEN_WASTELAND][MMM_GRANITE] = 0.0f; t.w[MMBIO_FROZEN_WASTELAND][MMM_DIRT] = 0.6f; t.w[MMBIO_FROZEN_WASTELAND][MMM_SNOW] = 1.1f;
    t.w[MMBIO_SHREKS_SWAMP][MMM_CLAY] = 1.7f; t.w[MMBIO_SHREKS_SWAMP][MMM_MUD] = 2.2f; t.w[MMBIO_SHREKS_SWAMP][MMM_DIRT] = 0.6f;
    t.w[MMBIO_SPARSE_DESERT][MMM_MARBLE] = 2.0f; t.w[MMBIO_SPARSE_DESERT][MMM_ANDESITE] = 0.5f; t.w[MMBIO_SPARSE_DESERT][MMM_DIRT] = 0.0f;
    t.w[MMBIO_SPARSE_DESERT][MMM_SMOOTH_SAND] = 1.4f;
    t.w[MMBIO_TIANZI_MOUNTAINS][MMM_SANDSTONE] = 1.0f;
    t.w[MMBIO_JUNGLE][MMM_CLAY] = 1.0f; t.w[MMBIO_JUNGLE][MMM_MUD] = 1.0f; t.w[MMBIO_JUNGLE][MMM_DIRT] = 0.5f;
    t.w[MMBIO_RED_DESERT][MMM_RED_SANDSTONE] = 1.0f; t.w[MMBIO_RED_DESERT][MMM_DIRT] = 0.0f; t.w[MMBIO_RED_DESERT][MMM_RED_SAND] = 1.0f;
    t.w[MMBIO_PURPLE_MUSHROOMS][MMM_GRAVEL] = 0.4f;
    t.w[MMBIO_CRYSTALS][MMM_CALCITE] = 0.3f; t.w[MMBIO_CRYSTALS][MMM_GRAVEL] = 0.15f; t.w[MMBIO_CRYSTALS][MMM_CLAY] = 0.2f;
    t.w[MMBIO_CRYSTALS][MMM_DIRT] = 0.0f;
    t.w[MMBIO_OASIS][MMM_SANDSTONE] = 1.0f; t.w[MMBIO_OASIS][MMM_CLAY] = 0.4f; t.w[MMBIO_OASIS][MMM_DIRT] = 0.6f; t.w[MMBIO_OASIS][MMM_SAND] = 0.4f;
    t.w[MMBIO_DESERT][MMM_SANDSTONE] = 1.0f; t.w[MMBIO_DESERT][MMM_DIRT] = 0.0f; t.w[MMBIO_DESERT][MMM_SAND] = 1.0f;
    t.w[MMBIO_MOUNTAINS][MMM_GRAVEL] = 1.0f;
    return t;
}
__device__ constexpr MatWeights kMatWeights = make_material_weights();

// 8-neighbour offsets N, NE, E, SE, S, SW, W, NW; odd = diagonal (util/enums.hpp:29-38)
__device__ constexpr int kDirX[8] = {0, 1, 1, 1, 0, -1, -1, -1};
__device__ constexpr int kDirZ[8] = {1, 1, 0, -1, -1, -1, 0, 1};

#define MM_SQRT_2 1.41421356237309504880168872420f

// ---------------------------------------------------------------------------------------------------------
// surface biome noise (biomeFuncs.hpp:109-128) and weights (:158-185)
// ---------------------------------------------------------------------------------------------------------
struct BiomeNoise { float n[6]; };   // ocean, beach, rocky, magic, temperature, moisture

MM_DEV float single_biome_noise(float px, float py, float scale, float ox, float oy, float thr)
{
    return smoothstep(-thr, thr, simplex2(px * scale + ox, py * scale + oy));
}

MM_DEV BiomeNoise biome_noise(float wx, float wz)
{
    const f2 warp = fbm2from2<3>(wx * 0.0150f, wz * 0.0150f);
    const float px = (wx + warp.x * 20.f) * 0.32f;
    const float py = (wz + warp.y * 20.f) * 0.32f;
    BiomeNoise b;
    const float oceanNoise = simplex2(px * 0.0007f + 2853.49f, py * 0.0007f + -9481.42f);
    b.n[0] = smoothstep(0.01f, -0.02f, oceanNoise);
    b.n[1] = smoothstep(-0.15f, -0.05f, oceanNoise);
    b.n[2] = single_biome_noise(px, py, 0.0015f, -8102.35f, -7620.23f, 0.08f);
    b.n[3] = single_biome_noise(px, py, 0.0030f, 5612.35f, 9182.49f, 0.07f);
    b.n[4] = single_biome_noise(px, py, 0.0012f, -4021.34f, -8720.12f, 0.06f);
    b.n[5] = single_biome_noise(px, py, 0.0050f, 1835.32f, 3019.39f, 0.12f);
    return b;
}

MM_DEV float biome_weight(int biome, const BiomeNoise& bn)
{
    float w = 1.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint8_t r = kBiomeRules[biome][k];
        if (r == RP) w *= bn.n[k];
        else if (r == RN) w *= 1.f - bn.n[k];
    }
    return w;
}

// getRandomBiome (biomeFuncs.hpp:39-53): w[stride * i]
MM_DEV int random_biome(const float* w, int stride, float rand)
{
    for (int i = 0; i < MMGEN_NUM_BIOMES; ++i) {
        rand -= w[stride * i];
        if (rand <= 0.f) return i;
    }
    return MMBIO_PLAINS;
}

// ---------------------------------------------------------------------------------------------------------
// per-biome height functions (biomeFuncs.hpp:224-383)
// ---------------------------------------------------------------------------------------------------------
MM_DEV float biome_height(int biome, float x, float z)
{
    switch (biome) {
    case MMBIO_CORAL_REEF: return 107.f + 16.f * fbm2<5>(x * 0.0065f, z * 0.0065f);
    case MMBIO_ARCHIPELAGO: {
        float island = (fbm2<4>(x * 0.0055f, z * 0.0055f) + 1.f) * 0.5f;
        island = powf_(island, 2.4f);
        island = smoothstep(1.f, 0.f, island);
        const float islandHeight = 22.f * island;
        const float base = 107.f + 24.f * fbm2<5>(x * 0.0060f, z * 0.0060f);
        return base + islandHeight;
    }
    case MMBIO_WARM_OCEAN: return 93.f + 18.f * fbm2<5>(x * 0.0055f, z * 0.0055f);
    case MMBIO_ICEBERGS: return 66.f + 18.f * fbm2<5>(x * 0.0060f, z * 0.0060f);
    case MMBIO_COOL_OCEAN: return 80.f + 22.f * fbm2<5>(x * 0.0065f, z * 0.0065f);
    case MMBIO_ROCKY_BEACH: return 134.f + 8.f * fbm2<5>(x * 0.0070f, z * 0.0070f);
    case MMBIO_TROPICAL_BEACH: return 129.5f + 6.f * fbm2<5>(x * 0.0045f, z * 0.0045f);
    case MMBIO_BEACH: return 132.f + 5.f * fbm2<5>(x * 0.0055f, z * 0.0055f);
    case MMBIO_SAVANNA: {
        const f2 off = fbm2from2<5>(x * 0.0040f, z * 0.0040f);
        const float nx = x + off.x * 100.f, nz = z + off.y * 100.f;
        float p1 = worley2(nx * 0.0070f, nz * 0.0070f).d1;
        p1 = smoothstep(0.30f, 0.20f, p1) * (1.f + 0.3f * simplex2(nx * 0.0100f, nz * 0.0100f));
        float p2 = worley2((nx + -3910.12f) * 0.0045f, (nz + -9012.34f) * 0.0045f).d1;
        p2 = smoothstep(0.16f, 0.08f, p2) * (1.f + 0.2f * simplex2(nx * 0.0130f, nz * 0.0130f));
        const float plateau = (p1 * 14.f) + (p2 * 9.f);
        return (136.f + 9.f * fbm2<4>(x * 0.0080f, z * 0.0080f)) + plateau;
    }
    case MMBIO_MESA: {
        const float mx = x * 0.7f, mz = z * 0.7f;
        const f2 o = fbm2from2<5>(mx * 0.0050f, mz * 0.0050f);
        const float ox = o.x * 300.f, oz = o.y * 300.f;
        const Worley2 w = worley2((mx + ox) * 0.0030f, (mz + oz) * 0.0030f);
        const float river = (w.d2 - w.d1) * 0.5f;
        float base = 122.f;
        base += 10.f * smoothstep(0.00f, 0.05f, river);
        base += (37.5f + 5.0f * fbm2<4>((mx + 0.02f * ox) * 0.0300f, (mz + 0.02f * oz) * 0.0300f)) * smoothstep(0.07f, 0.22f, river);
        return base + 6.f * simplex2(mx * 0.0250f, mz * 0.0250f);
    }
    case MMBIO_FROZEN_WASTELAND: return 136.f + 16.f * fbm2<5>(x * 0.0035f, z * 0.0035f);
    case MMBIO_REDWOOD_FOREST: return 134.f + 8.f * fbm2<5>(x * 0.0120f, z * 0.0120f);
    case MMBIO_SHREKS_SWAMP: return 130.f + 12.f * fbm2<5>(x * 0.0080f, z * 0.0080f);
    case MMBIO_SPARSE_DESERT: {
        const f2 o = simplex2from2(x * 0.0080f, z * 0.0080f);
        const float dunes = powf_(worley2((x + o.x * 20.0f) * 0.0160f, (z + o.y * 20.0f) * 0.0160f).d1, 2.f) * 18.f;
        return (132.f + 4.f * fbm2<4>(x * 0.0070f, z * 0.0070f)) + dunes;
    }
    case MMBIO_LUSH_BIRCH_FOREST: {
        const float hills = (simplex2(x * 0.0012f, z * 0.0012f) + 0.8f) * 20.f;
        return (135.f + 8.f * fbm2<5>(x * 0.0090f, z * 0.0090f)) + hills;
    }
    case MMBIO_TIANZI_MOUNTAINS: {
        const f2 o = simplex2from2(x * 0.0800f, z * 0.0800f);
        const float nx = (x + o.x * 3.0f) * 0.0150f, nz = (z + o.y * 3.0f) * 0.0150f;
        const float w1 = smoothstep(0.45f, 0.35f, worley2(nx, nz).d1) * 1.2f;
        const float w2 = smoothstep(0.45f, 0.35f, worley2(nx * 1.4f + 4292.12f, nz * 1.4f + 9183.27f).d1) * 0.6f;
        float mountains = w1 + w2;
        mountains *= 54.f + 7.f * fbm2<3>(nx * 1.7f, nz * 1.7f);
        const float hills = 16.f * simplex2(x * 0.0150f, z * 0.0150f);
        return ((128.f + hills) + 9.f * fbm2<3>(x * 0.0070f, z * 0.0070f)) + mountains;
    }
    case MMBIO_JUNGLE: {
        const float hills = (simplex2(x * 0.0030f, z * 0.0030f) + 0.5f) * 25.f;
        return (139.f + 8.f * fbm2<5>(x * 0.0120f, z * 0.0120f)) + hills;
    }
    case MMBIO_RED_DESERT: return 137.f + 13.f * fbm2<5>(x * 0.0075f, z * 0.0075f);
    case MMBIO_PURPLE_MUSHROOMS: return 136.f + 9.f * fbm2<5>(x * 0.0140f, z * 0.0140f);
    case MMBIO_CRYSTALS: {
        const float towersBase = simplex2(x * 0.0030f, z * 0.0030f);
        const Worley2 w = worley2(x * 0.0700f, z * 0.0700f);
        const f3 color = rand3from2(w.closest.x, w.closest.y);
        float tw = (w.d2 - w.d1) * 0.5f;
        tw = smoothstep(0.10f, 0.15f, tw);
        tw *= 0.4f + 1.2f * color.x;
        float towers = (60.f * tw) * smoothstep(0.70f, 0.74f, towersBase);
        towers += 18.f * smoothstep(0.35f, 0.8f, towersBase);
        const float base = 137.f + 8.f * fbm2<5>(x * 0.0200f, z * 0.0200f);
        return base + towers;
    }
    case MMBIO_OASIS: return 132.f + 9.f * fbm2<5>(x * 0.0120f, z * 0.0120f);
    case MMBIO_DESERT: return 136.f + 6.f * fbm2<5>(x * 0.0110f, z * 0.0110f);
    case MMBIO_PLAINS: return 144.f + 8.f * fbm2<5>(x * 0.0080f, z * 0.0080f);
    case MMBIO_MOUNTAINS: {
        float n = powf_(__builtin_fabsf(fbm2<5>(x * 0.0035f, z * 0.0035f)) + 0.05f, 2.f);
        n += ((fbm2<5>(x * 0.0050f, z * 0.0050f) - 0.5f) * 2.f) * 0.05f;
        return (165.f + (140.f * (n - 0.15f))) + (n * (20.f * fbm2<5>(x * 0.0350f, z * 0.0350f)));
    }
    }
    return (float)MMGEN_SEA_LEVEL;
}

// One column of kernGenerateHeightfield (chunk.cu:162-184): weights for all 24 biomes, height = sum of w * h over w > 0.
MM_DEV float column_height(int wxi, int wzi, float* w24 /* nullable */)
{
    const float wx = (float)wxi, wz = (float)wzi;
    const BiomeNoise bn = biome_noise(wx, wz);
    float height = 0.f;
    for (int b = 0; b < MMGEN_NUM_BIOMES; ++b) {
        const float w = biome_weight(b, bn);
        if (w > 0.f) height += w * biome_height(b, wx, wz);
        if (w24) w24[b] = w;
    }
    return height;
}

// ---------------------------------------------------------------------------------------------------------
// cave biome (biomeFuncs.hpp:130-220).  Weights by rule table :769-776:
//   NONE = none; CRYSTAL = (1-none) shallow rocky; LUSH = (1-none) shallow (1-rocky); WARPED = (1-shallow) warped;
//   AMBER = (1-shallow)(1-warped).  The noises are evaluated lazily in the order the cumulative test consumes them —
//   the values used are identical to evaluating all four up front, the unused ones are simply never computed.
// ---------------------------------------------------------------------------------------------------------
// Exact pruning (the values that ARE computed are the reference's; what is skipped provably cannot change the result): the depth bands
// are smoothsteps of the warped height py between band edges that are fbm2<3> offsets around fixed levels, and |fbm2<3>| <= A = 0.875 *
// MM_SIMPLEX2_BOUND (mm_noise.cuh: the adversarial supremum of simplex2 plus the rounding slack, valid inside the pruning domain).
// So once py (3 of the 9 warp evaluations) is known:
//   py >= top - 19 + 23 A + slack  =>  none == 1 exactly  =>  the draw minus 1 is <= 0  =>  NONE         (skips 6 simplex3 + 6 simplex2)
//   py <= top - 82 - 25 A - slack  =>  none == 0 and shallow == 0 exactly  =>  only the two deep biomes remain; a caller that has no
//                                     use for them (wantDeep = false) gets NONE                          (skips 6 simplex3 + 12 simplex2)
#ifndef MM_CAVE_BIOME_PRUNE
#define MM_CAVE_BIOME_PRUNE 1
#endif
// PRUNE = the column lies inside the pruning domain (prune_domain(wx, wz), mm_noise.cuh): a compile-time switch, so that the pruned path
// carries no trace of the plain one (k_fill sits exactly at its register budget); kernels pick the instantiation per workgroup or per lane.
// The part of getCaveBiome after the position warp: rocky, one depth band, the draw (biomeFuncs.hpp:135-220).  ox, oz = the warp's x and z
// components, py = the warped height.
template <bool PRUNE>
MM_DEV int cave_biome_draw(int wx, int wy, int wz, float maxHeight, int seed, bool wantDeep, bool crystalOnly, const float ox, const float py, const float oz)
{
    const float fx = (float)wx, fz = (float)wz;
    const float top = (float)MMGEN_SEA_LEVEL + 0.15f * (maxHeight - (float)MMGEN_SEA_LEVEL);
    constexpr float kA = 0.875f * MM_SIMPLEX2_BOUND;
    constexpr bool prune = PRUNE && MM_CAVE_BIOME_PRUNE;
    const float px = (fx + ox * 30.f) * 1.f, pz = (fz + oz * 30.f) * 1.f;
    const float qx = px * 0.2000f, qz = pz * 0.2000f;

    MinStd rng = rng4(wx, wy, wz, seed);
    float rand = rng.u01();

    // crystalOnly: the caller only needs to know whether the result is CRYSTAL_CAVES (k_fill: a voxel further than 7 blocks from every cave
    // surface - LUSH_CAVES only converts within 1.5 + 4.5 simplex3 <= 7.67 blocks of one - that is not the top of a cave floor).  CRYSTAL's
    // weight is (1 - none) shallow rocky, and rocky = smoothstep(-0.05, 0.05, simplex3(p * 0.0022)) is exactly 0 over half the world in
    // patches hundreds of blocks wide: there the draw cannot select CRYSTAL (it subtracts +0) and neither depth band is evaluated.  rocky
    // is therefore evaluated BEFORE the bands (same value, same place in the draw).
    float rocky = 0.f;
    const bool rockyMatters = !prune || !(py <= ((((top - 72.f) - 18.f * kA) - 10.f) - 7.f * kA) - 0.05f);      // shallow can be non-zero
    if (rockyMatters) {
        rocky = smoothstep(-0.05f, 0.05f, simplex3(px * 0.0022f + -9193.23f, py * 0.0022f + -6813.39f, pz * 0.0022f + (float)-2171.23));
        if (crystalOnly && rocky == 0.f) return MMCB_NONE;
    }

    // The two depth bands: none = smoothstep(n2sEnd, n2sStart, py), shallow = smoothstep(s2dEnd, s2dStart, py), each edge pair built from
    // two fbm2<3>.  A smoothstep is exactly 0 at or below its lower edge and exactly 1 at or above its upper edge, and the edges are
    // confined by the same bound as above, so py decides which band can be strictly inside its transition:
    //   py <= (top - 19 - 23 A) - 5 - 3 A - slack  =>  none == 0 exactly (rand - 0 == rand)          py >= (top - 72) + 18 A + slack  =>  shallow == 1 exactly
    // The first limit lies 7 blocks above the second, so at most ONE band needs its 6 simplex2 (the reference evaluates 12); lanes that
    // need the upper band and lanes that need the lower one run the same instructions on per-lane constants.
    static_assert((((-19.f - 23.f * kA) - 5.f) - 3.f * kA) - 0.05f > ((-72.f + 18.f * kA) + 0.05f) + 0.5f, "the two transition bands must not overlap");
    float none, shallow;
    if constexpr (prune) {
        const bool noneZero = py <= ((((top - 19.f) - 23.f * kA) - 5.f) - 3.f * kA) - 0.05f;
        const bool shallowOne = py >= ((top - 72.f) + 18.f * kA) + 0.05f;
        const bool shallowZero = py <= ((((top - 72.f) - 18.f * kA) - 10.f) - 7.f * kA) - 0.05f;
        const int band = !noneZero ? 0 : ((shallowOne || shallowZero) ? -1 : 1);
        float bandVal = 0.f;
        if (band >= 0) {
            const bool b0 = band == 0;
            const float f1 = b0 ? fbm2<3, true>(qx, qz) : fbm2<3, true>(qx + -4921.34f, qz + 8402.13f);
            const float edge1 = (b0 ? (top - 19.f) : (top - 72.f)) + (b0 ? 23.f : 18.f) * f1;
            const float f2 = fbm2<3, true>(qx + (b0 ? 3821.34f : 9411.32f), qz + (b0 ? 4920.32f : -3921.34f));
            const float edge0 = (edge1 - (b0 ? 5.f : 10.f)) + (b0 ? 3.f : 7.f) * f2;
            bandVal = smoothstep(edge0, edge1, py);
        }
        none = band == 0 ? bandVal : 0.f;
        rand -= none;                                   // NONE
        if (rand <= 0.f) return MMCB_NONE;
        shallow = band == 1 ? bandVal : (shallowZero ? 0.f : 1.f);
    } else {
        const float n2sStart = (top - 19.f) + 23.f * fbm2<3, true>(qx, qz);
        const float n2sEnd = (n2sStart - 5.f) + 3.f * fbm2<3, true>(qx + 3821.34f, qz + 4920.32f);
        none = smoothstep(n2sEnd, n2sStart, py);
        rand -= none;                                   // NONE
        if (rand <= 0.f) return MMCB_NONE;
        const float s2dStart = (top - 72.f) + 18.f * fbm2<3, true>(qx + -4921.34f, qz + 8402.13f);
        const float s2dEnd = (s2dStart - 10.f) + 7.f * fbm2<3, true>(qx + 9411.32f, qz + -3921.34f);
        shallow = smoothstep(s2dEnd, s2dStart, py);
    }

    // rand > 0 here.  A weight that is exactly 0 (rocky, warped are in [0, 1]) subtracts +0 and cannot select its biome, so
    // the simplex3 behind warped is only evaluated when its factor is non-zero (shallowW != 0 implies rockyMatters: rocky was evaluated).
    const float shallowW = (1.f - none) * shallow;
    if (shallowW != 0.f) {
        rand -= shallowW * rocky;                   // CRYSTAL_CAVES
        if (rand <= 0.f) return MMCB_CRYSTAL_CAVES;
        rand -= shallowW * (1.f - rocky);           // LUSH_CAVES
        if (rand <= 0.f) return crystalOnly ? MMCB_NONE : MMCB_LUSH_CAVES;
    }
    const float deepW = 1.f - shallow;
    // wantDeep = false: the caller has no use for WARPED_FOREST / AMBER_FOREST (k_fill: those two only re-skin the top DEEPSLATE /
    // BLACKSTONE block of a cave floor, biomeFuncs.hpp), so their simplex3 is not evaluated and NONE stands in for both
    if (deepW != 0.f && wantDeep) {
        const float warped = smoothstep(-0.05f, 0.05f, simplex3(px * 0.0030f + 5821.32f, py * 0.0030f + 4920.12f, pz * 0.0030f + 7931.59f));
        rand -= deepW * warped;                     // WARPED_FOREST
        if (rand <= 0.f) return MMCB_WARPED_FOREST;
        rand -= deepW * (1.f - warped);             // AMBER_FOREST
        if (rand <= 0.f) return MMCB_AMBER_FOREST;
    }
    return MMCB_NONE;
}

template <bool PRUNE>
MM_DEV int cave_biome_t(int wx, int wy, int wz, float maxHeight, int seed, bool wantDeep, bool crystalOnly)
{
    const float fx = (float)wx, fy = (float)wy, fz = (float)wz;
    const float top = (float)MMGEN_SEA_LEVEL + 0.15f * (maxHeight - (float)MMGEN_SEA_LEVEL);
    const float sx = fx * 0.0470f, sy = fy * 0.0470f, sz = fz * 0.0470f;
    constexpr float kA = 0.875f * MM_SIMPLEX2_BOUND;            // >= |fbm2<3>| inside the pruning domain (mm_noise.cuh)
    constexpr bool prune = PRUNE && MM_CAVE_BIOME_PRUNE;
    // fbm3From3<3> component by component (rng.hpp:180-186), y first; one rolled loop = one inlined simplex body (code size)
    float o[3], py = 0.f;
#pragma unroll 1
    for (int it = 0; it < 3; ++it) {
        const int k = it == 0 ? 1 : (it == 1 ? 0 : 2);
        const float ax = k == 0 ? 0.f : (k == 1 ? 5923.45f : 1765.68f), ay = k == 0 ? 0.f : (k == 1 ? 4129.42f : 4704.36f),
                    az = k == 0 ? 0.f : (k == 1 ? 5790.48f : 5692.12f);
        o[k] = k == 0 ? fbm3<3>(sx, sy, sz) : fbm3<3>(sx + ax, sy + ay, sz + az);
        if (it == 0) {
            py = (fy + o[1] * 24.f) * 1.f;
            if constexpr (prune) {
                if (py >= (top - 19.f) + 23.f * kA + 0.05f) return MMCB_NONE;
                if (!wantDeep && py <= ((top - 72.f) - 18.f * kA - 10.f) - 7.f * kA - 0.05f) return MMCB_NONE;
            }
        }
    }
    return cave_biome_draw<PRUNE>(wx, wy, wz, maxHeight, seed, wantDeep, crystalOnly, o[0], py, o[2]);
}

// Split at the warped height py: cave_biome_py evaluates the y component of the warp (3 of the 9 simplex3) and decides what py alone
// decides; cave_biome_rest takes py and does the rest.  cave_biome_t = both in one go (k_fill); k_cave_biomes runs the two halves as
// separate dense phases (every lane of the first does the same three evaluations; only the survivors, compacted again, pay for the rest).
template <bool PRUNE>
MM_DEV bool cave_biome_py(int wx, int wy, int wz, float maxHeight, bool wantDeep, float& py)      // true: the result is NONE
{
    const float fx = (float)wx, fy = (float)wy, fz = (float)wz;
    const float top = (float)MMGEN_SEA_LEVEL + 0.15f * (maxHeight - (float)MMGEN_SEA_LEVEL);
    const float sx = fx * 0.0470f, sy = fy * 0.0470f, sz = fz * 0.0470f;
    constexpr float kA = 0.875f * MM_SIMPLEX2_BOUND;            // >= |fbm2<3>| inside the pruning domain (mm_noise.cuh)
    constexpr bool prune = PRUNE && MM_CAVE_BIOME_PRUNE;
    // fbm3From3<3> component by component (rng.hpp:180-186), y first
    const float oy = fbm3<3>(sx + 5923.45f, sy + 4129.42f, sz + 5790.48f);
    py = (fy + oy * 24.f) * 1.f;
    if constexpr (prune) {
        if (py >= (top - 19.f) + 23.f * kA + 0.05f) return true;
        if (!wantDeep && py <= ((top - 72.f) - 18.f * kA - 10.f) - 7.f * kA - 0.05f) return true;
    }
    return false;
}

template <bool PRUNE>
MM_DEV int cave_biome_rest(int wx, int wy, int wz, float maxHeight, int seed, bool wantDeep, bool crystalOnly, const float py)
{
    const float fx = (float)wx, fy = (float)wy, fz = (float)wz;
    const float sx = fx * 0.0470f, sy = fy * 0.0470f, sz = fz * 0.0470f;
    // the x and z components of the warp; one rolled loop = one inlined simplex body (code size)
    float o[3];
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
        const int k = it == 0 ? 0 : 2;
        const float ax = k == 0 ? 0.f : 1765.68f, ay = k == 0 ? 0.f : 4704.36f, az = k == 0 ? 0.f : 5692.12f;
        o[k] = k == 0 ? fbm3<3>(sx, sy, sz) : fbm3<3>(sx + ax, sy + ay, sz + az);
    }
    return cave_biome_draw<PRUNE>(wx, wy, wz, maxHeight, seed, wantDeep, crystalOnly, o[0], py, o[2]);
}

MM_DEV int cave_biome(int wx, int wy, int wz, float maxHeight, int seed)
{
    return prune_domain(wx, wz) ? cave_biome_t<true>(wx, wy, wz, maxHeight, seed, true, false) : cave_biome_t<false>(wx, wy, wz, maxHeight, seed, true, false);
}

// ---------------------------------------------------------------------------------------------------------
// block rules (biomeFuncs.hpp:385-707).  Blocks are uint8 ids.
// ---------------------------------------------------------------------------------------------------------
MM_DEV bool biome_block_pre(uint8_t& block, int biome, int wx, int wy, int wz, float height)
{
    if (biome == MMBIO_CRYSTALS && height > 176.f) {
        const float quartzStart = 140.f + 15.f * fbm2<3>((float)wx * 0.0080f, (float)wz * 0.0080f);
        if ((float)wy > quartzStart) { block = MMB_QUARTZ; return true; }
    }
    return false;
}

MM_DEV void biome_block_post(uint8_t& block, int biome, int wx, int wy, int wz, bool isTop)
{
    const float fx = (float)wx, fy = (float)wy, fz = (float)wz;
    switch (biome) {
    case MMBIO_ARCHIPELAGO: {
        if (wy < MMGEN_SEA_LEVEL || block == MMB_WATER) return;
        const float dirtHeight = ((float)MMGEN_SEA_LEVEL + 1.5f) + 1.7f * fbm2<3>(fx * 0.0065f, fz * 0.0065f);
        if (fy > dirtHeight) block = isTop ? MMB_GRASS_BLOCK : MMB_DIRT;
        return;
    }
    case MMBIO_TROPICAL_BEACH:
        if (isTop && block != MMB_SMOOTH_SAND && block != MMB_WATER) block = MMB_SMOOTH_SAND;
        return;
    case MMBIO_BEACH:
        if (isTop && block != MMB_SAND && block != MMB_WATER) block = MMB_SAND;
        return;
    case MMBIO_MESA: {
        if (fy < 90.f || block == MMB_WATER) return;
        const float start = 108.f + 12.f * fbm2<3>(fx * 0.0040f, fz * 0.0040f);
        if (fy < start) return;
        if (block == MMB_CLAY && fy < start + 20.f) return;
        float s = (fy + 3.f * simplex3(fx * 0.0100f, fz * 0.0100f, fy * 0.0300f)) - start;
        s = gmod(s, 32.f);
        uint8_t t;
        if (s < 5.f) t = MMB_TERRACOTTA;
        else if (s < 8.f) t = MMB_ORANGE_TERRACOTTA;
        else if (s < 12.f) t = MMB_RED_TERRACOTTA;
        else if (s < 14.f) t = MMB_WHITE_TERRACOTTA;
        else if (s < 20.f) t = MMB_TERRACOTTA;
        else if (s < 21.f) t = MMB_ORANGE_TERRACOTTA;
        else if (s < 26.f) t = MMB_YELLOW_TERRACOTTA;
        else if (s < 29.f) t = MMB_PURPLE_TERRACOTTA;
        else t = MMB_TERRACOTTA;
        block = t;
        return;
    }
    case MMBIO_FROZEN_WASTELAND:
        if (block == MMB_WATER) block = MMB_PACKED_ICE;
        return;
    case MMBIO_SHREKS_SWAMP: {
        if (fy < 100.f) return;
        if (block == MMB_DIRT || block == MMB_JUNGLE_GRASS_BLOCK) {
            const float mudEnd = ((float)MMGEN_SEA_LEVEL + 0.8f) + 1.1f * simplex2(fx * 0.0300f, fz * 0.0300f);
            if (fy < mudEnd) block = MMB_MUD;
        }
        return;
    }
    case MMBIO_TIANZI_MOUNTAINS: {
        if (fy < 90.f || block == MMB_WATER || block == MMB_DIRT || block == MMB_GRASS_BLOCK) return;
        const float start = 112.f + 16.f * fbm2<3>(fx * 0.0200f, fz * 0.0200f);
        if (fy < start) return;
        block = MMB_SMOOTH_SANDSTONE;
        return;
    }
    case MMBIO_CRYSTALS:
        if (!isTop || block == MMB_QUARTZ) return;
        if (rand1from2((float)(wx + 913213), (float)(wz + 85941)) < 0.1f) block = MMB_MYCELIUM;
        return;
    case MMBIO_MOUNTAINS: {
        if (fy < 190.f) return;
        const float snowStart = 202.f + 5.f * fbm2<3>(fx * 0.0500f, fz * 0.0500f);
        if (fy < snowStart) return;
        block = MMB_SNOW;
        return;
    }
    default: return;
    }
}

// true when caveBiomeBlockPostProcess could change `block` for some cave biome (all its rules start from one of
// these blocks, biomeFuncs.hpp:606,647,673-683,693-703); otherwise the cave biome need not be evaluated at all.
MM_DEV bool cave_post_can_apply(uint8_t block) { return block == MMB_STONE || block == MMB_DEEPSLATE || block == MMB_BLACKSTONE; }

// caveBiomeBlockPostProcess (biomeFuncs.hpp:596-707) in three stages, so that a kernel can run each stage DENSELY over the voxels that
// reach it (k_fill): the rules of CRYSTAL_CAVES and LUSH_CAVES both start with one simplex3, and only the few lush voxels within a
// noise-dependent distance of a cave floor / ceiling go on to the expensive clay-or-moss evaluation.  `block` is one of
// STONE / DEEPSLATE / BLACKSTONE (cave_post_can_apply), the only blocks any rule touches.
MM_DEV void cave_post_noise_pos(int caveBiome, int wx, int wy, int wz, float& ax, float& ay, float& az)
{
    if (caveBiome == MMCB_CRYSTAL_CAVES) {
        ax = (float)(wx + wy) * 0.05f; ay = (float)(wz + 5819323) * 0.05f; az = ((float)(wx + wz) * 2.0f) * 0.05f;
    } else {                                                 // LUSH_CAVES
        ax = (float)wx * 0.025f; ay = (float)wy * 0.025f; az = (float)wz * 0.025f;
    }
}

// n = simplex3(cave_post_noise_pos).  Returns true when the voxel needs lush_clay_or_moss(); otherwise `block` is final.
MM_DEV bool cave_post_apply(uint8_t& block, int caveBiome, float n, int wx, int wy, int wz, int caveBottomDepth, int caveTopDepth)
{
    if (caveBiome == MMCB_CRYSTAL_CAVES) {
        if (n < -0.25f) { block = MMB_QUARTZ; return false; }
        if (block == MMB_BLACKSTONE) return false;
        const float chance = (block == MMB_STONE) ? 0.5f : 0.4f;
        const uint8_t cobble = (block == MMB_STONE) ? MMB_COBBLESTONE : MMB_COBBLED_DEEPSLATE;
        if (rand1from3((float)wx, (float)wy, (float)wz) < chance) block = cobble;
        return false;
    }
    const float threshold = 1.5f + 4.5f * n;                 // LUSH_CAVES
    const float bd = (float)caveBottomDepth, td = (float)caveTopDepth;
    return (bd >= 0.f && bd <= threshold) || (td >= 0.f && td <= threshold);
}

template <class Cells>
MM_DEV uint8_t lush_clay_or_moss(int wx, int wy, int wz, const Cells& cells)
{
    const float nx = (float)wx * 0.025f, nz = (float)wz * 0.025f;
    const float ny = (float)wy * 0.025f + 192031.9821f;
    const f3 o = fbm3from3<3>(nx * 0.4f, ny * 0.4f, nz * 0.4f);
    const float clay = worley3(nx + o.x * 2.f, ny + o.y * 2.f, nz + o.z * 2.f, cells).d1;
    return clay < 0.25f ? MMB_CLAY : MMB_MOSS;
}

MM_DEV void cave_biome_block_post(uint8_t& block, int caveBiome, int wx, int wy, int wz, int caveBottomDepth, int caveTopDepth)
{
    if (caveBiome == MMCB_NONE) return;
    const bool isTop = caveBottomDepth == 0;
    switch (caveBiome) {
    case MMCB_CRYSTAL_CAVES:
    case MMCB_LUSH_CAVES: {
        if (!cave_post_can_apply(block)) return;
        float ax, ay, az;
        cave_post_noise_pos(caveBiome, wx, wy, wz, ax, ay, az);
        if (cave_post_apply(block, caveBiome, simplex3(ax, ay, az), wx, wy, wz, caveBottomDepth, caveTopDepth))
            block = lush_clay_or_moss(wx, wy, wz, CellDirect());
        return;
    }
    case MMCB_WARPED_FOREST:
        if (!isTop) return;
        if (block == MMB_DEEPSLATE) block = MMB_WARPED_DEEPSLATE;
        else if (block == MMB_BLACKSTONE) block = MMB_WARPED_BLACKSTONE;
        return;
    case MMCB_AMBER_FOREST:
        if (!isTop) return;
        if (block == MMB_DEEPSLATE) block = MMB_AMBER_DEEPSLATE;
        else if (block == MMB_BLACKSTONE) block = MMB_AMBER_BLACKSTONE;
        return;
    default: return;
    }
}

}  // namespace mm
