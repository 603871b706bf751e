// mmgen feature stages for gfx950:
//   k_feature_placements  (F1)  per-column placement generation, stable block-wide compaction into per-chunk lists
//   k_gather_placements   (F2)  concatenation of the 49 neighbour lists in the reference's fixed offset order + height bounds
//   k_apply_features      (K6b) first-match scan of the gathered lists per voxel (the second half of kernFill)
//   k_decorators          (D1)  per-chunk decorator pass, parallel over columns by jumping the chunk's minstd stream ahead
// Behavioural spec: chunk.cu:999-1196 (placements, gather), :1438-1509 (list scan in kernFill), :1555-1601 (host bounds /
// truncation), :1634-1747 (decorators); gen tables biomeFuncs.hpp:974-1252.
//
// In the reference F1 and D1 run on the CPU between GPU stages (host round trips); here they are device kernels so that the
// whole chunk pipeline stays resident in HBM.  Ordering is part of the output (first match wins in fill): lists are emitted
// in column order (idx2d ascending = z outer, x inner, chunk.cu:1149-1155) and, within a column, in emission order.
#include <hip/hip_runtime.h>
#include "mm_features.cuh"
#include "mmgen_features.h"
#include "mmgen_prof.h"

namespace mm {

// ---------------------------------------------------------------------------------------------------------
// gen tables (biomeFuncs.hpp:974-1040, 1188-1208, 1078-1178, 1228-1252)
// ---------------------------------------------------------------------------------------------------------
struct SurfGen { uint8_t feature; uint8_t cell; uint8_t pad; uint8_t canReplace; float chance; uint8_t nTop; uint8_t topMat[2]; float topMin[2]; };
struct CaveGen { uint8_t feature; uint8_t cell; uint8_t pad; uint8_t minLayerHeight; float chance; uint8_t canReplace; uint8_t fromCeiling; uint8_t canLava; };
struct DecoGen { uint8_t block; float chance; uint8_t nUnder; uint8_t under[3]; uint8_t replace; uint8_t second; uint8_t fromCeiling; };

#define SG(f, cell, pad, chance, rep, n, m0, t0, m1, t1) {f, cell, pad, rep, chance, n, {m0, m1}, {t0, t1}}
__device__ constexpr int kSurfGenCount[MMGEN_NUM_BIOMES] = {2, 0, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 2, 0, 1, 2, 4, 2, 2, 2, 2, 2, 0, 0};
__device__ constexpr SurfGen kSurfGens[MMGEN_NUM_BIOMES][4] = {
    /* CORAL_REEF */ {SG(MMF_CORAL, 5, 0, 0.65f, 1, 2, MMM_SMOOTH_SAND, 0.3f, MMM_SAND, 0.3f), SG(MMF_KELP, 8, 0, 0.50f, 1, 2, MMM_SMOOTH_SAND, 0.3f, MMM_SAND, 0.3f)},
    /* ARCHIPELAGO */ {}, /* WARM_OCEAN */ {},
    /* ICEBERGS */ {SG(MMF_ICEBERG, 112, 6, 0.70f, 1, 0, 0, 0.f, 0, 0.f)},
    /* COOL_OCEAN */ {}, /* ROCKY_BEACH */ {},
    /* TROPICAL_BEACH */ {SG(MMF_PALM_TREE, 48, 3, 0.35f, 1, 1, MMM_SMOOTH_SAND, 0.3f, 0, 0.f)},
    /* BEACH */ {},
    /* SAVANNA */ {SG(MMF_ACACIA_TREE, 36, 4, 0.3f, 1, 1, MMM_DIRT, 0.5f, 0, 0.f)},
    /* MESA */ {}, /* FROZEN_WASTELAND */ {},
    /* REDWOOD_FOREST */ {SG(MMF_REDWOOD_TREE, 16, 2, 0.70f, 1, 1, MMM_DIRT, 0.5f, 0, 0.f)},
    /* SHREKS_SWAMP */ {SG(MMF_CYPRESS_TREE, 18, 3, 0.6f, 1, 2, MMM_DIRT, 0.5f, MMM_MUD, 0.5f), SG(MMF_BIRCH_TREE, 16, 2, 0.15f, 1, 1, MMM_DIRT, 0.4f, 0, 0.f)},
    /* SPARSE_DESERT */ {},
    /* LUSH_BIRCH_FOREST */ {SG(MMF_BIRCH_TREE, 9, 2, 0.7f, 1, 1, MMM_DIRT, 0.5f, 0, 0.f)},
    /* TIANZI_MOUNTAINS */ {SG(MMF_PINE_TREE, 7, 1, 0.80f, 0, 0, 0, 0.f, 0, 0.f), SG(MMF_PINE_SHRUB, 6, 1, 0.80f, 0, 0, 0, 0.f, 0, 0.f)},
    /* JUNGLE */ {SG(MMF_RAFFLESIA, 54, 6, 0.50f, 1, 1, MMM_DIRT, 0.5f, 0, 0.f), SG(MMF_LARGE_JUNGLE_TREE, 28, 3, 0.70f, 1, 1, MMM_DIRT, 0.5f, 0, 0.f),
                   SG(MMF_SMALL_JUNGLE_TREE, 10, 2, 0.82f, 1, 1, MMM_DIRT, 0.5f, 0, 0.f), SG(MMF_TINY_JUNGLE_TREE, 6, 1, 0.28f, 1, 1, MMM_DIRT, 0.5f, 0, 0.f)},
    /* RED_DESERT */ {SG(MMF_PALM_TREE, 40, 3, 0.20f, 1, 1, MMM_RED_SAND, 0.3f, 0, 0.f), SG(MMF_CACTUS, 16, 2, 0.20f, 1, 1, MMM_RED_SAND, 0.5f, 0, 0.f)},
    /* PURPLE_MUSHROOMS */ {SG(MMF_MEDIUM_PURPLE_MUSHROOM, 10, 2, 0.50f, 1, 1, MMM_DIRT, 0.3f, 0, 0.f), SG(MMF_PURPLE_MUSHROOM, 11, 3, 0.45f, 1, 1, MMM_DIRT, 0.5f, 0, 0.f)},
    /* CRYSTALS */ {SG(MMF_MEDIUM_CRYSTAL, 28, 6, 0.9f, 1, 0, 0, 0.f, 0, 0.f), SG(MMF_CRYSTAL, 52, 10, 0.8f, 1, 0, 0, 0.f, 0, 0.f)},
    /* OASIS */ {SG(MMF_PALM_TREE, 24, 3, 0.35f, 1, 1, MMM_SAND, 0.3f, 0, 0.f), SG(MMF_CACTUS, 16, 2, 0.40f, 1, 1, MMM_SAND, 0.5f, 0, 0.f)},
    /* DESERT */ {SG(MMF_PALM_TREE, 64, 3, 0.30f, 1, 1, MMM_SAND, 0.3f, 0, 0.f), SG(MMF_CACTUS, 16, 2, 0.70f, 1, 1, MMM_SAND, 0.5f, 0, 0.f)},
    /* PLAINS */ {}, /* MOUNTAINS */ {}};

// caveFeature, cell, pad, minLayerHeight, chance, canReplace, fromCeiling, canLava
__device__ constexpr int kCaveGenCount[MMGEN_NUM_CAVE_BIOMES] = {0, 3, 2, 2, 2};
__device__ constexpr CaveGen kCaveGens[MMGEN_NUM_CAVE_BIOMES][3] = {
    {},
    {{MMCF_STORMLIGHT_SPHERE, 32, 4, 4, 0.80f, 1, 0, 0}, {MMCF_CEILING_STORMLIGHT_SPHERE, 32, 4, 4, 0.80f, 1, 1, 0}, {MMCF_CRYSTAL_PILLAR, 28, 5, 10, 0.60f, 0, 1, 0}},
    {{MMCF_GLOWSTONE_CLUSTER, 24, 3, 16, 0.60f, 0, 1, 0}, {MMCF_CAVE_VINE, 4, 0, 4, 0.40f, 0, 1, 0}},
    {{MMCF_GLOWSTONE_CLUSTER, 16, 3, 16, 0.80f, 0, 1, 0}, {MMCF_WARPED_FUNGUS, 7, 1, 6, 0.75f, 0, 0, 0}},
    {{MMCF_GLOWSTONE_CLUSTER, 18, 3, 16, 0.75f, 0, 1, 0}, {MMCF_AMBER_FUNGUS, 5, 1, 9, 0.60f, 0, 0, 0}}};

// feature height bounds (biomeFuncs.hpp:1042-1074, 1210-1223)
__device__ constexpr int kFeatureBounds[MMGEN_NUM_FEATURES][2] = {
    {0, 0}, {-6, 6}, {-3, 12}, {0, 20}, {0, 110}, {0, 15}, {-5, 75}, {-3, 50}, {0, 30}, {0, 15}, {0, 8}, {0, 10}, {0, 38}, {0, 17}, {0, 5},
    {0, 6}, {0, 120}, {-3, 32}, {-6, 64}, {0, 28}, {0, 15}};
__device__ constexpr int kCaveFeatureBounds[MMGEN_NUM_CAVE_FEATURES][2] = {
    {0, 0}, {-3, 3}, {-3, 3}, {0, 0}, {0, 6}, {-12, 12}, {-12, 12}, {-8, 8}, {-2, 3}, {-2, 5}};

// decorators: block, chance, nUnder, under[3], replace (AIR default / WATER), second block (AIR = none), fromCeiling
#define DG(b, ch, n, u0, u1, u2, rep, sec, ceil) {b, ch, n, {u0, u1, u2}, rep, sec, ceil}
#define GB MMB_GRASS_BLOCK
#define JGB MMB_JUNGLE_GRASS_BLOCK
__device__ constexpr int kDecoCount[MMGEN_NUM_BIOMES] = {7, 2, 0, 0, 0, 0, 1, 0, 1, 0, 0, 5, 5, 0, 4, 0, 5, 1, 4, 4, 2, 1, 7, 2};
__device__ constexpr DecoGen kDecoGens[MMGEN_NUM_BIOMES][7] = {
    /* CORAL_REEF */ {DG(MMB_SEAGRASS, 0.200f, 2, MMB_SAND, MMB_SMOOTH_SAND, 0, MMB_WATER, MMB_AIR, 0),
                      DG(MMB_TALL_SEAGRASS_BOTTOM, 0.040f, 2, MMB_SAND, MMB_SMOOTH_SAND, 0, MMB_WATER, MMB_TALL_SEAGRASS_TOP, 0),
                      DG(MMB_BRAIN_CORAL, 0.030f, 2, MMB_SAND, MMB_SMOOTH_SAND, 0, MMB_WATER, MMB_WATER, 0),
                      DG(MMB_BUBBLE_CORAL, 0.030f, 2, MMB_SAND, MMB_SMOOTH_SAND, 0, MMB_WATER, MMB_WATER, 0),
                      DG(MMB_FIRE_CORAL, 0.030f, 2, MMB_SAND, MMB_SMOOTH_SAND, 0, MMB_WATER, MMB_WATER, 0),
                      DG(MMB_HORN_CORAL, 0.030f, 2, MMB_SAND, MMB_SMOOTH_SAND, 0, MMB_WATER, MMB_WATER, 0),
                      DG(MMB_TUBE_CORAL, 0.030f, 2, MMB_SAND, MMB_SMOOTH_SAND, 0, MMB_WATER, MMB_WATER, 0)},
    /* ARCHIPELAGO */ {DG(MMB_GRASS, 0.200f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0), DG(MMB_LILY_OF_THE_VALLEY, 0.025f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0)},
    {}, {}, {}, {},
    /* TROPICAL_BEACH */ {DG(MMB_JUNGLE_GRASS, 0.1f, 1, JGB, 0, 0, MMB_AIR, MMB_AIR, 0)},
    {},
    /* SAVANNA */ {DG(MMB_SAVANNA_GRASS, 0.1f, 1, MMB_SAVANNA_GRASS_BLOCK, 0, 0, MMB_AIR, MMB_AIR, 0)},
    {}, {},
    /* REDWOOD_FOREST */ {DG(MMB_GRASS, 0.200f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0), DG(MMB_TALL_GRASS_BOTTOM, 0.080f, 1, GB, 0, 0, MMB_AIR, MMB_TALL_GRASS_TOP, 0),
                          DG(MMB_OXEYE_DAISY, 0.040f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0), DG(MMB_LILY_OF_THE_VALLEY, 0.040f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0),
                          DG(MMB_PEONY_BOTTOM, 0.020f, 1, GB, 0, 0, MMB_AIR, MMB_PEONY_TOP, 0)},
    /* SHREKS_SWAMP */ {DG(MMB_JUNGLE_GRASS, 0.300f, 1, JGB, 0, 0, MMB_AIR, MMB_AIR, 0), DG(MMB_JUNGLE_FERN, 0.050f, 1, JGB, 0, 0, MMB_AIR, MMB_AIR, 0),
                        DG(MMB_CORNFLOWER, 0.030f, 1, JGB, 0, 0, MMB_AIR, MMB_AIR, 0), DG(MMB_BLUE_ORCHID, 0.030f, 1, JGB, 0, 0, MMB_AIR, MMB_AIR, 0),
                        DG(MMB_ALLIUM, 0.030f, 1, JGB, 0, 0, MMB_AIR, MMB_AIR, 0)},
    {},
    /* LUSH_BIRCH_FOREST */ {DG(MMB_GRASS, 0.300f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0), DG(MMB_PEONY_BOTTOM, 0.020f, 1, GB, 0, 0, MMB_AIR, MMB_PEONY_TOP, 0),
                             DG(MMB_LILAC_BOTTOM, 0.020f, 1, GB, 0, 0, MMB_AIR, MMB_LILAC_TOP, 0), DG(MMB_DANDELION, 0.040f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0)},
    {},
    /* JUNGLE */ {DG(MMB_JUNGLE_GRASS, 0.400f, 1, JGB, 0, 0, MMB_AIR, MMB_AIR, 0), DG(MMB_TALL_JUNGLE_GRASS_BOTTOM, 0.200f, 1, JGB, 0, 0, MMB_AIR, MMB_TALL_JUNGLE_GRASS_TOP, 0),
                  DG(MMB_PITCHER_BOTTOM, 0.030f, 1, JGB, 0, 0, MMB_AIR, MMB_PITCHER_TOP, 0), DG(MMB_JUNGLE_FERN, 0.120f, 1, JGB, 0, 0, MMB_AIR, MMB_AIR, 0),
                  DG(MMB_BLUE_ORCHID, 0.040f, 1, JGB, 0, 0, MMB_AIR, MMB_AIR, 0)},
    /* RED_DESERT */ {DG(MMB_DEAD_BUSH, 0.020f, 1, MMB_RED_SAND, 0, 0, MMB_AIR, MMB_AIR, 0)},
    /* PURPLE_MUSHROOMS */ {DG(MMB_SMALL_PURPLE_MUSHROOM, 0.100f, 1, MMB_MYCELIUM, 0, 0, MMB_AIR, MMB_AIR, 0),
                            DG(MMB_SMALL_MAGENTA_CRYSTAL, 0.005f, 3, MMB_STONE, MMB_TUFF, MMB_CALCITE, MMB_AIR, MMB_AIR, 0),
                            DG(MMB_SMALL_CYAN_CRYSTAL, 0.005f, 3, MMB_STONE, MMB_TUFF, MMB_CALCITE, MMB_AIR, MMB_AIR, 0),
                            DG(MMB_SMALL_GREEN_CRYSTAL, 0.005f, 3, MMB_STONE, MMB_TUFF, MMB_CALCITE, MMB_AIR, MMB_AIR, 0)},
    /* CRYSTALS */ {DG(MMB_SMALL_PURPLE_MUSHROOM, 0.020f, 1, MMB_MYCELIUM, 0, 0, MMB_AIR, MMB_AIR, 0),
                    DG(MMB_SMALL_MAGENTA_CRYSTAL, 0.025f, 3, MMB_STONE, MMB_TUFF, MMB_CALCITE, MMB_AIR, MMB_AIR, 0),
                    DG(MMB_SMALL_CYAN_CRYSTAL, 0.025f, 3, MMB_STONE, MMB_TUFF, MMB_CALCITE, MMB_AIR, MMB_AIR, 0),
                    DG(MMB_SMALL_GREEN_CRYSTAL, 0.025f, 3, MMB_STONE, MMB_TUFF, MMB_CALCITE, MMB_AIR, MMB_AIR, 0)},
    /* OASIS */ {DG(MMB_JUNGLE_GRASS, 0.200f, 1, JGB, 0, 0, MMB_AIR, MMB_AIR, 0), DG(MMB_CORNFLOWER, 0.020f, 1, JGB, 0, 0, MMB_AIR, MMB_AIR, 0)},
    /* DESERT */ {DG(MMB_DEAD_BUSH, 0.030f, 1, MMB_RED_SAND, 0, 0, MMB_AIR, MMB_AIR, 0)},
    /* PLAINS */ {DG(MMB_GRASS, 0.200f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0), DG(MMB_RED_TULIP, 0.010f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0),
                  DG(MMB_ORANGE_TULIP, 0.010f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0), DG(MMB_WHITE_TULIP, 0.010f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0),
                  DG(MMB_PINK_TULIP, 0.010f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0), DG(MMB_DANDELION, 0.030f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0),
                  DG(MMB_POPPY, 0.030f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0)},
    /* MOUNTAINS */ {DG(MMB_GRASS, 0.050f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0), DG(MMB_LILY_OF_THE_VALLEY, 0.015f, 1, GB, 0, 0, MMB_AIR, MMB_AIR, 0)}};

__device__ constexpr int kCaveDecoCount[MMGEN_NUM_CAVE_BIOMES] = {0, 6, 3, 3, 2};
__device__ constexpr DecoGen kCaveDecoGens[MMGEN_NUM_CAVE_BIOMES][6] = {
    {},
    {DG(MMB_SMALL_MAGENTA_CRYSTAL, 0.015f, 0, 0, 0, 0, MMB_AIR, MMB_AIR, 0), DG(MMB_SMALL_CYAN_CRYSTAL, 0.015f, 0, 0, 0, 0, MMB_AIR, MMB_AIR, 0),
     DG(MMB_SMALL_GREEN_CRYSTAL, 0.015f, 0, 0, 0, 0, MMB_AIR, MMB_AIR, 0), DG(MMB_HANGING_SMALL_MAGENTA_CRYSTAL, 0.015f, 0, 0, 0, 0, MMB_AIR, MMB_AIR, 1),
     DG(MMB_HANGING_SMALL_CYAN_CRYSTAL, 0.015f, 0, 0, 0, 0, MMB_AIR, MMB_AIR, 1), DG(MMB_HANGING_SMALL_GREEN_CRYSTAL, 0.015f, 0, 0, 0, 0, MMB_AIR, MMB_AIR, 1)},
    {DG(MMB_GRASS, 0.100f, 1, MMB_MOSS, 0, 0, MMB_AIR, MMB_AIR, 0), DG(MMB_TALL_GRASS_BOTTOM, 0.030f, 1, MMB_MOSS, 0, 0, MMB_AIR, MMB_TALL_GRASS_TOP, 0),
     DG(MMB_TORCHFLOWER, 0.020f, 1, MMB_MOSS, 0, 0, MMB_AIR, MMB_AIR, 0)},
    {DG(MMB_WARPED_MUSHROOM, 0.020f, 2, MMB_WARPED_DEEPSLATE, MMB_WARPED_BLACKSTONE, 0, MMB_AIR, MMB_AIR, 0),
     DG(MMB_WARPED_ROOTS, 0.060f, 2, MMB_WARPED_DEEPSLATE, MMB_WARPED_BLACKSTONE, 0, MMB_AIR, MMB_AIR, 0),
     DG(MMB_NETHER_SPROUTS, 0.040f, 2, MMB_WARPED_DEEPSLATE, MMB_WARPED_BLACKSTONE, 0, MMB_AIR, MMB_AIR, 0)},
    {DG(MMB_INFECTED_MUSHROOM, 0.020f, 2, MMB_AMBER_DEEPSLATE, MMB_AMBER_BLACKSTONE, 0, MMB_AIR, MMB_AIR, 0),
     DG(MMB_AMBER_ROOTS, 0.060f, 2, MMB_AMBER_DEEPSLATE, MMB_AMBER_BLACKSTONE, 0, MMB_AIR, MMB_AIR, 0)}};

// Chebyshev reach max(|dx|, |dz|) in blocks beyond which placeFeature / placeCaveFeature cannot return true
__device__ constexpr int kFeatureReach[MMGEN_NUM_FEATURES] = {
    /*NONE*/ 0, /*SPHERE*/ 5, /*CORAL*/ 8, /*KELP*/ 0, /*ICEBERG*/ 40, /*ACACIA*/ 15, /*REDWOOD*/ 20, /*CYPRESS*/ 12, /*BIRCH*/ 8,
    /*PINE_TREE*/ 6, /*PINE_SHRUB*/ 6, /*RAFFLESIA*/ 15, /*LARGE_JUNGLE*/ 15, /*SMALL_JUNGLE*/ 8, /*TINY_JUNGLE*/ 1,
    /*MEDIUM_PURPLE_MUSHROOM*/ 8, /*PURPLE_MUSHROOM*/ 127, /*MEDIUM_CRYSTAL*/ 25, /*CRYSTAL*/ 25, /*PALM*/ 24, /*CACTUS*/ 5};
__device__ constexpr int kCaveFeatureReach[MMGEN_NUM_CAVE_FEATURES] = {
    /*NONE*/ 0, /*TEST pillars*/ 0, 0, /*CAVE_VINE*/ 0, /*GLOWSTONE_CLUSTER*/ 6, /*STORMLIGHT*/ 8, /*CEILING_STORMLIGHT*/ 8,
    /*CRYSTAL_PILLAR*/ 7, /*WARPED_FUNGUS*/ 6, /*AMBER_FUNGUS*/ 4};

// ---------------------------------------------------------------------------------------------------------
// F1 — placement generation (chunk.cu:999-1156)
// ---------------------------------------------------------------------------------------------------------
MM_DEV bool is_feature_pos(int wx, int wz, int cell, int pad, int seed)      // chunk.cu:999-1008
{
    const float fc = (float)cell;
    const int cornerX = (int)(__builtin_floorf((float)wx / fc) * fc), cornerZ = (int)(__builtin_floorf((float)wz / fc) * fc);
    const int inner = cell - 2 * pad;
    const f2 r = rand2from3((float)cornerX, (float)cornerZ, (float)seed);
    const int px = cornerX + pad + (int)__builtin_floorf(r.x * (float)inner);
    const int pz = cornerZ + pad + (int)__builtin_floorf(r.y * (float)inner);
    return wx == px && wz == pz;
}

#define F1_KEEP 4          // cave placements per column the counting pass of k_feature_placements keeps in registers (a column rarely has two)
MM_DEV unsigned pack_found(int feature, int canReplace, int y, int layerHeight)      // y, layerHeight in 0 .. 511
{
    return (unsigned)feature | ((unsigned)(canReplace != 0) << 8) | ((unsigned)(y & 1023) << 9) | ((unsigned)layerHeight << 19);
}
// Walks one column exactly like generateColumnFeaturePlacements (chunk.cu:1041-1145).  WRITE=false only counts.
template <bool WRITE>
MM_DEV void column_placements(int wx, int wz, float height, const float* cbw /*stride 256*/, const float* clayers /*stride 256*/,
                              const mmgen_cave_layer* ccl, int& nSurf, int& nCave, mmgen_feature_placement* surfOut,
                              mmgen_cave_feature_placement* caveOut, int caveCap, unsigned (*found)[F1_KEEP + 1] = nullptr /* counting pass: the
                              surface placement and the first F1_KEEP cave placements, packed (pack_found), so that the writing pass need not walk
                              the column again */)
{
    nSurf = 0; nCave = 0;
    const int ground = (int)height;
    MinStd rng = rng3(wx, wz, 329828101);
    // the column's 24 weights in one round trip, under the cave-layer walk (random_biome below would read them one dependent load at a time)
    float w24[MMGEN_NUM_BIOMES];
#pragma unroll
    for (int b = 0; b < MMGEN_NUM_BIOMES; ++b) w24[b] = cbw[256 * b];

    bool surfaceIsCave = false;
    for (int k = 0; k < MMGEN_MAX_CAVE_LAYERS_PER_COLUMN; ++k) {
        const int start = ccl[k].start, end = ccl[k].end;
        if (start == 384 || ground <= start) break;
        const int layerHeight = end - start;
        for (int pass = 0; pass < 2; ++pass) {           // pass 0: bottom biome gens (floor), pass 1: top biome gens (ceiling)
            if (pass == 1 && end == 384) break;
            const int cb = pass == 0 ? ccl[k].bottom_biome : ccl[k].top_biome;
            const bool top = pass == 1;
            for (int g = 0; g < kCaveGenCount[cb]; ++g) {
                const CaveGen& gen = kCaveGens[cb][g];
                const int seed = (int)gen.feature * (top ? 58321 : 98239) + k * (top ? 871503 : 191702);
                const float rand = rng.u01();             // every gen consumes one draw, accepted or not
                if (rand >= gen.chance || (top != (gen.fromCeiling != 0)) || (!gen.canLava && (top ? end : (start + 1)) <= MMGEN_LAVA_LEVEL)
                    || layerHeight < (int)gen.minLayerHeight)
                    continue;
                if (is_feature_pos(wx, wz, gen.cell, gen.pad, seed)) {
                    if (!WRITE && found && nCave < F1_KEEP) (*found)[1 + nCave] = pack_found(gen.feature, gen.canReplace, start + 1, layerHeight);
                    if (WRITE && nCave < caveCap) {
                        mmgen_cave_feature_placement p = {};
                        p.feature = gen.feature; p.pos[0] = wx; p.pos[1] = start + 1; p.pos[2] = wz; p.layer_height = layerHeight;
                        p.can_replace_blocks = gen.canReplace;
                        caveOut[nCave] = p;
                    }
                    ++nCave;
                    break;
                }
            }
        }
        if (ground > start && ground <= end) { surfaceIsCave = true; break; }
    }

    if (!surfaceIsCave) {
        int biome = MMBIO_PLAINS;
        {
            float rand = rng.u01();
            bool found = false;
#pragma unroll
            for (int b = 0; b < MMGEN_NUM_BIOMES; ++b) { rand -= w24[b]; if (!found && rand <= 0.f) { biome = b; found = true; } }      // random_biome (mm_biome.cuh), unrolled over registers
        }
        for (int g = 0; g < kSurfGenCount[biome]; ++g) {
            const SurfGen& gen = kSurfGens[biome][g];
            if (rng.u01() >= gen.chance) continue;
            if (gen.nTop > 0) {
                bool canPlace = false;
                for (int t = 0; t < gen.nTop; ++t) {
                    const int l = gen.topMat[t];
                    const float ls = clayers[256 * l];
                    const float le = clayers[256 * (l + 1)];
                    if (ls > height || le < height || gmin(le, height) - ls < gen.topMin[t]) continue;
                    canPlace = true;
                    break;
                }
                if (!canPlace) continue;
            }
            if (is_feature_pos(wx, wz, gen.cell, gen.pad, (int)gen.feature * 518721)) {
                if (!WRITE && found) (*found)[0] = pack_found(gen.feature, gen.canReplace, ground + 1, 0);
                if (WRITE) {
                    mmgen_feature_placement p = {};
                    p.feature = gen.feature; p.pos[0] = wx; p.pos[1] = ground + 1; p.pos[2] = wz; p.can_replace_blocks = gen.canReplace;
                    surfOut[0] = p;
                }
                nSurf = 1;
                break;
            }
        }
    }
}

// Lazy ring (region path): a ring chunk outside the rectangle only matters through the placements that can reach the rectangle, and
// whether a column can produce one is decidable BEFORE its caves exist: a cave feature reaches at most kCaveFeatureReach blocks, and a
// surface feature only ever stands on the jittered grid point of its gen (is_feature_pos depends on position and gen alone, chunk.cu:999-1008;
// the rng stream only decides which of the possible gens fires).  Columns that can produce nothing that reaches the rectangle need no
// cave noise at all: that is ~80 % of a 3-chunk ring.  need[cell][column] = 1 for every column of a fully computed cell.
__global__ void __launch_bounds__(256)
k_ring_need(const float* __restrict__ bw, const int2* __restrict__ chunkPos, const int* __restrict__ chunkList, const uint8_t* __restrict__ cellLazy,
            int rx0, int rz0, int rx1, int rz1 /*the rectangle in block coordinates, inclusive*/, uint8_t* __restrict__ colNeed)
{
    const int chunk = chunkList[blockIdx.x], t = threadIdx.x;
    uint8_t need = 1;
    if (cellLazy[chunk]) {
        const int2 cp = chunkPos[chunk];
        const int wx = cp.x + (t & 15), wz = cp.y + (t >> 4);
        const int dist = imax(imax(rx0 - wx, wx - rx1), imax(imax(rz0 - wz, wz - rz1), 0));      // Chebyshev distance to the rectangle
        int maxCave = 0;
        for (int f = 0; f < MMGEN_NUM_CAVE_FEATURES; ++f) maxCave = imax(maxCave, kCaveFeatureReach[f]);
        need = dist <= maxCave;
        const float* cbw = bw + (size_t)MMGEN_BIOME_WEIGHTS_SIZE * chunk + t;
        for (int b = 0; b < MMGEN_NUM_BIOMES && !need; ++b) {
            // random_biome can return a biome only if its weight is positive - or biome 0 when the draw is exactly 0, or PLAINS when the
            // draw outlasts every weight (a draw that rounds to 1 over weights that sum to less)
            if (b != 0 && b != MMBIO_PLAINS && !(cbw[256 * b] > 0.f)) continue;
            for (int g = 0; g < kSurfGenCount[b] && !need; ++g) {
                const SurfGen& gen = kSurfGens[b][g];
                if (dist <= kFeatureReach[gen.feature] && is_feature_pos(wx, wz, gen.cell, gen.pad, (int)gen.feature * 518721)) need = 1;
            }
        }
    }
    colNeed[(size_t)256 * chunk + t] = need;
}

__global__ void __launch_bounds__(256)
k_feature_placements(const float* __restrict__ hf, const float* __restrict__ bw, const float* __restrict__ layers,
                     const mmgen_cave_layer* __restrict__ caveLayers, const int2* __restrict__ chunkPos,
                     mmgen_feature_placement* __restrict__ fpOut, mmgen_cave_feature_placement* __restrict__ cfpOut, int* __restrict__ counts,
                     const int* __restrict__ chunkList, const uint8_t* __restrict__ colNeed /*nullable: [chunk][256], 0 = column skipped (lazy ring)*/)
{
    // (no simplex noise in this kernel: the tables stay where they are)
    __shared__ int s_ns[4];
    const int chunk = chunkList ? chunkList[blockIdx.x] : blockIdx.x, t = threadIdx.x;
    const int2 cp = chunkPos[chunk];
    const int wx = cp.x + (t & 15), wz = cp.y + (t >> 4);
    const float height = hf[(size_t)256 * chunk + t];
    const float* cbw = bw + (size_t)MMGEN_BIOME_WEIGHTS_SIZE * chunk + t;
    const float* cl = layers + (size_t)MMGEN_LAYERS_SIZE * chunk + t;
    const mmgen_cave_layer* ccl = caveLayers + ((size_t)256 * chunk + t) * MMGEN_MAX_CAVE_LAYERS_PER_COLUMN;

    int ns = 0, nc = 0;
    unsigned found[F1_KEEP + 1] = {};
    const bool need = !colNeed || colNeed[(size_t)256 * chunk + t];
    if (need) column_placements<false>(wx, wz, height, cbw, cl, ccl, ns, nc, nullptr, nullptr, 0, &found);
    // exclusive prefix of the two per-column counts in column order: a shuffle scan inside each wave (both counts in one word: a chunk
    // holds far fewer than 65 536 of either), then the four wave totals through LDS
    const unsigned both = (unsigned)ns | ((unsigned)nc << 16);
    unsigned scan = both;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned up = (unsigned)__shfl_up((int)scan, d, 64);
        if ((t & 63) >= d) scan += up;
    }
    if ((t & 63) == 63) s_ns[t >> 6] = (int)scan;
    __syncthreads();
    unsigned base = 0, total = 0;
    for (int w = 0; w < 4; ++w) { const unsigned v = (unsigned)s_ns[w]; if (w < (t >> 6)) base += v; total += v; }
    const unsigned excl = base + scan - both;
    const int offS = (int)(excl & 0xffffu), offC = (int)(excl >> 16);
    if (t == 255) { counts[2 * chunk] = (int)(total & 0xffffu); counts[2 * chunk + 1] = (int)(total >> 16); }
    if (ns + nc == 0) return;
    mmgen_feature_placement* so = fpOut + (size_t)MMGEN_FP_CAP * chunk + offS;
    mmgen_cave_feature_placement* co = cfpOut + (size_t)MMGEN_CFP_CAP * chunk + offC;
    const int capLeft = imax(0, MMGEN_CFP_CAP - offC);
    if (nc > F1_KEEP) { column_placements<true>(wx, wz, height, cbw, cl, ccl, ns, nc, so, co, capLeft); return; }      // more than the registers kept: walk again
    if (ns) {
        mmgen_feature_placement p = {};
        p.feature = (uint8_t)(found[0] & 255u); p.pos[0] = wx; p.pos[1] = (int)((found[0] >> 9) & 1023u); p.pos[2] = wz; p.can_replace_blocks = (uint8_t)((found[0] >> 8) & 1u);
        so[0] = p;
    }
#pragma unroll
    for (int i = 0; i < F1_KEEP; ++i) {
        if (i < nc && i < capLeft) {
            mmgen_cave_feature_placement p = {};
            p.feature = (uint8_t)(found[1 + i] & 255u); p.pos[0] = wx; p.pos[1] = (int)((found[1 + i] >> 9) & 1023u); p.pos[2] = wz;
            p.layer_height = (int)(found[1 + i] >> 19); p.can_replace_blocks = (uint8_t)((found[1 + i] >> 8) & 1u);
            co[i] = p;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// F2 — gather in the reference's fixed offset order (chunk.cu:1158-1187) + host part of Chunk::fill (chunk.cu:1555-1601):
// union of height bounds over the un-truncated lists, truncation to 2048 / 4096, NONE sentinel.
// ---------------------------------------------------------------------------------------------------------
__device__ constexpr int kGatherDX[49] = {0, 0, 1, 1, 1, 0, -1, -1, -1, 2, 2, 2, 1, 0, -1, -2, -2, -2, -2, -2, -1, 0, 1, 2, 2,
                                          -3, -2, -1, 0, 1, 2, 3, 3, 3, 3, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -3, -3, -3};
__device__ constexpr int kGatherDZ[49] = {0, 1, 1, 0, -1, -1, -1, 0, 1, 0, 1, 2, 2, 2, 2, 2, 1, 0, -1, -2, -2, -2, -2, -2, -1,
                                          -3, -3, -3, -3, -3, -3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 3, 3, 3, 3, 2, 1, 0, -1, -2};

// stable (order-preserving) compaction of one gathered list into `out`, keeping the entries whose horizontal reach box meets the
// target chunk's 16 x 16 footprint; entries at index >= CAP are dropped first, exactly like the reference's truncation, so the kept
// entries are a subsequence of the reference's list and the dropped ones could not have claimed a voxel of this chunk.
template <class Entry, int SRC_CAP, int CAP, bool CAVE>
MM_DEV int gather_filtered(const Entry* __restrict__ src, const int* s_off, const int* s_srcChunk, int tot, int ox, int oz, Entry* __restrict__ out,
                           int* s_w /*[4]*/, int& lo, int& hi)
{
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int n = imin(tot, CAP);
    int base = 0;
    for (int r0 = 0; r0 < n; r0 += 256) {
        const int i = r0 + t;
        bool keep = false;
        Entry p = {};
        if (i < n) {
            int k = 0;
            while (s_off[k + 1] <= i) ++k;
            p = src[(size_t)SRC_CAP * s_srcChunk[k] + (i - s_off[k])];
            const int reach = CAVE ? kCaveFeatureReach[p.feature] : kFeatureReach[p.feature];
            keep = p.pos[0] + reach >= ox && p.pos[0] - reach <= ox + 15 && p.pos[2] + reach >= oz && p.pos[2] - reach <= oz + 15;
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int c = s_w[w]; if (w < wave) before += c; total += c; }
        if (keep) {
            out[base + before + __popcll(m & ((1ull << lane) - 1ull))] = p;
            if constexpr (CAVE) {
                lo = imin(lo, p.pos[1] + kCaveFeatureBounds[p.feature][0]); hi = imax(hi, p.pos[1] + p.layer_height + kCaveFeatureBounds[p.feature][1]);
            } else {
                lo = imin(lo, p.pos[1] + kFeatureBounds[p.feature][0]); hi = imax(hi, p.pos[1] + kFeatureBounds[p.feature][1]);
            }
        }
        base += total;
        __syncthreads();
    }
    return base;
}

__global__ void __launch_bounds__(256)
k_gather_placements(const mmgen_feature_placement* __restrict__ fp, const mmgen_cave_feature_placement* __restrict__ cfp,
                    const int* __restrict__ counts, const int* __restrict__ targetChunk /*[nOut] index into source grid*/,
                    int gridW, int gridH, mmgen_feature_placement* __restrict__ gfp, mmgen_cave_feature_placement* __restrict__ gcfp,
                    int* __restrict__ bounds, const int2* __restrict__ gridPos /*world block origin of every source-grid chunk; null = keep everything*/,
                    int* maxGathered /*nullable: [0] / [1] raised to the longest un-truncated surface / cave list*/,
                    int* capHost /*nullable: host-visible word raised to a source cell's cave count beyond MMGEN_CFP_CAP*/,
                    int* capMax /*nullable: raised to the largest cave count of the source cells*/,
                    unsigned* zeroWords /*nullable*/, int nZeroWords /*words the first workgroup clears: the rasterisers' work counters*/,
                    const float* __restrict__ hfGrid /*nullable*/, float* __restrict__ hfOut /*the targets' heightfields, dense (what k_select did)*/)
{
    __shared__ int s_offS[50], s_offC[50], s_src[49];
    __shared__ int s_b[4], s_w[4];
    const int o = blockIdx.x, t = threadIdx.x;
    const int c = targetChunk[o];
    const int cx = c % gridW, cz = c / gridW;
    // the 49 source cells in the reference's order: lane k of the first wave reads cell k's two counts, a shuffle scan gives the offsets
    if (t < 64) {
        int nS = 0, nC = 0, n = -1;
        if (t < 49) {
            const int nx = cx + kGatherDX[t], nz = cz + kGatherDZ[t];
            if (nx >= 0 && nx < gridW && nz >= 0 && nz < gridH) {
                n = nx + gridW * nz; nS = counts[2 * n];
                const int raw = counts[2 * n + 1];
                nC = imin(raw, MMGEN_CFP_CAP);
                // every list length the gather uses, local or received, against the product's one capacity (include/mmgen.h:
                // MMGEN_ERROR_PLACEMENT_OVERFLOW): a plain look first, the atomics only where they raise something
                if (capHost && raw > MMGEN_CFP_CAP) __hip_atomic_fetch_max(capHost, raw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (capMax && raw > *capMax) atomicMax(capMax, raw);
            }
            s_src[t] = n;
        }
        int inS = nS, inC = nC;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int uS = __shfl_up(inS, d, 64), uC = __shfl_up(inC, d, 64);
            if (t >= d) { inS += uS; inC += uC; }
        }
        if (t < 49) { s_offS[t] = inS - nS; s_offC[t] = inC - nC; }
        if (t == 48) { s_offS[49] = inS; s_offC[49] = inC; }
        if (t == 0) { s_b[0] = 384; s_b[1] = -1; s_b[2] = 384; s_b[3] = -1; }
    }
    if (zeroWords && o == 0) for (int i = t; i < nZeroWords; i += 256) zeroWords[i] = 0u;
    if (hfOut) hfOut[(size_t)256 * o + t] = hfGrid[(size_t)256 * c + t];
    __syncthreads();
    const int totS = s_offS[49], totC = s_offC[49];
    // (a plain look first: two atomics per chunk on two addresses serialise in L2 - 0.09 ms for the bench tile - and only a few ever raise the maximum)
    if (maxGathered && t == 0) {
        if (totS > maxGathered[0]) atomicMax(&maxGathered[0], totS);
        if (totC > maxGathered[1]) atomicMax(&maxGathered[1], totC);
    }
    mmgen_feature_placement* go = gfp + (size_t)MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK * o;
    mmgen_cave_feature_placement* gc = gcfp + (size_t)MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK * o;
    int lo0 = 384, hi0 = -1, lo1 = 384, hi1 = -1;
    if (gridPos) {
        // region path: the list only feeds k_apply_features, so entries that cannot reach this chunk are dropped here once instead of
        // being skipped 256 times by the per-column filters (the per-stage ABI keeps the reference's full lists: gridPos == null)
        const int2 org = gridPos[c];
        const int nS = gather_filtered<mmgen_feature_placement, MMGEN_FP_CAP, MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK, false>(fp, s_offS, s_src, totS, org.x, org.y,
                                                                                                                            go, s_w, lo0, hi0);
        const int nC = gather_filtered<mmgen_cave_feature_placement, MMGEN_CFP_CAP, MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK, true>(cfp, s_offC, s_src, totC, org.x,
                                                                                                                                     org.y, gc, s_w, lo1, hi1);
        atomicMin(&s_b[0], lo0); atomicMax(&s_b[1], hi0); atomicMin(&s_b[2], lo1); atomicMax(&s_b[3], hi1);
        __syncthreads();
        if (t == 0) {
            if (nS < MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK) { mmgen_feature_placement z = {}; go[nS] = z; }
            if (nC < MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK) { mmgen_cave_feature_placement z = {}; gc[nC] = z; }
            bounds[4 * o] = s_b[0]; bounds[4 * o + 1] = s_b[1]; bounds[4 * o + 2] = s_b[2]; bounds[4 * o + 3] = s_b[3];
        }
        return;
    }
    for (int i = t; i < totS; i += 256) {
        int k = 0;
        while (s_offS[k + 1] <= i) ++k;
        const mmgen_feature_placement p = fp[(size_t)MMGEN_FP_CAP * s_src[k] + (i - s_offS[k])];
        lo0 = imin(lo0, p.pos[1] + kFeatureBounds[p.feature][0]);
        hi0 = imax(hi0, p.pos[1] + kFeatureBounds[p.feature][1]);
        if (i < MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK) go[i] = p;
    }
    for (int i = t; i < totC; i += 256) {
        int k = 0;
        while (s_offC[k + 1] <= i) ++k;
        const mmgen_cave_feature_placement p = cfp[(size_t)MMGEN_CFP_CAP * s_src[k] + (i - s_offC[k])];
        lo1 = imin(lo1, p.pos[1] + kCaveFeatureBounds[p.feature][0]);
        hi1 = imax(hi1, p.pos[1] + p.layer_height + kCaveFeatureBounds[p.feature][1]);
        if (i < MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK) gc[i] = p;
    }
    atomicMin(&s_b[0], lo0); atomicMax(&s_b[1], hi0); atomicMin(&s_b[2], lo1); atomicMax(&s_b[3], hi1);
    __syncthreads();
    if (t == 0) {
        if (totS < MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK) { mmgen_feature_placement z = {}; go[totS] = z; }          // NONE sentinel
        if (totC < MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK) { mmgen_cave_feature_placement z = {}; gc[totC] = z; }
        bounds[4 * o] = s_b[0]; bounds[4 * o + 1] = s_b[1]; bounds[4 * o + 2] = s_b[2]; bounds[4 * o + 3] = s_b[3];
    }
}

// ---------------------------------------------------------------------------------------------------------
// K6b — feature evaluation per voxel (second half of kernFill, chunk.cu:1438-1509): k_apply_features, further down.
//
// The reference makes every voxel walk the whole gathered list (up to 2048 + 4096 entries).  A placement can only claim voxels within a
// fixed horizontal reach of its position (every rasteriser starts with, or implies, such a bound — table above, validated against the
// oracle in tests/test_oracle_stages.py::test_feature_reach_table), and for most features the placement's own first draws bound its
// claim per column far more tightly (surface_extent / cave_extent).  So the lists are filtered per UNIT of 8 columns in list order (wave
// ballots + popcount prefix: a stable compaction, first match still wins), and only (voxel, placement) pairs inside those extents are
// ever evaluated.  Entries skipped by the filters would have returned false.
// ---------------------------------------------------------------------------------------------------------
#define APPLY_COLS 4                      // waves per workgroup
#define APPLY_THREADS (64 * APPLY_COLS)
#ifndef APPLY_UNIT_W
#define APPLY_UNIT_W 4                    // one wave = one UNIT at a time: W x H columns of a chunk (streaming kernel: 4 x 1 1.78 ms, 2 x 2 1.81,
#define APPLY_UNIT_H 2                    // 4 x 2 1.35, 8 x 1 1.41, 2 x 4 1.44, 16 x 1 1.78, 8 x 2 1.72)
#endif
#define APPLY_UNIT_NCOL (APPLY_UNIT_W * APPLY_UNIT_H)
#define APPLY_UNITS_PER_CHUNK ((16 / APPLY_UNIT_W) * (16 / APPLY_UNIT_H))
#define APPLY_COUNTERS 8                  // work counters per launch, 64 B apart
#define APPLY_UNIT_CAP 128                // placements that can reach one unit, surface + cave
#define APPLY_ENT_CAP 192                 // (placement, column) pairs of one unit with a non-empty vertical extent
#define APPLY_ITEM_CAP 65535              // (placement, voxel) pairs of one unit: item offsets are kept as 16-bit numbers
static_assert(APPLY_UNIT_NCOL <= 16 && APPLY_UNIT_CAP <= 128, "s_ent packing");

MM_DEV int wave_min(int v) { for (int o = 32; o > 0; o >>= 1) v = imin(v, __shfl_xor(v, o)); return v; }
MM_DEV int wave_max(int v) { for (int o = 32; o > 0; o >>= 1) v = imax(v, __shfl_xor(v, o)); return v; }

// What ONE placement can claim in ONE column (dx, dz) = column - placement position: false when nothing, else the vertical extent relative
// to pos.y.  The tables are per-feature worst cases over every random draw; for the features that own most (voxel, placement) pairs of a
// generated world the rasteriser's own early-outs bound the claim much more tightly once the placement's first draws are known
// (fstate = its stream right after seeding; the draws below are the ones the rasteriser makes, in its order).  Everything excluded here
// returns false in the rasteriser: tests/test_gpu_features.py::test_tight_extents_lose_nothing.
//   PURPLE_MUSHROOM (featurePlacement.hpp:705-709): worst case reach 127 / 121 voxels tall (smallest universal scale AND the 20 % half-scale
//     draw: 7.9 M voxels per placement, 10 x all other features of the bench world together); the first three draws bound it to
//     |pos.xz| <= 35 and -1 <= pos.y <= height + 12 in the scaled frame.
//   CORAL: the first draw picks one of five shapes: two noisy ellipsoids (radius <= base + amplitude * MM_SIMPLEX3_BOUND), two
//     bundles of six rasterised segments inside [-6.5, 6.5] x [0, 8.5] x [-6.5, 6.5], one tube field that is empty from radius 3.7 on.
MM_DEV bool surface_extent(int feat, int fy, int dx, int dz, uint32_t fstate, bool noiseBounds, int& dlo, int& dhi)
{
    dlo = kFeatureBounds[feat][0]; dhi = kFeatureBounds[feat][1];
    if (feat == MMF_PURPLE_MUSHROOM) {
        MinStd frng; frng.x = fstate;
        const float universalScale = 1.f + frng.u01() * 1.2f;
        const bool half = frng.u01() < 0.2f;
        const float sc = half ? universalScale * 0.5f : universalScale;         // (p * s) * 0.5 == p * (s * 0.5): a power of two
        const float height = 25.f + frng.u01() * 30.f;
        const int reach = imin(kFeatureReach[feat], (int)(35.f / sc) + 2);      // + 2: rounding of the scaled coordinates and of this division
        dhi = imin(dhi, (int)((height + 12.f) / sc) + 2);
        if (iabs(dx) > reach || iabs(dz) > reach) return false;
        // the column in the rasteriser's scaled frame, computed the way it computes pos
        float px = (float)dx * universalScale, pz = (float)dz * universalScale;
        if (half) { px *= 0.5f; pz *= 0.5f; }
        const float hd = len2(px, pz);
        if (!(hd > 8.f)) return true;
        // beyond radius 8: pos.y >= height - 12 and |pos - top| <= 35, or the rasteriser returns false
        // (the rasteriser's |pos - top| is a rounded root of a sum that is monotone in dy^2 and equals hd's at dy = 0: the column is out iff
        // hd itself is beyond 35 - not iff fl(hd^2) > 35^2, which can hold for a kept voxel; see the stormlight spheres in cave_extent)
        if (hd > 35.f) return false;
        const float vr = __builtin_sqrtf(__builtin_fmaxf(35.f * 35.f - hd * hd, 0.f));
        dlo = imax(dlo, (int)((height - gmin(12.f, vr)) / sc) - 1);             // height - 12 >= 13
        dhi = imin(dhi, (int)((height + vr) / sc) + 2);
        if (!(hd > 11.6f)) return true;
        // beyond radius 11.6 the stem cannot reach (its spline stays inside the hull of the control points, |x|, |z| <= 6, radius <= 3):
        // only the cap is left, the points between the two planes through p1 and p2 = p1 + dir * len perpendicular to dir, within
        // `radius` of the axis.  p1, p2 and radius depend on the placement alone; they are rebuilt here with the rasteriser's own
        // statements (featurePlacement.hpp:712-760), and the slab 0 <= (pos - p1) . v <= v . v bounds this column's y.
        v3 ctrl[5];
        ctrl[0] = V3(0.f, 0.f, 0.f);
        const v3 endPoint = V3(0.f, height, 0.f);
#pragma unroll
        for (int i = 1; i < 5; ++i) {
            const float r0 = u11(frng), r1 = u11(frng), r2 = u11(frng);
            v3 off = V3(r0, r1, r2) * V3(6.f, 2.f, 6.f);
            if (i == 4) off = off * 0.6f;
            ctrl[i] = (endPoint * ((float)i / 4.f)) + off;
        }
        v3 spline[7];
        de_casteljau<5, 7>(ctrl, spline);
        const v3 p1 = spline[6];
        const v3 p2 = p1 + norm3(p1 - spline[5]) * (3.f + frng.u01() * 1.5f);
        const float radius = (7.f * frng.u01() + 12.f) * mixf(0.8f, 1.2f, (height - 33.f) / 40.f);
        const v3 v = p2 - p1;
        const float vv = dot3(v, v);
        if (len2(px - p1.x, pz - p1.z) > (radius + __builtin_sqrtf(vv)) + 0.01f) return false;     // farther than radius from every axis point
        if (!(v.y > 0.2f)) return true;                                          // (nearly) horizontal axis: the slab does not bound y
        const float a = (px - p1.x) * v.x + (pz - p1.z) * v.z;
        const float ysLo = p1.y + (0.f - a) / v.y, ysHi = p1.y + (vv - a) / v.y;
        dlo = imax(dlo, (int)__builtin_floorf(ysLo / sc) - 1);
        dhi = imin(dhi, (int)__builtin_floorf(ysHi / sc) + 2);
        return dlo <= dhi;
    }
    if (feat == MMF_CORAL) {
        if (fy > MMGEN_SEA_LEVEL - 6) return false;
        const int d2 = dx * dx + dz * dz;
        if (d2 > 64) return false;                                              // len2(pos.xz) > 8
        MinStd frng; frng.x = fstate;
        const int kind = (int)(frng.u01() * 5.f);
        if (kind == 0 || kind == 1) {
            if (!noiseBounds) return true;                                          // outside the pruning domain the simplex bound is not used
            // len3(x, y * ys, z) < radius, radius < base + amp * MM_SIMPLEX3_BOUND (+ 0.01: rounding)
            const float rmax = (kind == 0 ? (2.8f + 1.4f * frng.u01()) + 0.4f * MM_SIMPLEX3_BOUND : (2.2f + 1.7f * frng.u01()) + 1.2f * MM_SIMPLEX3_BOUND) + 0.01f;
            const float rest = rmax * rmax - (float)d2;
            if (rest <= 0.f) return false;
            const int dy = (int)(__builtin_sqrtf(rest) / (kind == 0 ? 1.15f : 1.25f)) + 1;
            dlo = imax(dlo, -dy); dhi = imin(dhi, dy);
            return true;
        }
        if (kind == 2 || kind == 3) {                                           // floor of a point of a segment inside the box above
            dlo = imax(dlo, -1); dhi = imin(dhi, 9);
            return iabs(dx) <= 7 && iabs(dz) <= 7;
        }
        if (kind == 4) {                                                        // h <= (1 + d2nd / 2) * 3.5 - 2 with d2nd <= sqrt(5); 0 * ... - 2 from radius 3.7
            dlo = imax(dlo, -1); dhi = imin(dhi, 6);
            return d2 <= 13;
        }
        return true;
    }
    const int d2 = dx * dx + dz * dz;
    if (feat == MMF_REDWOOD_TREE) {
        // pos *= sc; out when pos.y > height + 8, hd > 12, or below leavesStart - 4 outside the trunk's radius of 3
        MinStd frng; frng.x = fstate;
        const float sc = 0.6f + 0.3f * frng.u01();
        const float height = 27.f + 13.f * frng.u01();
        const float leavesStart = 10.f + 4.f * frng.u01();
        const float hd = len2((float)dx * sc, (float)dz * sc);
        // leaf clusters: |pos - center| <= 5 horizontally, |center.xz| <= 3.75 hr, hr = 1.1 - 0.5 ratio <= 1.307 (the cell is at most 6 below
        // leavesStart, leavesEnd - leavesStart >= 14.5): 9.9; branches stay within 0.5 of the segment to the centre; the trunk's radius is < 3
        if (hd > 10.f) return false;
        // trunk up to pos.y = height, leaves up to leavesEnd = height + 1.5 + u < height + 2.5
        dhi = imin(dhi, (int)((height + 2.5f) / sc) + 1);
        if (hd > 3.f) dlo = imax(dlo, (int)(leavesStart / sc) - 1);              // outside the trunk: leaves and branches, pos.y >= leavesStart
        return true;
    }
    if (feat == MMF_BIRCH_TREE) {
        MinStd frng; frng.x = fstate;
        int height = (int)(6.2f + 4.f * frng.u01());
        if (frng.u01() < 0.08f) height = (int)((float)height * 1.9f);
        dlo = imax(dlo, 0); dhi = imin(dhi, height + 6);
        if (d2 == 0) return true;
        // leaves: height - 4.5 <= y <= height + 8.1, radius <= 5 * max(0.5 x^3 - 1.5 x^2 + x) * 3.6 = 3.4642
        if (d2 > 13) return false;
        dlo = imax(dlo, height - 5);
        return true;
    }
    if (feat == MMF_PINE_TREE) {
        MinStd frng; frng.x = fstate;
        const int height = (int)(7.f + 4.f * frng.u01());
        dlo = imax(dlo, 0); dhi = imin(dhi, height + 4);
        if (d2 == 0) return true;
        if (d2 > 8) return false;                                              // leaves: len2 < mix(3, 1, ratio) <= 3, height - 6.5 <= y <= height + 3
        dlo = imax(dlo, height - 7); dhi = imin(dhi, height + 3);
        return true;
    }
    if (feat == MMF_PINE_SHRUB) {
        MinStd frng; frng.x = fstate;
        const int height = (int)(2.f + 2.f * frng.u01());
        dlo = imax(dlo, 0); dhi = imin(dhi, height + 4);
        if (d2 == 0) return true;
        if (d2 > 8) return false;                                              // jungle_leaves: radius <= 2.5 * 1.2 = 3, 0 <= y - (height - 1) <= 2.5
        dlo = imax(dlo, height - 1); dhi = imin(dhi, height + 2);
        return true;
    }
    if (feat == MMF_MEDIUM_PURPLE_MUSHROOM) {
        MinStd frng; frng.x = fstate;
        const int height = (int)(1.5f + 2.3f * frng.u01());
        dhi = imin(dhi, height + 1);
        if (d2 == 0) return true;
        if (d2 > 6) return false;                                              // cap: y == height + 1 within radius 1.8 or 2.5
        dlo = imax(dlo, height + 1);
        return true;
    }
    if (feat == MMF_MEDIUM_CRYSTAL || feat == MMF_CRYSTAL) {
        if (fy > 180) return false;
        // pos = (fp + (0, 2, 0)) * scale.  Main crystal: within 1.0 * 5.2 of the segment to endPos (|endPos.xz| <= 12, endPos.y = 18 + 8 r1);
        // small crystals: 0.8 pos within 3.0 of a segment of horizontal length <= 9: |pos.xz| <= 17.3, -5.3 <= pos.y <= endPos.y + 2
        MinStd frng; frng.x = fstate;
        float scale = 0.55f + 0.4f * frng.u01();
        if (feat == MMF_MEDIUM_CRYSTAL) scale *= 2.f;
        frng.u01();
        const float endY = 18.f + 8.f * frng.u01();
        const int reach = (int)(17.3f / scale) + 1;
        if (iabs(dx) > reach || iabs(dz) > reach) return false;
        dhi = imin(dhi, (int)((endY + 2.f) / scale) - 1);                       // (dy + 2) * scale <= endY + 2; + 1 for the rounding
        dlo = imax(dlo, -(int)(5.3f / scale) - 3);
        return true;
    }
    return true;
}

//   WARPED_FUNGUS: stem in its own column, shroomlights in the four neighbours, cap only within radius 3.7 and, within radius 2.3, exactly
//     one voxel thick (capStart == capEnd); AMBER_FUNGUS: stem, and a cap ring at Manhattan distance 1 or 2.  Neither depends on the
//     layer height, which the table bound adds.
MM_DEV bool cave_extent(int feat, int lh, int dx, int dz, uint32_t fstate, bool noiseBounds, int& dlo, int& dhi)
{
    dlo = kCaveFeatureBounds[feat][0]; dhi = lh + kCaveFeatureBounds[feat][1];
    const int ml = iabs(dx) + iabs(dz);
    if (feat == MMCF_WARPED_FUNGUS) {
        if (ml > 6) return false;
        MinStd frng; frng.x = fstate;
        const int height = (int)(2.5f + 3.0f * frng.u01());
        // (the table bounds are the reference's own per-feature height test, chunk.cu:1438-1509: part of the result, only ever narrowed)
        dhi = imin(dhi, height + 1);                                            // stem <= height, shroomlight <= height, cap <= capEnd <= height + 1
        if (ml <= 1) return true;
        const float capRadius = len2((float)dx, (float)dz);
        if (capRadius > 3.7f) return false;
        const int capEnd = height + 1 - (int)(capRadius / 2.5f);
        dhi = imin(dhi, capEnd);
        if (!(capRadius - 2.3f > 0.f)) dlo = imax(dlo, capEnd);                 // capStart = (int)((float)capEnd - (4.2 s) * 0) = capEnd
        return true;
    }
    if (feat == MMCF_AMBER_FUNGUS) {
        if (ml > 2) return false;                                               // the cap needs ml == capDist, 1 or 2
        MinStd frng; frng.x = fstate;
        const int height = (int)(4.5f + 4.5f * frng.u01());
        if (ml == 0) { dlo = imax(dlo, 0); dhi = imin(dhi, height + 1); }
        else { dlo = imax(dlo, height / 2 - 1); dhi = imin(dhi, height); }
        return true;
    }
    const int d2 = dx * dx + dz * dz;
    if (feat == MMCF_CAVE_VINE) {                                               // hangs height <= min(14, layerHeight) blocks from the ceiling
        MinStd frng; frng.x = fstate;
        const int height = imin((int)(3.f + 12.f * frng.u01()), lh);
        dlo = imax(dlo, lh - height); dhi = imin(dhi, lh);
        return true;
    }
    if (feat == MMCF_GLOWSTONE_CLUSTER && noiseBounds) {
        // r = |(dx, 1.35 dyTop, dz)| * s < 3.5 + 2 simplex2 <= 3.5 + 2 MM_SIMPLEX2_BOUND = 5.82
        MinStd frng; frng.x = fstate;
        const float sc = 1.f + 0.5f * frng.u01();
        constexpr float rMax = 3.5f + 2.f * MM_SIMPLEX2_BOUND + 0.01f;
        const float rest = rMax * rMax - (float)d2 * (sc * sc);
        if (rest <= 0.f) return false;
        const int m = (int)(__builtin_sqrtf(rest) / (1.35f * sc)) + 1;
        dlo = imax(dlo, lh - m); dhi = imin(dhi, lh + m);
        return true;
    }
    if (feat == MMCF_STORMLIGHT_SPHERE || feat == MMCF_CEILING_STORMLIGHT_SPHERE) {
        MinStd frng; frng.x = fstate;
        const float radius = 3.5f + 4.f * frng.u01();
        // The rasteriser keeps a voxel iff fl(sqrt(fl(d2 + dy^2))) <= radius.  The column's nearest voxel (dy = 0) is therefore tested with the
        // SAME rounded root, not with radius^2 >= d2: for radius == fl(sqrt(d2)) rounded down, fl(radius^2) < d2 although the voxel is kept
        // (chunks (-93,-80) / (-93,-79) of the seed-0 world: a ceiling sphere of radius fl(sqrt(34)) lost its eight rim voxels; found by
        // the full-world diff against the oracle's digests, round 6).  sqrt is monotone, so a column whose dy = 0 voxel fails has none.
        if (__builtin_sqrtf((float)d2) > radius) return false;
        const float rest = __builtin_fmaxf(radius * radius - (float)d2, 0.f);
        const int m = (int)__builtin_sqrtf(rest) + 1;              // |dy| <= sqrt(radius^2 (1 + 2^-22) - d2) < sqrt(rest) + 1
        const int c = feat == MMCF_STORMLIGHT_SPHERE ? 0 : lh;
        dlo = imax(dlo, c - m); dhi = imin(dhi, c + m);
        return true;
    }
    if (feat == MMCF_CRYSTAL_PILLAR && lh != 0) {                               // d <= 4 (2 (hr - 0.5)^2 + 0.5) <= 4, spherical caps of that radius at both ends
        if (d2 > 16) return false;                                              // (layerHeight 0: hr = 0 / 0, every comparison with the radius is false: table bounds)
        dlo = imax(dlo, -5); dhi = imin(dhi, lh + 5);
        return true;
    }
    return true;
}

// Stable WAVE-wide compaction of the list entries that can reach ANY column of the unit (wx0 .. wx0 + APPLY_UNIT_W - 1, wz0 .. wz0 + APPLY_UNIT_H - 1)
// into records
//   .x = fx, .y = fz, .z = fy | feature << 9 | canReplace << 14 | layerHeight << 15 | reach << 24, .w = the placement's stream right after seeding
// list order kept: 64 entries per round (filter_round), ballot + popcount prefix, no workgroup barrier.  The chunk's list is read from
// global memory ONCE per unit (round 2 read it once per column: eight dependent memory round trips per column at 3 waves per SIMD
// were a quarter of the kernel), the placements are seeded once per unit.
// kFeatureReach / kCaveFeatureReach as immediates (8 bits per feature): a table lookup indexed per lane is one more dependent memory
// round trip in a chain that is nothing but round trips
template <bool CAVE>
MM_DEV int reach_of(int feat)
{
    constexpr int N = CAVE ? MMGEN_NUM_CAVE_FEATURES : MMGEN_NUM_FEATURES;
    static_assert(N <= 32, "four 64-bit words");
    unsigned long long w[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
    for (int f = 0; f < N; ++f) w[f >> 3] |= (unsigned long long)(CAVE ? kCaveFeatureReach[f] : kFeatureReach[f]) << (8 * (f & 7));
    const unsigned long long v = feat < 16 ? (feat < 8 ? w[0] : w[1]) : (feat < 24 ? w[2] : w[3]);
    return (int)((v >> (8 * (feat & 7))) & 255ull);
}

// One round of filter_unit's walk: entries r0 .. r0 + 63 of `list`; the placements that can reach the unit are appended at s_unit[base ..]
// (the caller keeps 64 slots free).  Returns the number appended; ended = the list ends inside this round (first NONE, or its capacity).
template <class Entry, int LIST_CAP, bool CAVE, int UW, int UH>
MM_DEV int filter_round(const Entry* __restrict__ list, int r0, int wx0, int wz0, int4* s_unit, int base, bool& ended)
{
    static_assert(LIST_CAP % 64 == 0, "whole rounds");
    const int lane = threadIdx.x & 63;
    const Entry en = list[r0 + lane];
    const int feat = en.feature, fx = en.pos[0], fy = en.pos[1], fz = en.pos[2], canReplace = en.can_replace_blocks != 0;
    int lh = 0;
    if constexpr (CAVE) lh = en.layer_height;
    const unsigned long long noneMask = __ballot(feat == 0);                // the lists end at the first NONE
    const int firstNone = noneMask ? (int)__builtin_ctzll(noneMask) : 64;
    const int reach = reach_of<CAVE>(feat);
    const bool cand = lane < firstNone && wx0 - fx <= reach && fx - (wx0 + UW - 1) <= reach && wz0 - fz <= reach && fz - (wz0 + UH - 1) <= reach;
    const unsigned long long cm = __ballot(cand);
    if (cand)
        s_unit[base + __popcll(cm & ((1ull << lane) - 1ull))] =
            make_int4(fx, fz, (fy & 511) | (feat << 9) | (canReplace << 14) | (lh << 15) | (reach << 24),
                      (int)(CAVE ? cave_feature_stream(fx, fy, fz) : surface_feature_stream(fx, fy, fz)));
    ended = firstNone < 64 || r0 + 64 >= LIST_CAP;
    return __popcll(cm);
}

#ifndef MM_APPLY_WAVES
#define MM_APPLY_WAVES 4        // waves per SIMD = persistent workgroups per CU.  Left alone the union of the 31 rasterisers takes 155 VGPRs (3 waves);
                                // held to 128 the compiler still needs no scratch, 4 x 40 688 B of LDS just fit a CU (the pair buffer holds 192 entries for that), and a unit's chain of
                                // dependent round trips has a third more waves to hide behind: 1.36 -> 1.25 ms (round 4)
#endif
// Persistent workgroups of four independent WAVES; a wave takes one UNIT (APPLY_UNIT_W x APPLY_UNIT_H columns of a chunk) at a time and
// STREAMS it through three bounded per-wave LDS buffers (one loop, every step's code exists once):
//   A. placements: the chunk's (already chunk-prefiltered) lists are walked 64 entries at a time; the placements that can reach the unit
//      gather in `unit` (list order, surface list first).  When another round might not fit, or the lists are at their end:
//   B. pairs: every (placement, column) pair of the gathered placements gets the vertical extent the placement can claim in that column
//      (surface_extent / cave_extent, clipped to the chunk's bounds); the non-empty ones are appended PLACEMENT-MAJOR to `ent` with an
//      exclusive scan over their voxel counts, which gives every (placement, column, y) triple an item number.  When another round of
//      64 pairs might not fit, or the pairs are at their end:
//   C. items: walked 64 at a time, every lane evaluates ONE (voxel, placement) pair.  Placement-major order over several columns keeps
//      most of a batch on one placement - one rasteriser, one geometry; per column (round 2) a batch was the ~20 items of one column from
//      three or four different rasterisers, executed one after the other.  "First match in list order wins" (chunk.cu:1438-1509): the
//      items ascend in list order - inside a batch, from batch to batch, and from one flush of the buffers to the next - so the
//      placements of a batch that hit something write their voxels one placement after the other, each only where nothing was
//      written before.  One byte per voxel, no atomics.
//   D. the claimed voxels are written back.
// On generated terrain a unit is one A, one B and one C step; lists of any length (the 2 048 / 4 096 entries of the reference) only mean
// more flushes.  No workgroup barrier after the noise tables are in LDS.
// UW x UH = the unit's shape: 4 x 2 columns for launches that give every wave dozens of units (the optimum of round 3's sweep); 2 x 1 for
// small ones (k_apply_features_small: a streaming strip's 1 120 units of 4 x 2 are one round for 4 096 waves, as long as its slowest unit -
// a jungle unit's items are two thirds of that), where four times the units of a quarter of the columns spread the items over the idle waves.
template <int UW, int UH>
MM_DEV void apply_features_body(uint8_t* __restrict__ blocks, const int2* __restrict__ chunkPos, const mmgen_feature_placement* __restrict__ gfp,
                 const mmgen_cave_feature_placement* __restrict__ gcfp, const int* __restrict__ bounds, const int* __restrict__ srcIdx, int nUnits,
                 unsigned* __restrict__ nextUnit)
{
    __shared__ int4 s_unit[APPLY_COLS][APPLY_UNIT_CAP];    // per wave: the gathered placements, surface then cave
    __shared__ unsigned s_ent[APPLY_COLS][APPLY_ENT_CAP];  // per wave: placement | column << 7 | lowest y << 11 | (voxels - 1) << 20
    __shared__ unsigned short s_pref[APPLY_COLS][APPLY_ENT_CAP + 2];      // exclusive prefix of the pairs' voxel counts (+ the total)
    __shared__ __attribute__((aligned(16))) uint8_t s_claim[APPLY_COLS][(UW * UH) * 384];      // per voxel: the block of the first placement that claimed it, or 255
    __shared__ unsigned s_air[APPLY_COLS][(UW * UH) * 384 / 32];      // per voxel: the base block is AIR (all the item test needs of it)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // wave index in an SGPR: so are chunk and column
    // the simplex tables (14 KB) are staged ONCE per workgroup (most units of a generated world have a coral, a fungus or a redwood in reach)
    noise_tables_init();
    int4* unit = s_unit[wave];
    unsigned* ent = s_ent[wave];
    unsigned short* pref = s_pref[wave];
    uint8_t* claim = s_claim[wave];
    unsigned* air = s_air[wave];
    constexpr int ROW_WORDS = UW * 384 / 4;
    static_assert(64 <= APPLY_UNIT_CAP && 64 <= APPLY_ENT_CAP && 64 * 384 <= APPLY_ITEM_CAP && 64 % (UW * UH) == 0, "a round of 64 always fits an empty buffer, and holds whole placements");
    // Units cost anything between nothing (ocean) and ~100 us (jungle) and a wave only gets a few dozen: a fixed assignment leaves most
    // waves idle while the unluckiest finishes (measured: 5.2 ms instead of 3.0).  The waves draw their units from counters instead;
    // the next draw is in flight while the current unit is worked on.  ONE counter serialises at ~11 ns per draw in L2 (measured: 1.5 ms
    // for the 131 072 units of the bench tile, whatever else the kernel did): APPLY_COUNTERS of them, a cache line apart, counter p
    // hands out the units u = p (mod APPLY_COUNTERS); a wave moves on to the next counter when its own has run dry.
    int part = (APPLY_COLS * blockIdx.x + wave) % APPLY_COUNTERS, dry = 0;
    (void)dry;
    unsigned drawn = 0u;
    if (lane == 0) drawn = atomicAdd(&nextUnit[16 * part], 1u);
  for (;;) {
    const int u = __builtin_amdgcn_readfirstlane((int)drawn) * APPLY_COUNTERS + part;
    if (u >= nUnits) {
#if MM_COUNTER_PROBE_LOADS
        const int nx = next_live_counter(nextUnit, APPLY_COUNTERS, part, nUnits);
        if (nx < 0) break;
        part = nx;
#else
        if (++dry == APPLY_COUNTERS) break;
        part = (part + 1) % APPLY_COUNTERS;
#endif
    }
    if (lane == 0) drawn = atomicAdd(&nextUnit[16 * part], 1u);
    if (u >= nUnits) continue;
    const int chunk = u / ((16 / UW) * (16 / UH)), uu = u % ((16 / UW) * (16 / UH));      // dense output / list index; positions are read at srcIdx[chunk]
    const int x0 = UW * (uu % (16 / UW)), z0 = UH * (uu / (16 / UW));
    const mmgen_feature_placement* listS = gfp + (size_t)MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK * chunk;
    const mmgen_cave_feature_placement* listC = gcfp + (size_t)MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK * chunk;
    const int2 cp = chunkPos[srcIdx ? srcIdx[chunk] : chunk];
    const int b0 = bounds[4 * chunk], b1 = bounds[4 * chunk + 1], b2 = bounds[4 * chunk + 2], b3 = bounds[4 * chunk + 3];
    const bool doS = gfp && b0 <= b1, doC = gcfp && b2 <= b3;
    if (!doS && !doC) continue;
    const int wx0 = cp.x + x0, wz0 = cp.y + z0;
    const int sLo = imax(b0, 0), sHi = imin(b1, 383), cLo = imax(b2, 0), cHi = imin(b3, 383);   // the chunk's height bounds (chunk.cu:1555-1570)
    uint8_t* unitBlocks = blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * chunk + 384 * (16 * z0 + x0);      // row cz of the unit: + 384 * 16 * cz

    int phase = doS ? 0 : 1, r0 = 0;                        // list being walked (0 surface, 1 cave, 2 = both at their end) and the next round
    int nU = 0, nUS = 0;                                    // gathered placements, and how many of them are surface placements
    bool staged = false;                                    // the unit's air bits are in LDS, the claims cleared
    for (;;) {
        // ---- A. gather placements until another round might not fit or the lists end
        while (phase < 2 && nU + 64 <= APPLY_UNIT_CAP) {
            bool ended;
            if (phase == 0) {
                const int n = filter_round<mmgen_feature_placement, MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK, false, UW, UH>(listS, r0, wx0, wz0, unit, nU, ended);
                nU += n; nUS += n;
            } else {
                nU += filter_round<mmgen_cave_feature_placement, MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK, true, UW, UH>(listC, r0, wx0, wz0, unit, nU, ended);
            }
            r0 += 64;
            if (ended) { r0 = 0; phase = (phase == 0 && doC) ? 1 : 2; }
        }
        if (nU == 0) break;                                 // (only when the lists are at their end)
        wave_lds_sync();

        // ---- B + C. pairs of the gathered placements, items whenever the pair buffer fills up and at the end
        const int nPairs = nU * (UW * UH);
        int nEnt = 0, total = 0;
        for (int p0 = 0;;) {
            const bool pairsDone = p0 >= nPairs;
            if (pairsDone || nEnt + 64 > APPLY_ENT_CAP || total + 64 * 384 > APPLY_ITEM_CAP) {
                if (total > 0) {
                    if (!staged) {
                        // the unit's air bits (UW x 384 contiguous bytes per row; one mask word = 32 voxels = two 16-byte loads), claims cleared
                        static_assert(MMB_AIR == 0, "zero-byte test below");
#pragma unroll
                        for (int cz = 0; cz < UH; ++cz) {
                            for (int m = lane; m < ROW_WORDS / 8; m += 64) {
                                const uint4* src = (const uint4*)(unitBlocks + 384 * 16 * cz) + 2 * m;
                                unsigned bits = 0u;
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const uint4 a = src[h];
                                    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                                    for (int q = 0; q < 4; ++q) {
                                        const uint32_t z = ~(((w[q] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w[q] | 0x7f7f7f7fu);      // 0x80 in every byte that is 0
                                        bits |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (16 * h + 4 * q);
                                    }
                                }
                                air[(ROW_WORDS / 8) * cz + m] = bits;
                            }
                            for (int i = lane; i < ROW_WORDS; i += 64) ((uint32_t*)claim)[ROW_WORDS * cz + i] = 0xffffffffu;
                        }
                        staged = true;
                    }
                    if (lane == 0) pref[nEnt] = (unsigned short)total;
                    wave_lds_sync();
                    for (int j0 = 0; j0 < total; j0 += 64) {
                        const int j = j0 + lane;
                        bool placed = false;
                        int k = 0, v = 0;
                        uint8_t fb = 0;
                        if (j < total) {
                            int e = 0, eh = nEnt;                       // pref[e] <= j < pref[eh]
                            while (eh - e > 1) { const int mid = (e + eh) >> 1; if ((int)pref[mid] <= j) e = mid; else eh = mid; }
                            const unsigned en = ent[e];
                            k = en & 127;
                            const int c = (en >> 7) & 15, y = (int)((en >> 11) & 511) + (j - (int)pref[e]);
                            v = 384 * c + y;
                            const int4 rc = unit[k];
                            if (((air[v >> 5] >> (v & 31)) & 1u) || ((rc.z >> 14) & 1)) {
                                const int fy = rc.z & 511, feature = (rc.z >> 9) & 31, wx = wx0 + c % UW, wz = wz0 + c / UW;
                                placed = k >= nUS ? place_cave_feature(feature, rc.x, fy, rc.y, (rc.z >> 15) & 511, wx, y, wz, (uint32_t)rc.w, fb)
                                                  : place_feature(feature, rc.x, fy, rc.y, wx, y, wz, (uint32_t)rc.w, fb);
                            }
                        }
                        // first match in list order wins: the batch's placements in ascending order, one masked write each (a placement has one item per voxel)
                        unsigned long long todo = __ballot(placed);
                        while (todo) {
                            const int kk = __shfl(k, (int)__builtin_ctzll(todo));
                            const bool mine = placed && k == kk;
                            if (mine && claim[v] == 255) claim[v] = fb;
                            todo &= ~__ballot(mine);
                            wave_lds_sync();
                        }
                    }
                    wave_lds_sync();                               // ent / pref are refilled
                }
                nEnt = 0; total = 0;
                if (pairsDone) break;
            }
            // one round of 64 (placement, column) pairs = whole placements, placement-major
            const int p = p0 + lane, k = p / (UW * UH), c = p % (UW * UH);
            int n = 0, lo = 0;
            if (p < nPairs) {
                const int4 rc = unit[k];
                const int fx = rc.x, fz = rc.y, fy = rc.z & 511, feat = (rc.z >> 9) & 31, reach = rc.z >> 24;
                const int wx = wx0 + c % UW, wz = wz0 + c / UW;
                if (iabs(wx - fx) <= reach && iabs(wz - fz) <= reach) {
                    const bool cave = k >= nUS;
                    const bool noiseBounds = prune_domain(wx, wz);      // the two extents that use a simplex bound (coral ellipsoids, glowstone)
                    int dlo, dhi;
                    const bool any = cave ? cave_extent(feat, (rc.z >> 15) & 511, wx - fx, wz - fz, (uint32_t)rc.w, noiseBounds, dlo, dhi)
                                          : surface_extent(feat, fy, wx - fx, wz - fz, (uint32_t)rc.w, noiseBounds, dlo, dhi);
                    if (any) {
                        lo = imax(fy + dlo, cave ? cLo : sLo);
                        n = imax(imin(fy + dhi, cave ? cHi : sHi) - lo + 1, 0);
                    }
                }
            }
            p0 += 64;
            const unsigned long long vm = __ballot(n > 0);
            if (vm == 0ull) continue;
            int incl = n;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
            if (n > 0) {
                const int slot = nEnt + __popcll(vm & ((1ull << lane) - 1ull));
                ent[slot] = (unsigned)k | (c << 7) | (lo << 11) | ((n - 1) << 20);
                pref[slot] = (unsigned short)(total + incl - n);
            }
            nEnt += __popcll(vm);
            total += __builtin_amdgcn_readfirstlane(__shfl(incl, 63));
        }
        nU = 0; nUS = 0;                                    // the placement buffer is free again
        if (phase == 2) break;
    }
    if (!staged) continue;
    // ---- D. the claimed voxels back to the chunk
    wave_lds_sync();
#pragma unroll
    for (int cz = 0; cz < UH; ++cz)
        for (int i = lane; i < ROW_WORDS; i += 64) {
            const uint32_t cl = ((const uint32_t*)claim)[ROW_WORDS * cz + i];
            if (cl != 0xffffffffu) {
                const uint32_t old = ((const uint32_t*)(unitBlocks + 384 * 16 * cz))[i];
                uint32_t out = 0u;
#pragma unroll
                for (int b = 0; b < 4; ++b) { const uint32_t cb = (cl >> (8 * b)) & 255u; out |= (cb != 255u ? cb : (old >> (8 * b)) & 255u) << (8 * b); }
                ((uint32_t*)(unitBlocks + 384 * 16 * cz))[i] = out;
            }
        }
    wave_lds_sync();                                       // the wave's LDS lists are re-used by its next unit
  }
}

__attribute__((amdgpu_waves_per_eu(MM_APPLY_WAVES, MM_APPLY_WAVES)))
__global__ void __launch_bounds__(APPLY_THREADS)
k_apply_features(uint8_t* __restrict__ blocks, const int2* __restrict__ chunkPos, const mmgen_feature_placement* __restrict__ gfp,
                 const mmgen_cave_feature_placement* __restrict__ gcfp, const int* __restrict__ bounds, const int* __restrict__ srcIdx, int nUnits,
                 unsigned* __restrict__ nextUnit)
{
    apply_features_body<APPLY_UNIT_W, APPLY_UNIT_H>(blocks, chunkPos, gfp, gcfp, bounds, srcIdx, nUnits, nextUnit);
}

#define APPLY_SMALL_W 2
#define APPLY_SMALL_H 1
#ifndef APPLY_SMALL_MAX_CHUNKS
#define APPLY_SMALL_MAX_CHUNKS 256                  // launches of at most this many chunks take the 2 x 1 units (256 chunks x 128 units = 8 per wave)
#endif
__attribute__((amdgpu_waves_per_eu(MM_APPLY_WAVES, MM_APPLY_WAVES)))
__global__ void __launch_bounds__(APPLY_THREADS)
k_apply_features_small(uint8_t* __restrict__ blocks, const int2* __restrict__ chunkPos, const mmgen_feature_placement* __restrict__ gfp,
                       const mmgen_cave_feature_placement* __restrict__ gcfp, const int* __restrict__ bounds, const int* __restrict__ srcIdx, int nUnits,
                       unsigned* __restrict__ nextUnit)
{
    apply_features_body<APPLY_SMALL_W, APPLY_SMALL_H>(blocks, chunkPos, gfp, gcfp, bounds, srcIdx, nUnits, nextUnit);
}

// ---------------------------------------------------------------------------------------------------------
// D1 — decorators (chunk.cu:1634-1747).  The reference consumes ONE minstd stream per chunk sequentially over the columns:
// 2 draws per column + 2 per used cave layer.  minstd has no increment, so x_{n+k} = a^k x_n mod m: every lane jumps the
// stream to its column's first draw and all 256 columns run in parallel (decorators only touch their own column).
// ---------------------------------------------------------------------------------------------------------
// a * b mod 2^31 - 1 for a, b below the modulus: 2^31 = 1 (mod m), so the 62-bit product folds twice (no 64-bit division)
MM_DEV uint32_t mulmod(uint32_t a, uint32_t b)
{
    const uint64_t p = (uint64_t)a * b;
    uint64_t r = (p & 0x7fffffffull) + (p >> 31);             // < 2^32
    r = (r & 0x7fffffffull) + (r >> 31);                      // <= 2^31
    return (uint32_t)(r >= 0x7fffffffull ? r - 0x7fffffffull : r);
}

MM_DEV void try_place_decorator(uint8_t* col, int y, const DecoGen& g)     // tryPlaceSingleDecorator chunk.cu:1634-1677
{
    if (y < 0 || y > 383) return;       // canonical: out-of-column positions are no-ops (open-to-sky ceiling decorators, y == 384)
    const uint8_t cur = col[y];
    if (cur != g.replace) return;
    const int uo = g.fromCeiling ? 1 : -1;
    if (y + uo < 0 || y + uo > 383) return;
    const uint8_t under = col[y + uo];
    if (under < MMB_NUM_NON_SOLID_BLOCKS) return;
    if (g.nUnder > 0) {
        bool ok = false;
        for (int i = 0; i < g.nUnder; ++i) ok = ok || under == g.under[i];
        if (!ok) return;
    }
    if (g.second != MMB_AIR) {
        const int oo = -uo;
        if (y + oo < 0 || y + oo > 383) return;
        if (col[y + oo] != g.replace) return;
        col[y + oo] = g.second;
    }
    col[y] = g.block;
}

__global__ void __launch_bounds__(256)
k_decorators(uint8_t* __restrict__ blocks, const float* __restrict__ hf, const float* __restrict__ bw,
             const mmgen_cave_layer* __restrict__ caveLayers, const int2* __restrict__ chunkPos, const int* __restrict__ srcIdx)
{
    // (no simplex noise in this kernel: the tables stay where they are)
    __shared__ int s_draws[4];
    const int outChunk = blockIdx.x, t = threadIdx.x;
    const int chunk = srcIdx ? srcIdx[outChunk] : outChunk;
    const mmgen_cave_layer* ccl = caveLayers + ((size_t)256 * chunk + t) * MMGEN_MAX_CAVE_LAYERS_PER_COLUMN;
    // the used slots (the first unused one starts at 384), four loads in flight per round trip instead of one
    int used = MMGEN_MAX_CAVE_LAYERS_PER_COLUMN;
#pragma unroll 1
    for (int k0 = 0; k0 < MMGEN_MAX_CAVE_LAYERS_PER_COLUMN; k0 += 4) {
        const int s0 = ccl[k0].start, s1 = ccl[k0 + 1].start, s2 = ccl[k0 + 2].start, s3 = ccl[k0 + 3].start;
        const int first = s0 == 384 ? 0 : (s1 == 384 ? 1 : (s2 == 384 ? 2 : (s3 == 384 ? 3 : 4)));
        if (first < 4) { used = k0 + first; break; }
    }
    // draws consumed by the columns before this one: a shuffle scan inside each wave, the four wave totals through LDS
    const int mine = 2 + 2 * used;
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if ((t & 63) >= d) incl += up; }
    if ((t & 63) == 63) s_draws[t >> 6] = incl;
    __syncthreads();
    int skip = incl - mine;
    for (int w = 0; w < (t >> 6); ++w) skip += s_draws[w];

    const int2 cp = chunkPos[chunk];
    MinStd rng = rng4(cp.x, 0, cp.y, 7589341);
    // jump ahead by `skip` draws: x <- a^skip * x mod m
    uint32_t mult = 1u, base = 48271u;
    for (int e = skip; e > 0; e >>= 1) { if (e & 1) mult = mulmod(mult, base); base = mulmod(base, base); }
    rng.x = mulmod(rng.x, mult);

    uint8_t* col = blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * outChunk + 384 * t;
    const float* cbw = bw + (size_t)MMGEN_BIOME_WEIGHTS_SIZE * chunk + t;
    int biome = MMBIO_PLAINS;
    {
        // the 24 weights in one round trip instead of one dependent load per step of random_biome's walk (mm_biome.cuh)
        float w24[MMGEN_NUM_BIOMES];
#pragma unroll
        for (int b = 0; b < MMGEN_NUM_BIOMES; ++b) w24[b] = cbw[256 * b];
        float r = rng.u01();
        bool found = false;
#pragma unroll
        for (int b = 0; b < MMGEN_NUM_BIOMES; ++b) { r -= w24[b]; if (!found && r <= 0.f) { biome = b; found = true; } }
    }
    float rand = rng.u01();
    for (int g = 0; g < kDecoCount[biome]; ++g) {
        const DecoGen& gen = kDecoGens[biome][g];
        if ((rand -= gen.chance) < 0.f) {
            try_place_decorator(col, (int)hf[(size_t)256 * chunk + t] + 1, gen);
            break;
        }
    }
    for (int k = 0; k < used; ++k) {
        float bottomRand = rng.u01();
        float topRand = rng.u01();
        const int cb = ccl[k].bottom_biome;
        // placedBottom / placedTop are never set in the reference: every gen whose cumulative chance is passed fires
        for (int g = 0; g < kCaveDecoCount[cb]; ++g) {
            const DecoGen& gen = kCaveDecoGens[cb][g];
            if (gen.fromCeiling) { if ((topRand -= gen.chance) < 0.f) try_place_decorator(col, ccl[k].end, gen); }
            else { if ((bottomRand -= gen.chance) < 0.f) try_place_decorator(col, ccl[k].start + 1, gen); }
        }
    }
}

// Test probe: rasterise ONE placement into a box (255 = not claimed), order z, x, y (y fastest).
__global__ void __launch_bounds__(256)
k_feature_box(int isCave, int feature, int fx, int fy, int fz, int layerHeight, int bx, int by, int bz, int sx, int sy, int sz, uint8_t* __restrict__ out)
{
    noise_tables_init();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= sx * sy * sz) return;
    const int y = i % sy, x = (i / sy) % sx, z = i / (sy * sx);
    uint8_t b = 0;
    const bool placed = isCave ? place_cave_feature(feature, fx, fy, fz, layerHeight, bx + x, by + y, bz + z, cave_feature_stream(fx, fy, fz), b)
                               : place_feature(feature, fx, fy, fz, bx + x, by + y, bz + z, surface_feature_stream(fx, fy, fz), b);
    out[i] = placed ? b : 255;
}

// Test probe: the constant rule tables of this library in the numeric layout of tools/extract_ref_tables.py (tests hold them to the
// literals of the reference's BiomeUtils::init).  Sections, all as floats: biome rules [24][6], grass [24], material infos [20][4],
// biome material weights [24][20], feature bounds [21][2], cave feature bounds [10][2], surface gens [24][4][11], cave gens [5][3][9],
// decorator gens [24][7][10], cave decorator gens [5][6][10]; then this library's own horizontal reach tables [21] + [10] (not a reference
// table: the bound the column filters rely on, validated by tests) and the gather order [49][2].
#define MMGEN_TABLE_DUMP_FLOATS (144 + 24 + 80 + 480 + 42 + 20 + 1056 + 135 + 1680 + 300 + 21 + 10 + 98)
__device__ void dump_deco(const DecoGen& g, float* o)
{
    int u[3] = {g.nUnder > 0 ? g.under[0] : 0, g.nUnder > 1 ? g.under[1] : 0, g.nUnder > 2 ? g.under[2] : 0};
    // ascending over the first nUnder entries (the reference keeps them in an unordered_set)
    for (int a = 0; a < g.nUnder; ++a) for (int b = a + 1; b < g.nUnder; ++b) if (u[b] < u[a]) { const int t = u[a]; u[a] = u[b]; u[b] = t; }
    o[0] = 1.f; o[1] = (float)g.block; o[2] = g.chance; o[3] = (float)g.nUnder; o[4] = (float)u[0]; o[5] = (float)u[1]; o[6] = (float)u[2];
    o[7] = (float)g.replace; o[8] = (float)g.second; o[9] = (float)g.fromCeiling;
}
__global__ void k_dump_tables(float* __restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < MMGEN_TABLE_DUMP_FLOATS; ++i) out[i] = 0.f;
    float* o = out;
    for (int b = 0; b < MMGEN_NUM_BIOMES; ++b) for (int k = 0; k < 6; ++k) *o++ = (float)kBiomeRules[b][k];
    for (int b = 0; b < MMGEN_NUM_BIOMES; ++b) *o++ = (float)kGrassBlock[b];
    for (int m = 0; m < MMGEN_NUM_MATERIALS; ++m) { *o++ = (float)kMaterialBlock[m]; *o++ = kMaterialThickness[m]; *o++ = kMaterialAmpOrTan[m]; *o++ = kMaterialScaleOrMaxSlope[m]; }
    for (int b = 0; b < MMGEN_NUM_BIOMES; ++b) for (int m = 0; m < MMGEN_NUM_MATERIALS; ++m) *o++ = kMatWeights.w[b][m];
    for (int f = 0; f < MMGEN_NUM_FEATURES; ++f) { *o++ = (float)kFeatureBounds[f][0]; *o++ = (float)kFeatureBounds[f][1]; }
    for (int f = 0; f < MMGEN_NUM_CAVE_FEATURES; ++f) { *o++ = (float)kCaveFeatureBounds[f][0]; *o++ = (float)kCaveFeatureBounds[f][1]; }
    for (int b = 0; b < MMGEN_NUM_BIOMES; ++b)
        for (int k = 0; k < 4; ++k, o += 11) {
            if (k >= kSurfGenCount[b]) continue;
            const SurfGen& g = kSurfGens[b][k];
            o[0] = 1.f; o[1] = (float)g.feature; o[2] = (float)g.cell; o[3] = (float)g.pad; o[4] = g.chance; o[5] = (float)g.canReplace; o[6] = (float)g.nTop;
            for (int j = 0; j < g.nTop; ++j) { o[7 + 2 * j] = (float)g.topMat[j]; o[8 + 2 * j] = g.topMin[j]; }
        }
    for (int b = 0; b < MMGEN_NUM_CAVE_BIOMES; ++b)
        for (int k = 0; k < 3; ++k, o += 9) {
            if (k >= kCaveGenCount[b]) continue;
            const CaveGen& g = kCaveGens[b][k];
            o[0] = 1.f; o[1] = (float)g.feature; o[2] = (float)g.cell; o[3] = (float)g.pad; o[4] = g.chance; o[5] = (float)g.minLayerHeight;
            o[6] = (float)g.canReplace; o[7] = (float)g.fromCeiling; o[8] = (float)g.canLava;
        }
    for (int b = 0; b < MMGEN_NUM_BIOMES; ++b) for (int k = 0; k < 7; ++k, o += 10) if (k < kDecoCount[b]) dump_deco(kDecoGens[b][k], o);
    for (int b = 0; b < MMGEN_NUM_CAVE_BIOMES; ++b) for (int k = 0; k < 6; ++k, o += 10) if (k < kCaveDecoCount[b]) dump_deco(kCaveDecoGens[b][k], o);
    for (int f = 0; f < MMGEN_NUM_FEATURES; ++f) *o++ = (float)kFeatureReach[f];
    for (int f = 0; f < MMGEN_NUM_CAVE_FEATURES; ++f) *o++ = (float)kCaveFeatureReach[f];
    for (int k = 0; k < 49; ++k) { *o++ = (float)kGatherDX[k]; *o++ = (float)kGatherDZ[k]; }
}

}  // namespace mm

namespace mmk {

#define LAUNCH(KID, KERNEL, GRID, BLOCK, STREAM, ...)                                        \
    do {                                                                                     \
        { const int ne_ = mm::noise_tables_ensure(STREAM); if (ne_) return ne_; }            \
        MMK_LAUNCH(KID, KERNEL, GRID, BLOCK, STREAM, __VA_ARGS__);                           \
    } while (0)

int prepare_features() { return mm::noise_tables_ensure(nullptr); }

int table_dump_floats() { return MMGEN_TABLE_DUMP_FLOATS; }
int launch_dump_tables(float* out, hipStream_t s)
{
    MMK_LAUNCH(KID_PROBE, mm::k_dump_tables, dim3(1), dim3(64), s, out);
    return 0;
}

int launch_feature_box(int isCave, int feature, const int* fpos, int layerHeight, const int* boxMin, const int* boxSize, uint8_t* out, hipStream_t s)
{
    const int n = boxSize[0] * boxSize[1] * boxSize[2];
    if (n <= 0) return 0;
    LAUNCH(KID_FEATURE_BOX, mm::k_feature_box, dim3((n + 255) / 256), dim3(256), s, isCave, feature, fpos[0], fpos[1], fpos[2], layerHeight,
           boxMin[0], boxMin[1], boxMin[2], boxSize[0], boxSize[1], boxSize[2], out);
    return 0;
}

int launch_feature_placements(const float* hf, const float* bw, const float* layers, const mmgen_cave_layer* cl, const int32_t* pos, int n,
                              mmgen_feature_placement* fp, mmgen_cave_feature_placement* cfp, int* counts, const int* chunkList, const uint8_t* colNeed,
                              hipStream_t s)
{
    if (n <= 0) return 0;
    LAUNCH(KID_FEATURE_PLACEMENTS, mm::k_feature_placements, dim3(n), dim3(256), s, hf, bw, layers, cl, (const int2*)pos, fp, cfp, counts, chunkList, colNeed);
    return 0;
}

int launch_ring_need(const float* bw, const int32_t* pos, const int* chunkList, int n, const uint8_t* cellLazy, int rx0, int rz0, int rx1, int rz1,
                     uint8_t* colNeed, hipStream_t s)
{
    if (n <= 0) return 0;
    LAUNCH(KID_RING_NEED, mm::k_ring_need, dim3(n), dim3(256), s, bw, (const int2*)pos, chunkList, cellLazy, rx0, rz0, rx1, rz1, colNeed);
    return 0;
}

int launch_gather_placements(const mmgen_feature_placement* fp, const mmgen_cave_feature_placement* cfp, const int* counts, const int* target,
                             int nOut, int gridW, int gridH, mmgen_feature_placement* gfp, mmgen_cave_feature_placement* gcfp, int* bounds,
                             const int32_t* gridPos, hipStream_t s, int* maxGathered, int* capHost, int* capMax, unsigned* zeroWords, int nZeroWords,
                             const float* hfGrid, float* hfOut)
{
    if (nOut <= 0) return 0;
    LAUNCH(KID_GATHER_PLACEMENTS, mm::k_gather_placements, dim3(nOut), dim3(256), s, fp, cfp, counts, target, gridW, gridH, gfp, gcfp, bounds,
           (const int2*)gridPos, maxGathered, capHost, capMax, zeroWords, nZeroWords, hfGrid, hfOut);
    return 0;
}

size_t apply_work_bytes() { return 64 * APPLY_COUNTERS; }

int launch_apply_features(uint8_t* blocks, const int32_t* pos, int n, const mmgen_feature_placement* gfp, const mmgen_cave_feature_placement* gcfp,
                          const int* bounds, const int* srcIdx, unsigned* workCounter, hipStream_t s, bool workCleared)
{
    if (n <= 0) return 0;
    if (!workCounter) return (int)hipErrorInvalidValue;
    const int cus = device_cus();
    if (!cus) return (int)hipErrorInvalidDevice;
    // persistent: MM_APPLY_WAVES waves per SIMD = that many 4-wave workgroups per CU; every wave walks its own units
    static const int smallMax = [] { const char* e = getenv("MMGEN_APPLY_SMALL_MAX_CHUNKS"); return e ? atoi(e) : APPLY_SMALL_MAX_CHUNKS; }();      // (A/B: 0 = never)
    const bool small = n <= smallMax;
    const long long units = (long long)n * (small ? (16 / APPLY_SMALL_W) * (16 / APPLY_SMALL_H) : APPLY_UNITS_PER_CHUNK), groups = (units + APPLY_COLS - 1) / APPLY_COLS,
                    fit = (long long)cus * MM_APPLY_WAVES;
    if (units > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if (!workCleared) {            // (the region's gather kernel clears them on its way)
        const hipError_t e = hipMemsetAsync(workCounter, 0, apply_work_bytes(), s);
        if (e != hipSuccess) return (int)e;
    }
    if (small)
        LAUNCH(KID_APPLY_FEATURES, mm::k_apply_features_small, dim3((unsigned)(groups < fit ? groups : fit)), dim3(APPLY_THREADS), s, blocks, (const int2*)pos, gfp, gcfp,
               bounds, srcIdx, (int)units, workCounter);
    else
        LAUNCH(KID_APPLY_FEATURES, mm::k_apply_features, dim3((unsigned)(groups < fit ? groups : fit)), dim3(APPLY_THREADS), s, blocks, (const int2*)pos, gfp, gcfp,
               bounds, srcIdx, (int)units, workCounter);
    return 0;
}

int launch_decorators(uint8_t* blocks, const float* hf, const float* bw, const mmgen_cave_layer* cl, const int32_t* pos, int n, const int* srcIdx, hipStream_t s)
{
    if (n <= 0) return 0;
    LAUNCH(KID_DECORATORS, mm::k_decorators, dim3(n), dim3(256), s, blocks, hf, bw, cl, (const int2*)pos, srcIdx);
    return 0;
}

}  // namespace mmk
