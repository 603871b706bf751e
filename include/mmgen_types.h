/* mmgen — wire types of the chunk-generation path (C ABI, no C++ / torch types).
 *
 * These PODs and enumerations are the layout contract between stages and with the caller.  They replace, value for
 * value and byte for byte, the reference's
 *   enum Block              src/terrain/block.hpp:5-154      (uint8 block ids written to Chunk::blocks)
 *   enum Biome/CaveBiome/Material/Feature/CaveFeature        src/terrain/biome.hpp:13-167
 *   struct CaveLayer (12 B)                                   src/terrain/biome.hpp:106-115
 *   struct FeaturePlacement (20 B), CaveFeaturePlacement (24 B)  src/terrain/biome.hpp:195-245
 * and the staging sizes of src/terrain/terrain.hpp:17-50.
 */
#ifndef MMGEN_TYPES_H
#define MMGEN_TYPES_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMGEN_CHUNK_COLS 256                 /* 16 x 16 columns, idx2d = x + 16 z          (biomeFuncs.hpp:11-23) */
#define MMGEN_CHUNK_HEIGHT 384
#define MMGEN_BLOCKS_PER_CHUNK 98304         /* idx = y + 384 (x + 16 z)                    (terrain.hpp:37)       */
#define MMGEN_NUM_BIOMES 24
#define MMGEN_NUM_CAVE_BIOMES 5
#define MMGEN_NUM_MATERIALS 20
#define MMGEN_NUM_FORWARD_MATERIALS 10
#define MMGEN_NUM_STRATIFIED_MATERIALS 12
#define MMGEN_NUM_ERODED_MATERIALS 8
#define MMGEN_NUM_FEATURES 21
#define MMGEN_NUM_CAVE_FEATURES 10
#define MMGEN_HEIGHTFIELD_SIZE 256
#define MMGEN_GATHERED_HEIGHTFIELD_SIZE 324  /* 18 x 18, devHeightfieldSize                 (terrain.hpp:41)       */
#define MMGEN_BIOME_WEIGHTS_SIZE 6144        /* devBiomeWeightsSize = 256 * 24, biome-major (terrain.hpp:42)       */
#define MMGEN_LAYERS_SIZE 5120               /* devLayersSize = 256 * 20, layer-major       (terrain.hpp:44)       */
#define MMGEN_MAX_CAVE_LAYERS_PER_COLUMN 32  /*                                              (biome.hpp:6)          */
#define MMGEN_CAVE_LAYERS_SIZE 8192          /* devCaveLayersSize = 256 * 32, column-major  (terrain.hpp:45)       */
#define MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK 2048        /* biome.hpp:7 */
#define MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK 4096   /* biome.hpp:8 */
#define MMGEN_FP_CAP 256                     /* per-chunk surface placements: at most one per column (chunk.cu:1136-1141) */
#define MMGEN_CFP_CAP 1024                   /* per-chunk cave placements kept on device (typ. < 80, biome.hpp:8 comment)  */
#define MMGEN_SEA_LEVEL 128
#define MMGEN_LAVA_LEVEL 8
#define MMGEN_ZONE_SIZE 12                   /* chunks per zone side                         (terrain.hpp:17)       */
#define MMGEN_EROSION_GRID_SIDE 384          /* EROSION_GRID_SIDE_LENGTH_BLOCKS             (terrain.hpp:18)       */
#define MMGEN_EROSION_GRID_NUM_COLS 147456   /* EROSION_GRID_NUM_COLS                        (terrain.hpp:19)       */
#define MMGEN_GATHERED_LAYERS_SIZE 1327105   /* devGatheredLayersSize = 147456 * 9 + 1 flag (terrain.hpp:47)       */

typedef struct mmgen_cave_layer {            /* biome.hpp:106-115 */
    int32_t start;                           /* exclusive (is not air); 384 = unused slot */
    int32_t end;                             /* inclusive (is air); 384 = open to the sky */
    uint8_t bottom_biome;                    /* cave biome of block y = start   */
    uint8_t top_biome;                       /* cave biome of block y = end + 1 */
    uint8_t padding[2];
} mmgen_cave_layer;

typedef struct mmgen_vertex {                /* Vertex, src/rendering/structs.hpp:25-31 (vec3, vec3, vec2, Mats : size_t) */
    float pos[3];                            /* chunk-local block coordinates */
    float nor[3];
    float uv[2];                             /* texture-atlas coordinates, tile = 1/16 */
    uint64_t material;                       /* Mats: 0 DIFFUSE, 1 WATER, 2 CRYSTAL, 3 SMOOTH_MICRO, 4 MICRO, 5 ROUGH_MICRO */
} mmgen_vertex;

typedef struct mmgen_feature_placement {     /* biome.hpp:195-200: feature@0, pos@4, canReplaceBlocks@16 */
    uint8_t feature;
    uint8_t pad0[3];
    int32_t pos[3];
    uint8_t can_replace_blocks;
    uint8_t pad1[3];
} mmgen_feature_placement;

typedef struct mmgen_cave_feature_placement { /* biome.hpp:239-245: feature@0, pos@4, layerHeight@16, canReplaceBlocks@20 */
    uint8_t feature;
    uint8_t pad0[3];
    int32_t pos[3];                          /* lowest air block of the cave layer */
    int32_t layer_height;
    uint8_t can_replace_blocks;
    uint8_t pad1[3];
} mmgen_cave_feature_placement;

/* enum Block (block.hpp:5-154): the first 56 ids (AIR .. KELP_END) are non-solid */
enum {
    MMB_AIR, MMB_WATER, MMB_LAVA, MMB_CAVE_VINES_MAIN, MMB_CAVE_VINES_GLOW_MAIN, MMB_CAVE_VINES_END, MMB_CAVE_VINES_GLOW_END,
    MMB_GRASS, MMB_JUNGLE_GRASS, MMB_SAVANNA_GRASS, MMB_WARPED_MUSHROOM, MMB_WARPED_ROOTS, MMB_NETHER_SPROUTS,
    MMB_INFECTED_MUSHROOM, MMB_AMBER_ROOTS, MMB_DANDELION, MMB_POPPY, MMB_PITCHER_BOTTOM, MMB_PITCHER_TOP, MMB_CORNFLOWER,
    MMB_BLUE_ORCHID, MMB_ALLIUM, MMB_RED_TULIP, MMB_ORANGE_TULIP, MMB_WHITE_TULIP, MMB_PINK_TULIP, MMB_LILAC_BOTTOM,
    MMB_LILAC_TOP, MMB_PEONY_BOTTOM, MMB_PEONY_TOP, MMB_OXEYE_DAISY, MMB_LILY_OF_THE_VALLEY, MMB_JUNGLE_FERN,
    MMB_SMALL_MAGENTA_CRYSTAL, MMB_SMALL_CYAN_CRYSTAL, MMB_SMALL_GREEN_CRYSTAL, MMB_SMALL_PURPLE_MUSHROOM, MMB_DEAD_BUSH,
    MMB_HANGING_SMALL_MAGENTA_CRYSTAL, MMB_HANGING_SMALL_CYAN_CRYSTAL, MMB_HANGING_SMALL_GREEN_CRYSTAL, MMB_TALL_GRASS_BOTTOM,
    MMB_TALL_GRASS_TOP, MMB_TALL_JUNGLE_GRASS_BOTTOM, MMB_TALL_JUNGLE_GRASS_TOP, MMB_TORCHFLOWER, MMB_BRAIN_CORAL,
    MMB_BUBBLE_CORAL, MMB_FIRE_CORAL, MMB_HORN_CORAL, MMB_TUBE_CORAL, MMB_SEAGRASS, MMB_TALL_SEAGRASS_BOTTOM,
    MMB_TALL_SEAGRASS_TOP, MMB_KELP_MAIN, MMB_KELP_END,
    MMB_BEDROCK,
    MMB_STONE, MMB_DIRT, MMB_GRASS_BLOCK, MMB_SAND, MMB_GRAVEL, MMB_MYCELIUM, MMB_SNOW, MMB_SNOWY_GRASS_BLOCK, MMB_MUSHROOM_STEM,
    MMB_MUSHROOM_UNDERSIDE, MMB_PURPLE_MUSHROOM_CAP, MMB_MARBLE, MMB_ANDESITE, MMB_CALCITE, MMB_BLACKSTONE, MMB_TUFF,
    MMB_DEEPSLATE, MMB_GRANITE, MMB_SLATE, MMB_SANDSTONE, MMB_CLAY, MMB_RED_SAND, MMB_RED_SANDSTONE, MMB_MUD,
    MMB_JUNGLE_GRASS_BLOCK, MMB_RAFFLESIA_PETAL, MMB_RAFFLESIA_CENTER, MMB_RAFFLESIA_SPIKES, MMB_RAFFLESIA_STEM, MMB_JUNGLE_WOOD,
    MMB_JUNGLE_LEAVES_PLAIN, MMB_JUNGLE_LEAVES_FRUITS, MMB_CACTUS, MMB_PALM_WOOD, MMB_PALM_LEAVES, MMB_MAGENTA_CRYSTAL,
    MMB_CYAN_CRYSTAL, MMB_GREEN_CRYSTAL, MMB_SMOOTH_SAND, MMB_TERRACOTTA, MMB_YELLOW_TERRACOTTA, MMB_ORANGE_TERRACOTTA,
    MMB_PURPLE_TERRACOTTA, MMB_RED_TERRACOTTA, MMB_WHITE_TERRACOTTA, MMB_QUARTZ, MMB_ICE, MMB_PACKED_ICE, MMB_BLUE_ICE,
    MMB_SAVANNA_GRASS_BLOCK, MMB_BIRCH_WOOD, MMB_BIRCH_LEAVES, MMB_YELLOW_BIRCH_LEAVES, MMB_ORANGE_BIRCH_LEAVES, MMB_ACACIA_WOOD,
    MMB_ACACIA_LEAVES, MMB_SMOOTH_SANDSTONE, MMB_PINE_WOOD, MMB_PINE_LEAVES_1, MMB_PINE_LEAVES_2, MMB_REDWOOD_WOOD,
    MMB_REDWOOD_LEAVES, MMB_CYPRESS_WOOD, MMB_CYPRESS_LEAVES, MMB_GLOWSTONE, MMB_SHROOMLIGHT, MMB_WARPED_DEEPSLATE,
    MMB_WARPED_BLACKSTONE, MMB_MOSS, MMB_AMBER_DEEPSLATE, MMB_AMBER_BLACKSTONE, MMB_WARPED_STEM, MMB_WARPED_WART, MMB_AMBER_STEM,
    MMB_AMBER_WART, MMB_COBBLESTONE, MMB_COBBLED_DEEPSLATE, MMB_BRAIN_CORAL_BLOCK, MMB_BUBBLE_CORAL_BLOCK, MMB_FIRE_CORAL_BLOCK,
    MMB_HORN_CORAL_BLOCK, MMB_TUBE_CORAL_BLOCK, MMB_SEA_LANTERN,
    MMB_NUM_BLOCKS
};
#define MMB_NUM_NON_SOLID_BLOCKS (MMB_KELP_END + 1)

/* enum Biome (biome.hpp:13-44) */
enum {
    MMBIO_CORAL_REEF, MMBIO_ARCHIPELAGO, MMBIO_WARM_OCEAN, MMBIO_ICEBERGS, MMBIO_COOL_OCEAN,
    MMBIO_ROCKY_BEACH, MMBIO_TROPICAL_BEACH, MMBIO_BEACH,
    MMBIO_SAVANNA, MMBIO_MESA, MMBIO_FROZEN_WASTELAND, MMBIO_REDWOOD_FOREST, MMBIO_SHREKS_SWAMP, MMBIO_SPARSE_DESERT,
    MMBIO_LUSH_BIRCH_FOREST, MMBIO_TIANZI_MOUNTAINS,
    MMBIO_JUNGLE, MMBIO_RED_DESERT, MMBIO_PURPLE_MUSHROOMS, MMBIO_CRYSTALS, MMBIO_OASIS, MMBIO_DESERT, MMBIO_PLAINS, MMBIO_MOUNTAINS
};
#define MMGEN_NUM_OCEAN_BIOMES 5
#define MMGEN_NUM_OCEAN_AND_BEACH_BIOMES 8

/* enum CaveBiome (biome.hpp:50-59) */
enum { MMCB_NONE, MMCB_CRYSTAL_CAVES, MMCB_LUSH_CAVES, MMCB_WARPED_FOREST, MMCB_AMBER_FOREST };

/* enum Material (biome.hpp:65-93): 0-9 forward stratified, 10-11 backward stratified, 12-19 eroded */
enum {
    MMM_BLACKSTONE, MMM_DEEPSLATE, MMM_SLATE, MMM_STONE, MMM_TUFF, MMM_CALCITE, MMM_GRANITE, MMM_TERRACOTTA, MMM_MARBLE, MMM_ANDESITE,
    MMM_RED_SANDSTONE, MMM_SANDSTONE,
    MMM_GRAVEL, MMM_CLAY, MMM_MUD, MMM_DIRT, MMM_RED_SAND, MMM_SAND, MMM_SMOOTH_SAND, MMM_SNOW
};

/* enum Feature (biome.hpp:117-158) */
enum {
    MMF_NONE, MMF_SPHERE, MMF_CORAL, MMF_KELP, MMF_ICEBERG, MMF_ACACIA_TREE, MMF_REDWOOD_TREE, MMF_CYPRESS_TREE, MMF_BIRCH_TREE,
    MMF_PINE_TREE, MMF_PINE_SHRUB, MMF_RAFFLESIA, MMF_LARGE_JUNGLE_TREE, MMF_SMALL_JUNGLE_TREE, MMF_TINY_JUNGLE_TREE,
    MMF_MEDIUM_PURPLE_MUSHROOM, MMF_PURPLE_MUSHROOM, MMF_MEDIUM_CRYSTAL, MMF_CRYSTAL, MMF_PALM_TREE, MMF_CACTUS
};

/* enum CaveFeature (biome.hpp:162-177) */
enum {
    MMCF_NONE, MMCF_TEST_GLOWSTONE_PILLAR, MMCF_TEST_SHROOMLIGHT_PILLAR, MMCF_CAVE_VINE, MMCF_GLOWSTONE_CLUSTER,
    MMCF_STORMLIGHT_SPHERE, MMCF_CEILING_STORMLIGHT_SPHERE, MMCF_CRYSTAL_PILLAR, MMCF_WARPED_FUNGUS, MMCF_AMBER_FUNGUS
};

/* mmgen_debug_probe function selectors (test-only entry point: one device math function per item) */
enum {
    MMGEN_PROBE_SIN, MMGEN_PROBE_COS, MMGEN_PROBE_POW, MMGEN_PROBE_ATAN2, MMGEN_PROBE_ACOS, MMGEN_PROBE_SIMPLEX2, MMGEN_PROBE_SIMPLEX3,
    MMGEN_PROBE_FBM2_5, MMGEN_PROBE_FBM3_4, MMGEN_PROBE_RAND3FROM3, MMGEN_PROBE_WORLEY2, MMGEN_PROBE_WORLEY3,
    MMGEN_PROBE_SPECIAL_CAVE_NOISE, MMGEN_PROBE_BIOME_HEIGHT, MMGEN_PROBE_CAVE_BIOME, MMGEN_PROBE_HASH, MMGEN_PROBE_RNG4_U01,
    MMGEN_PROBE_SIMPLEX3_SPLIT   /* simplex3 regrouped at the lattice (part1 -> corner gradients -> part3), as k_cave_voxels evaluates it */
};

#ifdef __cplusplus
}
#endif
#endif /* MMGEN_TYPES_H */
