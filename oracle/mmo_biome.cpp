// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// Restatement of src/terrain/biomeFuncs.hpp (file:line cited per function).
#include "mmo_biome.h"

namespace mmo {

// ===================================================================================================
// BiomeUtils::init — biomeFuncs.hpp:725-1256
// ===================================================================================================
Tables::Tables()
{
#define BW(biome) biomeNoiseWeights[(int)Biome::biome]
    // ocean, beach, rocky, magic, temperature, moisture      (biomeFuncs.hpp:735-762)
    BW(CORAL_REEF)        = {wP, wN, wP, wP, wI, wI};
    BW(ARCHIPELAGO)       = {wP, wN, wP, wN, wI, wI};
    BW(WARM_OCEAN)        = {wP, wN, wN, wI, wP, wI};
    BW(ICEBERGS)          = {wP, wN, wN, wP, wN, wI};
    BW(COOL_OCEAN)        = {wP, wN, wN, wN, wN, wI};
    BW(ROCKY_BEACH)       = {wP, wP, wP, wI, wI, wI};
    BW(TROPICAL_BEACH)    = {wP, wP, wN, wI, wP, wI};
    BW(BEACH)             = {wP, wP, wN, wI, wN, wI};
    BW(SAVANNA)           = {wN, wI, wP, wP, wP, wP};
    BW(MESA)              = {wN, wI, wP, wP, wP, wN};
    BW(FROZEN_WASTELAND)  = {wN, wI, wP, wP, wN, wP};
    BW(REDWOOD_FOREST)    = {wN, wI, wP, wP, wN, wN};
    BW(SHREKS_SWAMP)      = {wN, wI, wP, wN, wP, wP};
    BW(SPARSE_DESERT)     = {wN, wI, wP, wN, wP, wN};
    BW(LUSH_BIRCH_FOREST) = {wN, wI, wP, wN, wN, wP};
    BW(TIANZI_MOUNTAINS)  = {wN, wI, wP, wN, wN, wN};
    BW(JUNGLE)            = {wN, wI, wN, wP, wP, wP};
    BW(RED_DESERT)        = {wN, wI, wN, wP, wP, wN};
    BW(PURPLE_MUSHROOMS)  = {wN, wI, wN, wP, wN, wP};
    BW(CRYSTALS)          = {wN, wI, wN, wP, wN, wN};
    BW(OASIS)             = {wN, wI, wN, wN, wP, wP};
    BW(DESERT)            = {wN, wI, wN, wN, wP, wN};
    BW(PLAINS)            = {wN, wI, wN, wN, wN, wP};
    BW(MOUNTAINS)         = {wN, wI, wN, wN, wN, wN};
#undef BW
#define CW(b) caveBiomeNoiseWeights[(int)CaveBiome::b]
    // none, shallow, warped, rocky                             (biomeFuncs.hpp:769-776)
    CW(NONE)          = {wP, wI, wI, wI};
    CW(CRYSTAL_CAVES) = {wN, wP, wI, wP};
    CW(LUSH_CAVES)    = {wN, wP, wI, wN};
    CW(WARPED_FOREST) = {wI, wN, wP, wI};
    CW(AMBER_FOREST)  = {wI, wN, wN, wI};
#undef CW

    // biomeFuncs.hpp:786-801 (BiomeBlocks default grassBlock = DIRT, biome.hpp:60-63)
    for (int i = 0; i < numBiomes; ++i) biomeBlocks[i].grassBlock = Block::DIRT;
    biomeBlocks[(int)Biome::TROPICAL_BEACH].grassBlock = Block::JUNGLE_GRASS_BLOCK;
    biomeBlocks[(int)Biome::SAVANNA].grassBlock = Block::SAVANNA_GRASS_BLOCK;
    biomeBlocks[(int)Biome::FROZEN_WASTELAND].grassBlock = Block::SNOWY_GRASS_BLOCK;
    biomeBlocks[(int)Biome::REDWOOD_FOREST].grassBlock = Block::GRASS_BLOCK;
    biomeBlocks[(int)Biome::SHREKS_SWAMP].grassBlock = Block::JUNGLE_GRASS_BLOCK;
    biomeBlocks[(int)Biome::LUSH_BIRCH_FOREST].grassBlock = Block::GRASS_BLOCK;
    biomeBlocks[(int)Biome::TIANZI_MOUNTAINS].grassBlock = Block::GRASS_BLOCK;
    biomeBlocks[(int)Biome::JUNGLE].grassBlock = Block::JUNGLE_GRASS_BLOCK;
    biomeBlocks[(int)Biome::PURPLE_MUSHROOMS].grassBlock = Block::MYCELIUM;
    biomeBlocks[(int)Biome::OASIS].grassBlock = Block::JUNGLE_GRASS_BLOCK;
    biomeBlocks[(int)Biome::PLAINS].grassBlock = Block::GRASS_BLOCK;
    biomeBlocks[(int)Biome::MOUNTAINS].grassBlock = Block::GRASS_BLOCK;

    // biomeFuncs.hpp:808-837: material/block, thickness, noise amplitude, noise scale
#define MI(m, v1, v2, v3) materialInfos[(int)Material::m] = {Block::m, v1, v2, v3}
    MI(BLACKSTONE, 32.f, 32.f, 0.0030f);
    MI(DEEPSLATE, 66.f, 20.f, 0.0045f);
    MI(SLATE, 6.f, 24.f, 0.0062f);
    MI(STONE, 40.f, 30.f, 0.0050f);
    MI(TUFF, 24.f, 42.f, 0.0060f);
    MI(CALCITE, 20.f, 30.f, 0.0040f);
    MI(GRANITE, 18.f, 36.f, 0.0034f);
    MI(TERRACOTTA, 32.f, 16.f, 0.0020f);
    MI(MARBLE, 28.f, 56.f, 0.0050f);
    MI(ANDESITE, 24.f, 48.f, 0.0030f);
    MI(RED_SANDSTONE, 3.0f, 2.0f, 0.0035f);
    MI(SANDSTONE, 3.5f, 1.5f, 0.0025f);
    // thickness, angle of repose (degrees), maximum slope
    MI(GRAVEL, 2.5f, 55.f, 1.8f);
    MI(CLAY, 2.7f, 40.f, 1.8f);
    MI(MUD, 2.3f, 45.f, 1.6f);
    MI(DIRT, 4.2f, 40.f, 1.2f);
    MI(RED_SAND, 3.5f, 30.f, 1.5f);
    MI(SAND, 3.8f, 35.f, 1.4f);
    MI(SMOOTH_SAND, 4.5f, 65.f, 4.0f);
    MI(SNOW, 2.5f, 45.f, 1.5f);
#undef MI
    // biomeFuncs.hpp:843-847: tanf(glm::radians(angle)) runs on the host (MSVC libm) in the reference.  Frozen here as
    // hex floats (correctly rounded tan of the fp32 radian value deg * 0.0174532925f; tests re-derive them in fp64) so that the
    // contract does not depend on any libm.
    static const float tanAoR[8] = {0x1.6d9b1ap+0f, 0x1.ad9e76p-1f, 0x1p+0f, 0x1.ad9e76p-1f,
                                    0x1.279a74p-1f, 0x1.66819ap-1f, 0x1.127f34p+1f, 0x1p+0f};
    for (int l = numStratifiedMaterials; l < numMaterials; ++l)
        materialInfos[l].noiseAmplitudeOrTanAngleOfRepose = tanAoR[l - numStratifiedMaterials];

    // biomeFuncs.hpp:856-957: biomeMaterialWeights[material + numMaterials * biome]
#define CUR(m, w) biomeMaterialWeights[(int)Material::m + numMaterials * b] = w
#define BM(biome, m, w) biomeMaterialWeights[(int)Material::m + numMaterials * (int)Biome::biome] = w
    for (int i = 0; i < numBiomes * numMaterials; ++i) biomeMaterialWeights[i] = 1;
    for (int b = 0; b < numBiomes; ++b) {
        CUR(TERRACOTTA, 0.0f);
        CUR(RED_SANDSTONE, 0.0f);
        CUR(SANDSTONE, 0.0f);
        CUR(GRAVEL, 0.0f);
        CUR(CLAY, 0.0f);
        CUR(MUD, 0.0f);
        CUR(RED_SAND, 0.0f);
        CUR(SAND, 0.0f);
        CUR(SMOOTH_SAND, 0.0f);
        CUR(SNOW, 0.0f);
    }
    BM(CORAL_REEF, DIRT, 0.0f); BM(CORAL_REEF, SAND, 0.7f); BM(CORAL_REEF, SMOOTH_SAND, 0.8f);
    BM(ARCHIPELAGO, GRAVEL, 0.3f); BM(ARCHIPELAGO, DIRT, 0.0f); BM(ARCHIPELAGO, SAND, 0.8f);
    BM(WARM_OCEAN, DIRT, 0.0f); BM(WARM_OCEAN, SAND, 0.7f);
    BM(ICEBERGS, GRAVEL, 0.5f); BM(ICEBERGS, DIRT, 0.0f);
    BM(COOL_OCEAN, GRAVEL, 0.5f); BM(COOL_OCEAN, DIRT, 0.0f);
    BM(ROCKY_BEACH, DIRT, 0.0f); BM(ROCKY_BEACH, GRAVEL, 1.0f);
    BM(TROPICAL_BEACH, DIRT, 0.0f); BM(TROPICAL_BEACH, SMOOTH_SAND, 1.0f);
    BM(BEACH, DIRT, 0.0f); BM(BEACH, SAND, 1.0f);
    BM(SAVANNA, STONE, 0.6f); BM(SAVANNA, TUFF, 0.15f); BM(SAVANNA, CALCITE, 0.0f); BM(SAVANNA, GRANITE, 0.2f);
    BM(SAVANNA, TERRACOTTA, 3.2f); BM(SAVANNA, MARBLE, 0.0f);
    BM(MESA, CLAY, 0.8f); BM(MESA, DIRT, 0.0f);
    BM(FROZEN_WASTELAND, GRANITE, 0.0f); BM(FROZEN_WASTELAND, DIRT, 0.6f); BM(FROZEN_WASTELAND, SNOW, 1.1f);
    BM(SHREKS_SWAMP, CLAY, 1.7f); BM(SHREKS_SWAMP, MUD, 2.2f); BM(SHREKS_SWAMP, DIRT, 0.6f);
    BM(SPARSE_DESERT, MARBLE, 2.0f); BM(SPARSE_DESERT, ANDESITE, 0.5f); BM(SPARSE_DESERT, DIRT, 0.0f);
    BM(SPARSE_DESERT, SMOOTH_SAND, 1.4f);
    BM(TIANZI_MOUNTAINS, SANDSTONE, 1.0f);
    BM(JUNGLE, CLAY, 1.0f); BM(JUNGLE, MUD, 1.0f); BM(JUNGLE, DIRT, 0.5f);
    BM(RED_DESERT, RED_SANDSTONE, 1.0f); BM(RED_DESERT, DIRT, 0.0f); BM(RED_DESERT, RED_SAND, 1.0f);
    BM(PURPLE_MUSHROOMS, GRAVEL, 0.4f);
    BM(CRYSTALS, CALCITE, 0.3f); BM(CRYSTALS, GRAVEL, 0.15f); BM(CRYSTALS, CLAY, 0.2f); BM(CRYSTALS, DIRT, 0.0f);
    BM(OASIS, SANDSTONE, 1.0f); BM(OASIS, CLAY, 0.4f); BM(OASIS, DIRT, 0.6f); BM(OASIS, SAND, 0.4f);
    BM(DESERT, SANDSTONE, 1.0f); BM(DESERT, DIRT, 0.0f); BM(DESERT, SAND, 1.0f);
    BM(MOUNTAINS, GRAVEL, 1.0f);
#undef CUR
#undef BM

    // util/enums.hpp:29-38 (N, NE, E, SE, S, SW, W, NW)
    const ivec2 dv[8] = {{0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, -1}, {-1, 0}, {-1, 1}};
    for (int i = 0; i < 8; ++i) dirVecs2d[i] = dv[i];

    // ---- surface feature gens (biomeFuncs.hpp:974-1040): feature, gridCellSize, gridCellPadding, chancePerGridCell, top layers
    using M = Material;
#define FG(biome) biomeFeatureGens[(int)Biome::biome]
    FG(CORAL_REEF) = {FeatureGen(Feature::CORAL, 5, 0, 0.65f, {{M::SMOOTH_SAND, 0.3f}, {M::SAND, 0.3f}}),
                      FeatureGen(Feature::KELP, 8, 0, 0.50f, {{M::SMOOTH_SAND, 0.3f}, {M::SAND, 0.3f}})};
    FG(ICEBERGS) = {FeatureGen(Feature::ICEBERG, 112, 6, 0.70f, {})};
    FG(TROPICAL_BEACH) = {FeatureGen(Feature::PALM_TREE, 48, 3, 0.35f, {{M::SMOOTH_SAND, 0.3f}})};
    FG(SAVANNA) = {FeatureGen(Feature::ACACIA_TREE, 36, 4, 0.3f, {{M::DIRT, 0.5f}})};
    FG(REDWOOD_FOREST) = {FeatureGen(Feature::REDWOOD_TREE, 16, 2, 0.70f, {{M::DIRT, 0.5f}})};
    FG(SHREKS_SWAMP) = {FeatureGen(Feature::CYPRESS_TREE, 18, 3, 0.6f, {{M::DIRT, 0.5f}, {M::MUD, 0.5f}}),
                        FeatureGen(Feature::BIRCH_TREE, 16, 2, 0.15f, {{M::DIRT, 0.4f}})};
    FG(LUSH_BIRCH_FOREST) = {FeatureGen(Feature::BIRCH_TREE, 9, 2, 0.7f, {{M::DIRT, 0.5f}})};
    FG(TIANZI_MOUNTAINS) = {FeatureGen(Feature::PINE_TREE, 7, 1, 0.80f, {}).setNotReplaceBlocks(),
                            FeatureGen(Feature::PINE_SHRUB, 6, 1, 0.80f, {}).setNotReplaceBlocks()};
    FG(JUNGLE) = {FeatureGen(Feature::RAFFLESIA, 54, 6, 0.50f, {{M::DIRT, 0.5f}}),
                  FeatureGen(Feature::LARGE_JUNGLE_TREE, 28, 3, 0.70f, {{M::DIRT, 0.5f}}),
                  FeatureGen(Feature::SMALL_JUNGLE_TREE, 10, 2, 0.82f, {{M::DIRT, 0.5f}}),
                  FeatureGen(Feature::TINY_JUNGLE_TREE, 6, 1, 0.28f, {{M::DIRT, 0.5f}})};
    FG(RED_DESERT) = {FeatureGen(Feature::PALM_TREE, 40, 3, 0.20f, {{M::RED_SAND, 0.3f}}),
                      FeatureGen(Feature::CACTUS, 16, 2, 0.20f, {{M::RED_SAND, 0.5f}})};
    FG(PURPLE_MUSHROOMS) = {FeatureGen(Feature::MEDIUM_PURPLE_MUSHROOM, 10, 2, 0.50f, {{M::DIRT, 0.3f}}),
                            FeatureGen(Feature::PURPLE_MUSHROOM, 11, 3, 0.45f, {{M::DIRT, 0.5f}})};
    FG(CRYSTALS) = {FeatureGen(Feature::MEDIUM_CRYSTAL, 28, 6, 0.9f, {}),
                    FeatureGen(Feature::CRYSTAL, 52, 10, 0.8f, {})};
    FG(OASIS) = {FeatureGen(Feature::PALM_TREE, 24, 3, 0.35f, {{M::SAND, 0.3f}}),
                 FeatureGen(Feature::CACTUS, 16, 2, 0.40f, {{M::SAND, 0.5f}})};
    FG(DESERT) = {FeatureGen(Feature::PALM_TREE, 64, 3, 0.30f, {{M::SAND, 0.3f}}),
                  FeatureGen(Feature::CACTUS, 16, 2, 0.70f, {{M::SAND, 0.5f}})};
#undef FG
    // biomeFuncs.hpp:1042-1074: actual bounds = (pos.y + bounds[0], pos.y + bounds[1])
#define FHB(f, lo, hi) featureHeightBounds[(int)Feature::f] = {lo, hi}
    FHB(NONE, 0, 0); FHB(SPHERE, -6, 6); FHB(CORAL, -3, 12); FHB(KELP, 0, 20); FHB(ICEBERG, 0, 110);
    FHB(ACACIA_TREE, 0, 15); FHB(REDWOOD_TREE, -5, 75); FHB(CYPRESS_TREE, -3, 50); FHB(BIRCH_TREE, 0, 30);
    FHB(PINE_TREE, 0, 15); FHB(PINE_SHRUB, 0, 8); FHB(RAFFLESIA, 0, 10); FHB(TINY_JUNGLE_TREE, 0, 5);
    FHB(SMALL_JUNGLE_TREE, 0, 17); FHB(LARGE_JUNGLE_TREE, 0, 38); FHB(MEDIUM_PURPLE_MUSHROOM, 0, 6);
    FHB(PURPLE_MUSHROOM, 0, 120); FHB(MEDIUM_CRYSTAL, -3, 32); FHB(CRYSTAL, -6, 64); FHB(PALM_TREE, 0, 28);
    FHB(CACTUS, 0, 15);
#undef FHB

    // ---- surface decorators (biomeFuncs.hpp:1078-1178): decoratorBlock, chance, possibleUnderBlocks
    using B = Block;
    const std::vector<Block> coralReefBottomBlocks = {B::SAND, B::SMOOTH_SAND};
#define DG(biome) biomeDecoratorGens[(int)Biome::biome]
    DG(CORAL_REEF) = {
        DecoratorGen(B::SEAGRASS, 0.200f, coralReefBottomBlocks).setWater(),
        DecoratorGen(B::TALL_SEAGRASS_BOTTOM, 0.040f, coralReefBottomBlocks).setWater().setSecondDecoratorBlock(B::TALL_SEAGRASS_TOP),
        DecoratorGen(B::BRAIN_CORAL, 0.030f, coralReefBottomBlocks).setWater().setSecondDecoratorBlock(B::WATER),
        DecoratorGen(B::BUBBLE_CORAL, 0.030f, coralReefBottomBlocks).setWater().setSecondDecoratorBlock(B::WATER),
        DecoratorGen(B::FIRE_CORAL, 0.030f, coralReefBottomBlocks).setWater().setSecondDecoratorBlock(B::WATER),
        DecoratorGen(B::HORN_CORAL, 0.030f, coralReefBottomBlocks).setWater().setSecondDecoratorBlock(B::WATER),
        DecoratorGen(B::TUBE_CORAL, 0.030f, coralReefBottomBlocks).setWater().setSecondDecoratorBlock(B::WATER)};
    DG(ARCHIPELAGO) = {DecoratorGen(B::GRASS, 0.200f, {B::GRASS_BLOCK}), DecoratorGen(B::LILY_OF_THE_VALLEY, 0.025f, {B::GRASS_BLOCK})};
    DG(TROPICAL_BEACH) = {DecoratorGen(B::JUNGLE_GRASS, 0.1f, {B::JUNGLE_GRASS_BLOCK})};
    DG(SAVANNA) = {DecoratorGen(B::SAVANNA_GRASS, 0.1f, {B::SAVANNA_GRASS_BLOCK})};
    DG(REDWOOD_FOREST) = {
        DecoratorGen(B::GRASS, 0.200f, {B::GRASS_BLOCK}),
        DecoratorGen(B::TALL_GRASS_BOTTOM, 0.080f, {B::GRASS_BLOCK}).setSecondDecoratorBlock(B::TALL_GRASS_TOP),
        DecoratorGen(B::OXEYE_DAISY, 0.040f, {B::GRASS_BLOCK}),
        DecoratorGen(B::LILY_OF_THE_VALLEY, 0.040f, {B::GRASS_BLOCK}),
        DecoratorGen(B::PEONY_BOTTOM, 0.020f, {B::GRASS_BLOCK}).setSecondDecoratorBlock(B::PEONY_TOP)};
    DG(SHREKS_SWAMP) = {
        DecoratorGen(B::JUNGLE_GRASS, 0.300f, {B::JUNGLE_GRASS_BLOCK}), DecoratorGen(B::JUNGLE_FERN, 0.050f, {B::JUNGLE_GRASS_BLOCK}),
        DecoratorGen(B::CORNFLOWER, 0.030f, {B::JUNGLE_GRASS_BLOCK}), DecoratorGen(B::BLUE_ORCHID, 0.030f, {B::JUNGLE_GRASS_BLOCK}),
        DecoratorGen(B::ALLIUM, 0.030f, {B::JUNGLE_GRASS_BLOCK})};
    DG(LUSH_BIRCH_FOREST) = {
        DecoratorGen(B::GRASS, 0.300f, {B::GRASS_BLOCK}),
        DecoratorGen(B::PEONY_BOTTOM, 0.020f, {B::GRASS_BLOCK}).setSecondDecoratorBlock(B::PEONY_TOP),
        DecoratorGen(B::LILAC_BOTTOM, 0.020f, {B::GRASS_BLOCK}).setSecondDecoratorBlock(B::LILAC_TOP),
        DecoratorGen(B::DANDELION, 0.040f, {B::GRASS_BLOCK})};
    DG(JUNGLE) = {
        DecoratorGen(B::JUNGLE_GRASS, 0.400f, {B::JUNGLE_GRASS_BLOCK}),
        DecoratorGen(B::TALL_JUNGLE_GRASS_BOTTOM, 0.200f, {B::JUNGLE_GRASS_BLOCK}).setSecondDecoratorBlock(B::TALL_JUNGLE_GRASS_TOP),
        DecoratorGen(B::PITCHER_BOTTOM, 0.030f, {B::JUNGLE_GRASS_BLOCK}).setSecondDecoratorBlock(B::PITCHER_TOP),
        DecoratorGen(B::JUNGLE_FERN, 0.120f, {B::JUNGLE_GRASS_BLOCK}),
        DecoratorGen(B::BLUE_ORCHID, 0.040f, {B::JUNGLE_GRASS_BLOCK})};
    DG(RED_DESERT) = {DecoratorGen(B::DEAD_BUSH, 0.020f, {B::RED_SAND})};
    const std::vector<Block> smallCrystalBottomBlocks = {B::STONE, B::TUFF, B::CALCITE};
    DG(PURPLE_MUSHROOMS) = {
        DecoratorGen(B::SMALL_PURPLE_MUSHROOM, 0.100f, {B::MYCELIUM}), DecoratorGen(B::SMALL_MAGENTA_CRYSTAL, 0.005f, smallCrystalBottomBlocks),
        DecoratorGen(B::SMALL_CYAN_CRYSTAL, 0.005f, smallCrystalBottomBlocks), DecoratorGen(B::SMALL_GREEN_CRYSTAL, 0.005f, smallCrystalBottomBlocks)};
    DG(CRYSTALS) = {
        DecoratorGen(B::SMALL_PURPLE_MUSHROOM, 0.020f, {B::MYCELIUM}), DecoratorGen(B::SMALL_MAGENTA_CRYSTAL, 0.025f, smallCrystalBottomBlocks),
        DecoratorGen(B::SMALL_CYAN_CRYSTAL, 0.025f, smallCrystalBottomBlocks), DecoratorGen(B::SMALL_GREEN_CRYSTAL, 0.025f, smallCrystalBottomBlocks)};
    DG(OASIS) = {DecoratorGen(B::JUNGLE_GRASS, 0.200f, {B::JUNGLE_GRASS_BLOCK}), DecoratorGen(B::CORNFLOWER, 0.020f, {B::JUNGLE_GRASS_BLOCK})};
    DG(DESERT) = {DecoratorGen(B::DEAD_BUSH, 0.030f, {B::RED_SAND})};
    DG(PLAINS) = {
        DecoratorGen(B::GRASS, 0.200f, {B::GRASS_BLOCK}), DecoratorGen(B::RED_TULIP, 0.010f, {B::GRASS_BLOCK}),
        DecoratorGen(B::ORANGE_TULIP, 0.010f, {B::GRASS_BLOCK}), DecoratorGen(B::WHITE_TULIP, 0.010f, {B::GRASS_BLOCK}),
        DecoratorGen(B::PINK_TULIP, 0.010f, {B::GRASS_BLOCK}), DecoratorGen(B::DANDELION, 0.030f, {B::GRASS_BLOCK}),
        DecoratorGen(B::POPPY, 0.030f, {B::GRASS_BLOCK})};
    DG(MOUNTAINS) = {DecoratorGen(B::GRASS, 0.050f, {B::GRASS_BLOCK}), DecoratorGen(B::LILY_OF_THE_VALLEY, 0.015f, {B::GRASS_BLOCK})};
#undef DG

    // ---- cave feature gens (biomeFuncs.hpp:1188-1208): caveFeature, gridCellSize, gridCellPadding, chancePerGridCell
    using CF = CaveFeature;
#define CFG(b) caveBiomeFeatureGens[(int)CaveBiome::b]
    CFG(CRYSTAL_CAVES) = {
        CaveFeatureGen(CF::STORMLIGHT_SPHERE, 32, 4, 0.80f).setMinLayerHeight(4),
        CaveFeatureGen(CF::CEILING_STORMLIGHT_SPHERE, 32, 4, 0.80f).setMinLayerHeight(4).setGeneratesFromCeiling(),
        CaveFeatureGen(CF::CRYSTAL_PILLAR, 28, 5, 0.60f).setMinLayerHeight(10).setNotReplaceBlocks().setGeneratesFromCeiling()};
    CFG(LUSH_CAVES) = {
        CaveFeatureGen(CF::GLOWSTONE_CLUSTER, 24, 3, 0.60f).setMinLayerHeight(16).setNotReplaceBlocks().setGeneratesFromCeiling(),
        CaveFeatureGen(CF::CAVE_VINE, 4, 0, 0.40f).setMinLayerHeight(4).setNotReplaceBlocks().setGeneratesFromCeiling()};
    CFG(WARPED_FOREST) = {
        CaveFeatureGen(CF::GLOWSTONE_CLUSTER, 16, 3, 0.80f).setMinLayerHeight(16).setNotReplaceBlocks().setGeneratesFromCeiling(),
        CaveFeatureGen(CF::WARPED_FUNGUS, 7, 1, 0.75f).setMinLayerHeight(6).setNotReplaceBlocks()};
    CFG(AMBER_FOREST) = {
        CaveFeatureGen(CF::GLOWSTONE_CLUSTER, 18, 3, 0.75f).setMinLayerHeight(16).setNotReplaceBlocks().setGeneratesFromCeiling(),
        CaveFeatureGen(CF::AMBER_FUNGUS, 5, 1, 0.60f).setMinLayerHeight(9).setNotReplaceBlocks()};
#undef CFG
    // biomeFuncs.hpp:1210-1223: actual bounds = (pos.y + b[0], pos.y + layerHeight + b[1])
#define CHB(f, lo, hi) caveFeatureHeightBounds[(int)CaveFeature::f] = {lo, hi}
    CHB(NONE, 0, 0); CHB(TEST_GLOWSTONE_PILLAR, -3, 3); CHB(TEST_SHROOMLIGHT_PILLAR, -3, 3); CHB(CAVE_VINE, 0, 0);
    CHB(GLOWSTONE_CLUSTER, 0, 6); CHB(STORMLIGHT_SPHERE, -12, 12); CHB(CEILING_STORMLIGHT_SPHERE, -12, 12);
    CHB(CRYSTAL_PILLAR, -8, 8); CHB(WARPED_FUNGUS, -2, 3); CHB(AMBER_FUNGUS, -2, 5);
#undef CHB

    // ---- cave decorators (biomeFuncs.hpp:1228-1252)
#define CDG(b) caveBiomeDecoratorGens[(int)CaveBiome::b]
    CDG(CRYSTAL_CAVES) = {
        DecoratorGen(B::SMALL_MAGENTA_CRYSTAL, 0.015f, {}), DecoratorGen(B::SMALL_CYAN_CRYSTAL, 0.015f, {}),
        DecoratorGen(B::SMALL_GREEN_CRYSTAL, 0.015f, {}),
        DecoratorGen(B::HANGING_SMALL_MAGENTA_CRYSTAL, 0.015f, {}).setGeneratesFromCeiling(),
        DecoratorGen(B::HANGING_SMALL_CYAN_CRYSTAL, 0.015f, {}).setGeneratesFromCeiling(),
        DecoratorGen(B::HANGING_SMALL_GREEN_CRYSTAL, 0.015f, {}).setGeneratesFromCeiling()};
    CDG(LUSH_CAVES) = {
        DecoratorGen(B::GRASS, 0.100f, {B::MOSS}),
        DecoratorGen(B::TALL_GRASS_BOTTOM, 0.030f, {B::MOSS}).setSecondDecoratorBlock(B::TALL_GRASS_TOP),
        DecoratorGen(B::TORCHFLOWER, 0.020f, {B::MOSS})};
    CDG(WARPED_FOREST) = {
        DecoratorGen(B::WARPED_MUSHROOM, 0.020f, {B::WARPED_DEEPSLATE, B::WARPED_BLACKSTONE}),
        DecoratorGen(B::WARPED_ROOTS, 0.060f, {B::WARPED_DEEPSLATE, B::WARPED_BLACKSTONE}),
        DecoratorGen(B::NETHER_SPROUTS, 0.040f, {B::WARPED_DEEPSLATE, B::WARPED_BLACKSTONE})};
    CDG(AMBER_FOREST) = {
        DecoratorGen(B::INFECTED_MUSHROOM, 0.020f, {B::AMBER_DEEPSLATE, B::AMBER_BLACKSTONE}),
        DecoratorGen(B::AMBER_ROOTS, 0.060f, {B::AMBER_DEEPSLATE, B::AMBER_BLACKSTONE})};
#undef CDG
}

const Tables& T()
{
    static const Tables t;
    return t;
}

// ===================================================================================================
// biome noise / weights — biomeFuncs.hpp:109-220
// ===================================================================================================
static inline float getSingleBiomeNoise(vec2 pos, float noiseScale, vec2 offset, float smoothstepThreshold)
{
    return g_smoothstep(-smoothstepThreshold, smoothstepThreshold, simplex(pos * noiseScale + offset));
}

BiomeNoise getBiomeNoise(const vec2 worldBlockPos)     // biomeFuncs.hpp:114-128
{
    const vec2 noiseOffset = fbm2From2<3>(worldBlockPos * 0.0150f) * 20.f;
    const vec2 biomeNoisePos = (worldBlockPos + noiseOffset) * overallBiomeScale;

    BiomeNoise noise;
    float oceanNoise = simplex(biomeNoisePos * 0.0007f + vec2(2853.49f, -9481.42f));
    noise.ocean = g_smoothstep(0.01f, -0.02f, oceanNoise);
    noise.beach = g_smoothstep(-0.15f, -0.05f, oceanNoise);
    noise.rocky = getSingleBiomeNoise(biomeNoisePos, 0.0015f, vec2(-8102.35f, -7620.23f), 0.08f);
    noise.magic = getSingleBiomeNoise(biomeNoisePos, 0.0030f, vec2(5612.35f, 9182.49f), 0.07f);
    noise.temperature = getSingleBiomeNoise(biomeNoisePos, 0.0012f, vec2(-4021.34f, -8720.12f), 0.06f);
    noise.moisture = getSingleBiomeNoise(biomeNoisePos, 0.0050f, vec2(1835.32f, 3019.39f), 0.12f);
    return noise;
}

static inline float getSingleCaveBiomeNoise(vec3 pos, float noiseScale, vec3 offset, float smoothstepThreshold)
{
    return g_smoothstep(-smoothstepThreshold, smoothstepThreshold, simplex(pos * noiseScale + offset));
}

CaveBiomeNoise getCaveBiomeNoise(const vec3 worldBlockPos, float maxHeight)    // biomeFuncs.hpp:135-156
{
    const vec3 noiseOffset = fbm3From3<3>(worldBlockPos * 0.0470f) * vec3(30.f, 24.f, 30.f);
    const vec3 caveBiomeNoisePos = (worldBlockPos + noiseOffset) * vec3(overallCaveBiomeScale, 1.f, overallCaveBiomeScale);

    const vec2 noisePos2d = vec2(caveBiomeNoisePos.x, caveBiomeNoisePos.z) * 0.2000f;

    float caveNoiseTopHeight = (float)SEA_LEVEL + 0.15f * (maxHeight - (float)SEA_LEVEL);

    float noneToShallowStart = caveNoiseTopHeight - 19.f + 23.f * fbm<3>(noisePos2d);
    float noneToShallowEnd = noneToShallowStart - 5.f + 3.f * fbm<3>(noisePos2d + vec2(3821.34f, 4920.32f));

    float shallowToDeepStart = caveNoiseTopHeight - 72.f + 18.f * fbm<3>(noisePos2d + vec2(-4921.34f, 8402.13f));
    float shallowToDeepEnd = shallowToDeepStart - 10.f + 7.f * fbm<3>(noisePos2d + vec2(9411.32f, -3921.34f));

    CaveBiomeNoise noise;
    noise.none = g_smoothstep(noneToShallowEnd, noneToShallowStart, caveBiomeNoisePos.y);
    noise.shallow = g_smoothstep(shallowToDeepEnd, shallowToDeepStart, caveBiomeNoisePos.y);
    noise.warped = getSingleCaveBiomeNoise(caveBiomeNoisePos, 0.0030f, vec3(5821.32f, 4920.12f, 7931.59f), 0.05f);
    noise.rocky = getSingleCaveBiomeNoise(caveBiomeNoisePos, 0.0022f, vec3(-9193.23f, -6813.39f, (float)-2171.23), 0.05f);
    return noise;
}

static inline void applySingleBiomeNoise(float& totalWeight, const BiomeWeightType weight, const float noise)
{
    switch (weight) {
    case BiomeWeightType::W_POSITIVE: totalWeight *= noise; break;
    case BiomeWeightType::W_NEGATIVE: totalWeight *= 1.f - noise; break;
    default: break;
    }
}

// the reference's __constant__ copies of the rule tables (biomeFuncs.hpp:105-107)
#define dev_biomeNoiseWeights (T().biomeNoiseWeights)
#define dev_caveBiomeNoiseWeights (T().caveBiomeNoiseWeights)

float getBiomeWeight(Biome biome, const BiomeNoise& noise)     // biomeFuncs.hpp:171-185
{
    const auto& biomeWeights = dev_biomeNoiseWeights[(int)biome];

    float totalWeight = 1.f;

#define applyNoise(type) applySingleBiomeNoise(totalWeight, biomeWeights.type, noise.type)
    applyNoise(ocean);
    applyNoise(beach);
    applyNoise(rocky);
    applyNoise(magic);
    applyNoise(temperature);
    applyNoise(moisture);
#undef applyNoise

    return totalWeight;
}

float getCaveBiomeWeight(CaveBiome biome, const CaveBiomeNoise& noise)     // biomeFuncs.hpp:187-199
{
    const auto& caveBiomeWeights = dev_caveBiomeNoiseWeights[(int)biome];

    float totalWeight = 1.f;

#define applyNoise(type) applySingleBiomeNoise(totalWeight, caveBiomeWeights.type, noise.type)
    applyNoise(none);
    applyNoise(shallow);
    applyNoise(warped);
    applyNoise(rocky);
#undef applyNoise

    return totalWeight;
}

CaveBiome getCaveBiome(ivec3 worldBlockPos, float maxHeight, int seed)     // biomeFuncs.hpp:201-220
{
    CaveBiomeNoise noise = getCaveBiomeNoise(worldBlockPos, maxHeight);

    auto rng = makeSeededRandomEngine(worldBlockPos.x, worldBlockPos.y, worldBlockPos.z, seed);
    uniform_real_distribution<float> u01(0, 1);
    float rand = u01(rng);
    for (int caveBiomeIdx = 0; caveBiomeIdx < numCaveBiomes; ++caveBiomeIdx) {
        CaveBiome caveBiome = (CaveBiome)caveBiomeIdx;
        float weight = getCaveBiomeWeight(caveBiome, noise);
        rand -= weight;
        if (rand <= 0.f) return caveBiome;
    }
    return CaveBiome::NONE;
}

// ===================================================================================================
// getHeight — biomeFuncs.hpp:224-383
// ===================================================================================================
float getHeight(Biome biome, vec2 pos)
{
    switch (biome) {
    case Biome::CORAL_REEF:
        return 107.f + 16.f * fbm(pos * 0.0065f);
    case Biome::ARCHIPELAGO: {
        float islandNoise = (fbm<4>(pos * 0.0055f) + 1.f) * 0.5f;
        islandNoise = mm_powf(islandNoise, 2.4f);
        islandNoise = g_smoothstep(1.f, 0.f, islandNoise);
        float islandHeight = 22.f * islandNoise;
        float baseHeight = 107.f + 24.f * fbm(pos * 0.0060f);
        return baseHeight + islandHeight;
    }
    case Biome::WARM_OCEAN:
        return 93.f + 18.f * fbm(pos * 0.0055f);
    case Biome::ICEBERGS:
        return 66.f + 18.f * fbm(pos * 0.0060f);
    case Biome::COOL_OCEAN:
        return 80.f + 22.f * fbm(pos * 0.0065f);
    case Biome::ROCKY_BEACH:
        return 134.f + 8.f * fbm(pos * 0.0070f);
    case Biome::TROPICAL_BEACH:
        return 129.5f + 6.f * fbm(pos * 0.0045f);
    case Biome::BEACH:
        return 132.f + 5.f * fbm(pos * 0.0055f);
    case Biome::SAVANNA: {
        vec2 noiseOffsetPos = pos * 0.0040f;
        vec2 noiseOffset = fbm2From2<5>(noiseOffsetPos) * 100.f;
        vec2 noisePos = pos + noiseOffset;

        float plateauNoise1 = worley(noisePos * 0.0070f);
        plateauNoise1 = g_smoothstep(0.30f, 0.20f, plateauNoise1) * (1.f + 0.3f * simplex(noisePos * 0.0100f));

        float plateauNoise2 = worley((noisePos + vec2(-3910.12f, -9012.34f)) * 0.0045f);
        plateauNoise2 = g_smoothstep(0.16f, 0.08f, plateauNoise2) * (1.f + 0.2f * simplex(noisePos * 0.0130f));

        float plateauHeight = (plateauNoise1 * 14.f) + (plateauNoise2 * 9.f);
        return 136.f + 9.f * fbm<4>(pos * 0.0080f) + plateauHeight;
    }
    case Biome::MESA: {
        pos *= 0.7f;
        vec2 noiseOffsetPos = pos * 0.0050f;
        vec2 noiseOffset = fbm2From2<5>(noiseOffsetPos) * 300.f;
        float riverNoise;
        worley((pos + noiseOffset) * 0.0030f, nullptr, &riverNoise);

        float baseHeight = 122.f;
        baseHeight += 10.f * g_smoothstep(0.00f, 0.05f, riverNoise);
        baseHeight += (37.5f + 5.0f * fbm<4>((pos + 0.02f * noiseOffset) * 0.0300f)) * g_smoothstep(0.07f, 0.22f, riverNoise);

        return baseHeight + 6.f * simplex(pos * 0.0250f);
    }
    case Biome::FROZEN_WASTELAND:
        return 136.f + 16.f * fbm(pos * 0.0035f);
    case Biome::REDWOOD_FOREST:
        return 134.f + 8.f * fbm(pos * 0.0120f);
    case Biome::SHREKS_SWAMP:
        return 130.f + 12.f * fbm(pos * 0.0080f);
    case Biome::SPARSE_DESERT: {
        vec2 noiseOffset = simplex2From2(pos * 0.0080f) * 20.0f;
        float dunesNoise = mm_powf(worley((pos + noiseOffset) * 0.0160f), 2.f) * 18.f;
        return 132.f + 4.f * fbm<4>(pos * 0.0070f) + dunesNoise;
    }
    case Biome::LUSH_BIRCH_FOREST: {
        float hillsHeight = (simplex(pos * 0.0012f) + 0.8f) * 20.f;
        return 135.f + 8.f * fbm(pos * 0.0090f) + hillsHeight;
    }
    case Biome::TIANZI_MOUNTAINS: {
        vec2 noiseOffset = simplex2From2(pos * 0.0800f) * 3.0f;
        vec2 noisePos = (pos + noiseOffset) * 0.0150f;

        float worley1 = g_smoothstep(0.45f, 0.35f, worley(noisePos)) * 1.2f;
        float worley2 = g_smoothstep(0.45f, 0.35f, worley(noisePos * 1.4f + vec2(4292.12f, 9183.27f))) * 0.6f;
        float mountainsHeight = worley1 + worley2;
        mountainsHeight *= 54.f + 7.f * fbm<3>(noisePos * 1.7f);

        float hillsHeight = 16.f * simplex(pos * 0.0150f);

        return 128.f + hillsHeight + 9.f * fbm<3>(pos * 0.0070f) + mountainsHeight;
    }
    case Biome::JUNGLE: {
        float hillsHeight = (simplex(pos * 0.0030f) + 0.5f) * 25.f;
        return 139.f + 8.f * fbm(pos * 0.0120f) + hillsHeight;
    }
    case Biome::RED_DESERT:
        return 137.f + 13.f * fbm(pos * 0.0075f);
    case Biome::PURPLE_MUSHROOMS:
        return 136.f + 9.f * fbm(pos * 0.0140f);
    case Biome::CRYSTALS: {
        float towersBaseNoise = simplex(pos * 0.0030f);

        vec3 worleyColor;
        float towersWorleyNoise;
        worley(pos * 0.0700f, &worleyColor, &towersWorleyNoise);
        towersWorleyNoise = g_smoothstep(0.10f, 0.15f, towersWorleyNoise);
        towersWorleyNoise *= 0.4f + 1.2f * worleyColor.x;
        float towersHeight = 60.f * towersWorleyNoise * g_smoothstep(0.70f, 0.74f, towersBaseNoise);
        towersHeight += 18.f * g_smoothstep(0.35f, 0.8f, towersBaseNoise);

        float baseHeight = 137.f + 8.f * fbm(pos * 0.0200f);
        return baseHeight + towersHeight;
    }
    case Biome::OASIS:
        return 132.f + 9.f * fbm(pos * 0.0120f);
    case Biome::DESERT:
        return 136.f + 6.f * fbm(pos * 0.0110f);
    case Biome::PLAINS:
        return 144.f + 8.f * fbm(pos * 0.0080f);
    case Biome::MOUNTAINS: {
        float noise = mm_powf(fabsf(fbm(pos * 0.0035f)) + 0.05f, 2.f);
        noise += ((fbm(pos * 0.0050f) - 0.5f) * 2.f) * 0.05f;
        return 165.f + (140.f * (noise - 0.15f)) + (noise * (20.f * fbm(pos * 0.0350f)));
    }
    }
    return (float)SEA_LEVEL;
}

// ===================================================================================================
// block pre/post-process — biomeFuncs.hpp:385-707
// ===================================================================================================
bool biomeBlockPreProcess(Block* blockPtr, Biome biome, ivec3 worldBlockPos, float height)     // :385-406
{
    switch (biome) {
    case Biome::CRYSTALS: {
        if (height > 176.f) {
            float quartzStartHeight = 140.f + 15.f * fbm<3>(vec2((float)worldBlockPos.x, (float)worldBlockPos.z) * 0.0080f);
            if ((float)worldBlockPos.y > quartzStartHeight) {
                *blockPtr = Block::QUARTZ;
                return true;
            }
        }
        return false;
    }
    default: break;
    }
    return false;
}

bool biomeBlockPostProcess(Block* blockPtr, Biome biome, ivec3 worldBlockPos, float height, bool isTopBlock)   // :408-590
{
    (void)height;
    switch (biome) {
    case Biome::ARCHIPELAGO: {
        if (worldBlockPos.y < SEA_LEVEL || *blockPtr == Block::WATER) return false;
        float dirtHeight = (float)SEA_LEVEL + 1.5f + 1.7f * fbm<3>(vec2((float)worldBlockPos.x, (float)worldBlockPos.z) * 0.0065f);
        if ((float)worldBlockPos.y > dirtHeight) {
            *blockPtr = isTopBlock ? Block::GRASS_BLOCK : Block::DIRT;
            return true;
        }
        return false;
    }
    case Biome::TROPICAL_BEACH: {
        if (isTopBlock && *blockPtr != Block::SMOOTH_SAND && *blockPtr != Block::WATER) {
            *blockPtr = Block::SMOOTH_SAND;
            return true;
        }
        return false;
    }
    case Biome::BEACH: {
        if (isTopBlock && *blockPtr != Block::SAND && *blockPtr != Block::WATER) {
            *blockPtr = Block::SAND;
            return true;
        }
        return false;
    }
    case Biome::MESA: {
        if ((float)worldBlockPos.y < 90.f || *blockPtr == Block::WATER) return false;

        vec2 pos2d = vec2((float)worldBlockPos.x, (float)worldBlockPos.z);
        float terracottaStartHeight = 108.f + 12.f * fbm<3>(pos2d * 0.0040f);
        if ((float)worldBlockPos.y < terracottaStartHeight) return false;

        if (*blockPtr == Block::CLAY && (float)worldBlockPos.y < terracottaStartHeight + 20.f) return false;

        float sampleHeight = (float)worldBlockPos.y + 3.f * simplex(vec3(pos2d * 0.0100f, (float)worldBlockPos.y * 0.0300f)) - terracottaStartHeight;
        sampleHeight = g_mod(sampleHeight, 32.f);
        Block terracottaBlock;
        if (sampleHeight < 5.f) terracottaBlock = Block::TERRACOTTA;
        else if (sampleHeight < 8.f) terracottaBlock = Block::ORANGE_TERRACOTTA;
        else if (sampleHeight < 12.f) terracottaBlock = Block::RED_TERRACOTTA;
        else if (sampleHeight < 14.f) terracottaBlock = Block::WHITE_TERRACOTTA;
        else if (sampleHeight < 20.f) terracottaBlock = Block::TERRACOTTA;
        else if (sampleHeight < 21.f) terracottaBlock = Block::ORANGE_TERRACOTTA;
        else if (sampleHeight < 26.f) terracottaBlock = Block::YELLOW_TERRACOTTA;
        else if (sampleHeight < 29.f) terracottaBlock = Block::PURPLE_TERRACOTTA;
        else terracottaBlock = Block::TERRACOTTA;

        *blockPtr = terracottaBlock;
        return true;
    }
    case Biome::FROZEN_WASTELAND: {
        if (*blockPtr != Block::WATER) return false;
        *blockPtr = Block::PACKED_ICE;
        return true;
    }
    case Biome::SHREKS_SWAMP: {
        if ((float)worldBlockPos.y < 100.f) return false;
        if (*blockPtr == Block::DIRT || *blockPtr == Block::JUNGLE_GRASS_BLOCK) {
            float mudEnd = (float)SEA_LEVEL + 0.8f + 1.1f * simplex(vec2((float)worldBlockPos.x, (float)worldBlockPos.z) * 0.0300f);
            if ((float)worldBlockPos.y < mudEnd) {
                *blockPtr = Block::MUD;
                return true;
            }
        }
        return false;
    }
    case Biome::TIANZI_MOUNTAINS: {
        if ((float)worldBlockPos.y < 90.f || *blockPtr == Block::WATER || *blockPtr == Block::DIRT || *blockPtr == Block::GRASS_BLOCK)
            return false;
        float sandstoneStartHeight = 112.f + 16.f * fbm<3>(vec2((float)worldBlockPos.x, (float)worldBlockPos.z) * 0.0200f);
        if ((float)worldBlockPos.y < sandstoneStartHeight) return false;
        *blockPtr = Block::SMOOTH_SANDSTONE;
        return true;
    }
    case Biome::CRYSTALS: {
        if (!isTopBlock || *blockPtr == Block::QUARTZ) return false;
        if (rand1From2(vec2(worldBlockPos.x + 913213, worldBlockPos.z + 85941)) < 0.1f) {
            *blockPtr = Block::MYCELIUM;
            return true;
        }
        return false;
    }
    case Biome::MOUNTAINS: {
        if ((float)worldBlockPos.y < 190.f) return false;
        float snowStartHeight = 202.f + 5.f * fbm<3>(vec2((float)worldBlockPos.x, (float)worldBlockPos.z) * 0.0500f);
        if ((float)worldBlockPos.y < snowStartHeight) return false;
        *blockPtr = Block::SNOW;
        return true;
    }
    default: break;
    }
    return false;
}

bool caveBiomeBlockPostProcess(Block* blockPtr, CaveBiome caveBiome, ivec3 worldBlockPos, int caveBottomDepth, int caveTopDepth)   // :592-707
{
    if (caveBiome == CaveBiome::NONE) return false;

    bool isTopBlock = caveBottomDepth == 0;
    bool isBottomBlock = caveTopDepth == 0;
    (void)isBottomBlock;                                  // unused in the reference too

    switch (caveBiome) {
    case CaveBiome::CRYSTAL_CAVES: {
        if (*blockPtr != Block::STONE && *blockPtr != Block::DEEPSLATE && *blockPtr != Block::BLACKSTONE) return false;

        vec3 noisePos = vec3(worldBlockPos.x + worldBlockPos.y, worldBlockPos.z + 5819323, (worldBlockPos.x + worldBlockPos.z) * 2.0f) * 0.05f;
        float quartzNoise = simplex(noisePos);
        if (quartzNoise < -0.25f) {
            *blockPtr = Block::QUARTZ;
            return true;
        }

        if (*blockPtr == Block::BLACKSTONE) return false;

        float cobblestoneChance;
        Block cobblestoneBlock;
        if (*blockPtr == Block::STONE) {
            cobblestoneChance = 0.5f;
            cobblestoneBlock = Block::COBBLESTONE;
        } else {
            cobblestoneChance = 0.4f;
            cobblestoneBlock = Block::COBBLED_DEEPSLATE;
        }

        if (rand1From3(worldBlockPos) < cobblestoneChance) {
            *blockPtr = cobblestoneBlock;
            return true;
        }
        return false;
    }
    case CaveBiome::LUSH_CAVES: {
        if (*blockPtr != Block::STONE && *blockPtr != Block::DEEPSLATE && *blockPtr != Block::BLACKSTONE) return false;

        vec3 noisePos = vec3(worldBlockPos) * 0.025f;
        float threshold = 1.5f + 4.5f * simplex(noisePos);
        if (!isInRange((float)caveBottomDepth, 0.f, threshold) && !isInRange((float)caveTopDepth, 0.f, threshold)) return false;

        noisePos.y += 192031.9821f;
        vec3 noiseOffset = fbm3From3<3>(noisePos * 0.4f) * 2.f;
        float clayNoise = worley(noisePos + noiseOffset);

        *blockPtr = clayNoise < 0.25f ? Block::CLAY : Block::MOSS;
        return true;
    }
    case CaveBiome::WARPED_FOREST: {
        if (!isTopBlock) return false;
        if (*blockPtr == Block::DEEPSLATE) { *blockPtr = Block::WARPED_DEEPSLATE; return true; }
        else if (*blockPtr == Block::BLACKSTONE) { *blockPtr = Block::WARPED_BLACKSTONE; return true; }
        return false;
    }
    case CaveBiome::AMBER_FOREST: {
        if (!isTopBlock) return false;
        if (*blockPtr == Block::DEEPSLATE) { *blockPtr = Block::AMBER_DEEPSLATE; return true; }
        else if (*blockPtr == Block::BLACKSTONE) { *blockPtr = Block::AMBER_BLACKSTONE; return true; }
        return false;
    }
    default: break;
    }
    __builtin_unreachable();      // the reference falls off the end here (biomeFuncs.hpp:706-707); every CaveBiome value returns above
}

}  // namespace mmo
