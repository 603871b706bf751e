// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Restatement of the reference "content" layer:
//   src/terrain/block.hpp:5-154       enum Block
//   src/terrain/biome.hpp:13-260      enums Biome / CaveBiome / Material / Feature / CaveFeature, PODs, gen structs
//   src/terrain/biomeFuncs.hpp:41-53  getRandomBiome
//   src/terrain/biomeFuncs.hpp:109-220 biome / cave-biome noise and weights, getCaveBiome
//   src/terrain/biomeFuncs.hpp:224-383 getHeight (24 biomes)
//   src/terrain/biomeFuncs.hpp:385-707 biomeBlockPreProcess / biomeBlockPostProcess / caveBiomeBlockPostProcess
//   src/terrain/biomeFuncs.hpp:725-1256 BiomeUtils::init (tables)
#pragma once
#include <algorithm>
#include <initializer_list>
#include <vector>
#include <array>
#include <initializer_list>
#include "mmo_noise.h"

namespace mmo {

#define MAX_CAVE_LAYERS_PER_COLUMN 32
#define MAX_GATHERED_FEATURES_PER_CHUNK 2048
#define MAX_GATHERED_CAVE_FEATURES_PER_CHUNK 4096
#define SEA_LEVEL 128
#define LAVA_LEVEL 8
#define SQRT_2 1.41421356237309504880168872420f
#define PI 3.14159265358979323846264338327f
#define TWO_PI 6.28318530717958647692528676655f
#define PI_OVER_TWO 1.57079632679489661923132169163f
#define PI_OVER_FOUR 0.78539816339744830961566084581f

// block.hpp:5-154 — order is the wire contract (uint8 block ids)
enum class Block : unsigned char {
    AIR, WATER, LAVA, CAVE_VINES_MAIN, CAVE_VINES_GLOW_MAIN, CAVE_VINES_END, CAVE_VINES_GLOW_END, GRASS, JUNGLE_GRASS,
    SAVANNA_GRASS, WARPED_MUSHROOM, WARPED_ROOTS, NETHER_SPROUTS, INFECTED_MUSHROOM, AMBER_ROOTS, DANDELION, POPPY,
    PITCHER_BOTTOM, PITCHER_TOP, CORNFLOWER, BLUE_ORCHID, ALLIUM, RED_TULIP, ORANGE_TULIP, WHITE_TULIP, PINK_TULIP,
    LILAC_BOTTOM, LILAC_TOP, PEONY_BOTTOM, PEONY_TOP, OXEYE_DAISY, LILY_OF_THE_VALLEY, JUNGLE_FERN, SMALL_MAGENTA_CRYSTAL,
    SMALL_CYAN_CRYSTAL, SMALL_GREEN_CRYSTAL, SMALL_PURPLE_MUSHROOM, DEAD_BUSH, HANGING_SMALL_MAGENTA_CRYSTAL,
    HANGING_SMALL_CYAN_CRYSTAL, HANGING_SMALL_GREEN_CRYSTAL, TALL_GRASS_BOTTOM, TALL_GRASS_TOP, TALL_JUNGLE_GRASS_BOTTOM,
    TALL_JUNGLE_GRASS_TOP, TORCHFLOWER, BRAIN_CORAL, BUBBLE_CORAL, FIRE_CORAL, HORN_CORAL, TUBE_CORAL, SEAGRASS,
    TALL_SEAGRASS_BOTTOM, TALL_SEAGRASS_TOP, KELP_MAIN, KELP_END,
    BEDROCK,
    STONE, DIRT, GRASS_BLOCK, SAND, GRAVEL, MYCELIUM, SNOW, SNOWY_GRASS_BLOCK, MUSHROOM_STEM, MUSHROOM_UNDERSIDE,
    PURPLE_MUSHROOM_CAP, MARBLE, ANDESITE, CALCITE, BLACKSTONE, TUFF, DEEPSLATE, GRANITE, SLATE, SANDSTONE, CLAY, RED_SAND,
    RED_SANDSTONE, MUD, JUNGLE_GRASS_BLOCK, RAFFLESIA_PETAL, RAFFLESIA_CENTER, RAFFLESIA_SPIKES, RAFFLESIA_STEM, JUNGLE_WOOD,
    JUNGLE_LEAVES_PLAIN, JUNGLE_LEAVES_FRUITS, CACTUS, PALM_WOOD, PALM_LEAVES, MAGENTA_CRYSTAL, CYAN_CRYSTAL, GREEN_CRYSTAL,
    SMOOTH_SAND, TERRACOTTA, YELLOW_TERRACOTTA, ORANGE_TERRACOTTA, PURPLE_TERRACOTTA, RED_TERRACOTTA, WHITE_TERRACOTTA,
    QUARTZ, ICE, PACKED_ICE, BLUE_ICE, SAVANNA_GRASS_BLOCK, BIRCH_WOOD, BIRCH_LEAVES, YELLOW_BIRCH_LEAVES, ORANGE_BIRCH_LEAVES,
    ACACIA_WOOD, ACACIA_LEAVES, SMOOTH_SANDSTONE, PINE_WOOD, PINE_LEAVES_1, PINE_LEAVES_2, REDWOOD_WOOD, REDWOOD_LEAVES,
    CYPRESS_WOOD, CYPRESS_LEAVES, GLOWSTONE, SHROOMLIGHT, WARPED_DEEPSLATE, WARPED_BLACKSTONE, MOSS, AMBER_DEEPSLATE,
    AMBER_BLACKSTONE, WARPED_STEM, WARPED_WART, AMBER_STEM, AMBER_WART, COBBLESTONE, COBBLED_DEEPSLATE, BRAIN_CORAL_BLOCK,
    BUBBLE_CORAL_BLOCK, FIRE_CORAL_BLOCK, HORN_CORAL_BLOCK, TUBE_CORAL_BLOCK, SEA_LANTERN
};
static constexpr int numBlocks = (int)Block::SEA_LANTERN + 1;
static constexpr int numNonSolidBlocks = (int)Block::KELP_END + 1;

// biome.hpp:13-44
enum class Biome : unsigned char {
    CORAL_REEF, ARCHIPELAGO, WARM_OCEAN, ICEBERGS, COOL_OCEAN,
    ROCKY_BEACH, TROPICAL_BEACH, BEACH,
    SAVANNA, MESA, FROZEN_WASTELAND, REDWOOD_FOREST, SHREKS_SWAMP, SPARSE_DESERT, LUSH_BIRCH_FOREST, TIANZI_MOUNTAINS,
    JUNGLE, RED_DESERT, PURPLE_MUSHROOMS, CRYSTALS, OASIS, DESERT, PLAINS, MOUNTAINS
};
static constexpr int numBiomes = (int)Biome::MOUNTAINS + 1;
static constexpr int numOceanBiomes = (int)Biome::COOL_OCEAN + 1;
static constexpr int numOceanAndBeachBiomes = (int)Biome::BEACH + 1;

enum class CaveBiome : unsigned char { NONE, CRYSTAL_CAVES, LUSH_CAVES, WARPED_FOREST, AMBER_FOREST };
static constexpr int numCaveBiomes = (int)CaveBiome::AMBER_FOREST + 1;

enum class Material : unsigned char {
    BLACKSTONE, DEEPSLATE, SLATE, STONE, TUFF, CALCITE, GRANITE, TERRACOTTA, MARBLE, ANDESITE,
    RED_SANDSTONE, SANDSTONE,
    GRAVEL, CLAY, MUD, DIRT, RED_SAND, SAND, SMOOTH_SAND, SNOW
};
static constexpr int numMaterials = (int)Material::SNOW + 1;
static constexpr int numStratifiedMaterials = (int)Material::SANDSTONE + 1;
static constexpr int numForwardMaterials = (int)Material::ANDESITE + 1;
static constexpr int numErodedMaterials = numMaterials - numStratifiedMaterials;

struct MaterialInfo {
    Block block;
    float thickness;
    float noiseAmplitudeOrTanAngleOfRepose;
    float noiseScaleOrMaxSlope;
};

struct CaveLayer {      // biome.hpp:106-115, 12 bytes
    int start;          // exclusive (is not air)
    int end;            // inclusive (is air)
    CaveBiome bottomBiome;
    CaveBiome topBiome;
    char padding[2];
};
static_assert(sizeof(CaveLayer) == 12, "CaveLayer wire size");

enum class Feature : unsigned char {
    NONE, SPHERE, CORAL, KELP, ICEBERG, ACACIA_TREE, REDWOOD_TREE, CYPRESS_TREE, BIRCH_TREE, PINE_TREE, PINE_SHRUB,
    RAFFLESIA, LARGE_JUNGLE_TREE, SMALL_JUNGLE_TREE, TINY_JUNGLE_TREE, MEDIUM_PURPLE_MUSHROOM, PURPLE_MUSHROOM,
    MEDIUM_CRYSTAL, CRYSTAL, PALM_TREE, CACTUS
};
static constexpr int numFeatures = (int)Feature::CACTUS + 1;

enum class CaveFeature : unsigned char {
    NONE, TEST_GLOWSTONE_PILLAR, TEST_SHROOMLIGHT_PILLAR, CAVE_VINE, GLOWSTONE_CLUSTER, STORMLIGHT_SPHERE,
    CEILING_STORMLIGHT_SPHERE, CRYSTAL_PILLAR, WARPED_FUNGUS, AMBER_FUNGUS
};
static constexpr int numCaveFeatures = (int)CaveFeature::AMBER_FUNGUS + 1;

struct FeatureGenTopLayer { Material material; float minThickness; };

struct FeatureGen {
    Feature feature;
    int gridCellSize;
    int gridCellPadding;
    float chancePerGridCell;
    std::vector<FeatureGenTopLayer> possibleTopLayers;
    bool canReplaceBlocks = true;
    FeatureGen(Feature f, int cs, int pad, float chance, std::vector<FeatureGenTopLayer> tl)
        : feature(f), gridCellSize(cs), gridCellPadding(pad), chancePerGridCell(chance), possibleTopLayers(tl) {}
    FeatureGen& setNotReplaceBlocks() { canReplaceBlocks = false; return *this; }
};

struct FeaturePlacement {       // biome.hpp:195-200, 20 bytes: feature@0 pos@4 canReplace@16
    Feature feature;
    ivec3 pos;
    bool canReplaceBlocks;
};
static_assert(sizeof(FeaturePlacement) == 20, "FeaturePlacement wire size");

struct CaveFeatureGen {
    CaveFeature caveFeature;
    int gridCellSize;
    int gridCellPadding;
    float chancePerGridCell;
    int minLayerHeight = 0;
    bool canReplaceBlocks = true;
    bool generatesFromCeiling = false;
    bool canGenerateInLava = false;
    CaveFeatureGen(CaveFeature f, int cs, int pad, float chance)
        : caveFeature(f), gridCellSize(cs), gridCellPadding(pad), chancePerGridCell(chance) {}
    CaveFeatureGen& setMinLayerHeight(int h) { minLayerHeight = h; return *this; }
    CaveFeatureGen& setNotReplaceBlocks() { canReplaceBlocks = false; return *this; }
    CaveFeatureGen& setGeneratesFromCeiling() { generatesFromCeiling = true; return *this; }
    CaveFeatureGen& setCanGenerateInLava() { canGenerateInLava = true; return *this; }
};

struct CaveFeaturePlacement {   // biome.hpp:239-245, 24 bytes
    CaveFeature feature;
    ivec3 pos;
    int layerHeight;
    bool canReplaceBlocks;
};
static_assert(sizeof(CaveFeaturePlacement) == 24, "CaveFeaturePlacement wire size");

// std::unordered_set<Block> of the reference (biome.hpp:251-252) as a small vector with the three members the path uses
// (empty / find / end: membership only, iteration order never matters)
struct BlockSet {
    std::vector<Block> v;
    BlockSet() {}
    BlockSet(std::initializer_list<Block> l) : v(l) {}
    BlockSet(const std::vector<Block>& l) : v(l) {}
    bool empty() const { return v.empty(); }
    std::vector<Block>::const_iterator end() const { return v.end(); }
    std::vector<Block>::const_iterator find(Block b) const { return std::find(v.begin(), v.end(), b); }
};

struct DecoratorGen {           // biome.hpp:247-287
    Block decoratorBlock;
    float chance;
    BlockSet possibleUnderBlocks;
    BlockSet possibleReplaceBlocks{Block::AIR};
    Block secondDecoratorBlock = Block::AIR;
    bool generatesFromCeiling = false;
    DecoratorGen(Block b, float c, std::vector<Block> under) : decoratorBlock(b), chance(c), possibleUnderBlocks(under) {}
    DecoratorGen& setWater() { possibleReplaceBlocks = BlockSet{Block::WATER}; return *this; }
    DecoratorGen& setSecondDecoratorBlock(Block b) { secondDecoratorBlock = b; return *this; }
    DecoratorGen& setGeneratesFromCeiling() { generatesFromCeiling = true; return *this; }
};

// ------------------------------------------------------------------ tables (BiomeUtils::init, biomeFuncs.hpp:725-1256)
enum W : unsigned char { wI, wP, wN, W_IGNORED = wI, W_POSITIVE = wP, W_NEGATIVE = wN };
typedef W BiomeWeightType;               // biome.hpp's enum class BiomeWeightType; the short names keep the 24 x 6 rule table readable
struct BiomeWeights { W ocean, beach, rocky, magic, temperature, moisture; };
struct CaveBiomeWeights { W none, shallow, warped, rocky; };

struct Tables {
    BiomeWeights biomeNoiseWeights[numBiomes];
    CaveBiomeWeights caveBiomeNoiseWeights[numCaveBiomes];
    struct BiomeBlocks { Block grassBlock; } biomeBlocks[numBiomes];    // biome.hpp:60-63
    MaterialInfo materialInfos[numMaterials];
    float biomeMaterialWeights[numBiomes * numMaterials];
    ivec2 dirVecs2d[8];
    std::array<std::vector<FeatureGen>, numBiomes> biomeFeatureGens;
    ivec2 featureHeightBounds[numFeatures];
    std::array<std::vector<CaveFeatureGen>, numCaveBiomes> caveBiomeFeatureGens;
    ivec2 caveFeatureHeightBounds[numCaveFeatures];
    std::array<std::vector<DecoratorGen>, numBiomes> biomeDecoratorGens;
    std::array<std::vector<DecoratorGen>, numCaveBiomes> caveBiomeDecoratorGens;
    Tables();
};
const Tables& T();

// ------------------------------------------------------------------ biome noise / weights
struct BiomeNoise { float ocean, beach, rocky, magic, temperature, moisture; };
struct CaveBiomeNoise { float none, shallow, warped, rocky; };

static constexpr float overallBiomeScale = 0.32f;
static constexpr float overallCaveBiomeScale = 1.f;

template <int stride = 1>
static inline Biome getRandomBiome(const float* columnBiomeWeights, float rand)
{
    for (int i = 0; i < numBiomes; ++i) {
        rand -= columnBiomeWeights[stride * i];
        if (rand <= 0.f) return (Biome)i;
    }
    return Biome::PLAINS;
}

BiomeNoise getBiomeNoise(const vec2 worldBlockPos);
CaveBiomeNoise getCaveBiomeNoise(const vec3 worldBlockPos, float maxHeight);
float getBiomeWeight(Biome biome, const BiomeNoise& noise);
float getCaveBiomeWeight(CaveBiome biome, const CaveBiomeNoise& noise);
CaveBiome getCaveBiome(ivec3 worldBlockPos, float maxHeight, int seed);
float getHeight(Biome biome, vec2 pos);
bool biomeBlockPreProcess(Block* blockPtr, Biome biome, ivec3 worldBlockPos, float height);
bool biomeBlockPostProcess(Block* blockPtr, Biome biome, ivec3 worldBlockPos, float height, bool isTopBlock);
bool caveBiomeBlockPostProcess(Block* blockPtr, CaveBiome caveBiome, ivec3 worldBlockPos, int caveBottomDepth, int caveTopDepth);

}  // namespace mmo
