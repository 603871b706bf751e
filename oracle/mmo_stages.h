// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the chunk pipeline stages of src/terrain/chunk.cu, one function per reference
// kernel / host stage, operating on the reference's per-chunk staging layouts:
//   heightfield[256]            idx2d = x + 16 z                         (chunk.hpp:59)
//   biomeWeights[24][256]       biome-major planes                       (chunk.hpp:69)
//   layers[20][256]             layer-major planes, value = layer start  (chunk.hpp:63)
//   caveLayers[256][32]         column-major, 12 B each                  (chunk.hpp:66)
//   blocks[98304]               idx = y + 384 (x + 16 z)                 (chunk.hpp:72)
// Canonical choices for reference behaviour that is undefined or schedule dependent are listed in
// DESIGN.md "Canonical semantics" and marked CANONICAL below.
#pragma once
#include "mmo_biome.h"

namespace mmo {

static constexpr int ZONE_SIZE = 12;
static constexpr int EROSION_GRID_SIDE = ZONE_SIZE * 2 * 16;            // 384
static constexpr int EROSION_GRID_NUM_COLS = EROSION_GRID_SIDE * EROSION_GRID_SIDE;

// K1  kernGenerateHeightfield chunk.cu:150-185
void generateHeightfield(ivec2 chunkWorldBlockPos, float* heightfield /*256*/, float* biomeWeights /*24*256*/);
// single column (used for the slope ring: the height is a pure function of position)
float columnHeight(ivec2 worldPos, float* weights24 /*nullable*/);

// G1  otherChunkGatherHeightfield chunk.cu:237-293, restated as a pure function of position
void gatherHeightfield(ivec2 chunkWorldBlockPos, const float* heightfield /*256*/, float* gathered /*324*/);

// K2  kernGenerateLayers chunk.cu:322-415
void generateLayers(ivec2 chunkWorldBlockPos, const float* gatheredHeightfield /*324*/, const float* biomeWeights, float* layers /*20*256*/);

// E1+K3  copyLayers(to) + kernDoErosion loop chunk.cu:477-705 on the packed zone planes.
// gathered: [9][384*384] (8 eroded-layer starts + heightfield), eroded in place.  Returns the number of relaxation passes.
int erodeZonePlanes(float* gathered);
// E1 / E3  copyLayers chunk.cu:603-656 over chunk-major staging arrays (chunkIdx[576] with heightfields: into the zone planes;
// chunkIdx[144], heightfields = null: the eroded planes of the zone's own chunks back into `layers`)
void zoneCopyLayers(float* layers, float* heightfields, const int* chunkIdx, float* gatheredLayers, bool toGatheredLayers);
// E3  fixBackwardStratifiedLayers chunk.cu:725-749
void fixBackwardStratifiedLayers(float* layers /*20*256*/);

// C1  shouldGenerateCaveAtBlock chunk.cu:755-810
bool shouldGenerateCaveAtBlock(ivec3 worldPos, float maxHeight, float oceanAndBeachWeight);
// K4  kernGenerateCaves chunk.cu:812-937
void generateCaves(ivec2 chunkWorldBlockPos, const float* heightfield, const float* biomeWeights, CaveLayer* caveLayers /*256*32*/);

// F1  generateFeaturePlacements chunk.cu:999-1156
void generateFeaturePlacements(ivec2 chunkWorldBlockPos, const float* heightfield, const float* biomeWeights, const float* layers,
                               const CaveLayer* caveLayers, std::vector<FeaturePlacement>& out, std::vector<CaveFeaturePlacement>& caveOut);
// F2  gather order chunk.cu:1158-1167
extern const ivec2 gatherFeaturePlacementsChunkOffsets[49];

// L2  placeFeature / placeCaveFeature featurePlacement.hpp:147-1379
bool placeFeature(const FeaturePlacement& featurePlacement, ivec3 worldBlockPos, Block* blockPtr);
bool placeCaveFeature(const CaveFeaturePlacement& caveFeaturePlacement, ivec3 worldBlockPos, Block* blockPtr);

// K6  kernFill + host part of Chunk::fill chunk.cu:1202-1601 (lists are the gathered lists, un-truncated; truncation and
// the NONE sentinel are applied inside exactly as Chunk::fill does)
void fillChunk(ivec3 chunkWorldBlockPos, const float* heightfield, const float* biomeWeights, const float* layers, const CaveLayer* caveLayers,
               const FeaturePlacement* features, int numFeatures_, const CaveFeaturePlacement* caveFeatures, int numCaveFeatures_,
               Block* blocks /*98304*/);

// D1  placeDecorators chunk.cu:1634-1747
void placeDecorators(ivec3 chunkWorldBlockPos, const float* heightfield, const float* biomeWeights, const CaveLayer* caveLayers, Block* blocks);

// counters for the undefined-behaviour cases of SURVEY §7.3-3 (read by tests)
struct UbCounters {
    long long noLayerFound;        // thisLayerIdx == -1 (chunk.cu:1349-1363)
    long long caveLayerOverflow;   // > 32 cave layers in a column (chunk.cu:902-907)
    long long decoratorOutOfRange; // decorator y outside [0,383] (chunk.cu:1728)
};
extern UbCounters g_ub;

}  // namespace mmo
