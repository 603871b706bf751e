// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// Restatement of the kernels and host stages of src/terrain/chunk.cu (file:line cited per function).
#include "mmo_stages.h"
#include <cstring>
#include <algorithm>

namespace mmo {

UbCounters g_ub = {0, 0, 0};

static inline int posTo2dIndex16(int x, int z) { return x + 16 * z; }
static inline int posTo2dIndex(int x, int z) { return x + 16 * z; }                          // posTo2dIndex<16>, biomeFuncs.hpp:11-16
static inline int posTo3dIndex(ivec3 pos) { return pos.y + 384 * posTo2dIndex(pos.x, pos.z); }   // biomeFuncs.hpp:25-37
// the three places where the canonical semantics add a statement to a member function of the reference; all are named
// macros so that tools/extract_ref_literals.py can list and drop exactly these
#define CANONICAL_RETURN_FALSE return false
#define CANONICAL_NO_LAYER_FOUND(idx, blockPtr) if ((idx) < 0) { ++g_ub.noLayerFound; *(blockPtr) = Block::STONE; } else
#define CANONICAL_DECORATOR_RANGE(pos) if ((pos).y < 0 || (pos).y > 383) { ++g_ub.decoratorOutOfRange; return; }
// the reference's __constant__ / host copies of BiomeUtils' tables (biomeFuncs.hpp:709-723)
#define dev_materialInfos (T().materialInfos)
#define dev_biomeBlocks (T().biomeBlocks)
#define host_biomeFeatureGens (T().biomeFeatureGens)
#define host_caveBiomeFeatureGens (T().caveBiomeFeatureGens)
#define host_biomeDecoratorGens (T().biomeDecoratorGens)
#define host_caveBiomeDecoratorGens (T().caveBiomeDecoratorGens)
static inline int posTo2dIndex18(int x, int z) { return x + 18 * z; }

// ===================================================================================================
// K1 — kernGenerateHeightfield chunk.cu:150-185
// ===================================================================================================
float columnHeight(ivec2 worldPosI, float* weights24)
{
    const vec2 worldPos = vec2(worldPosI);     // int add, then convert (chunk.cu:162)
    const BiomeNoise biomeNoise = getBiomeNoise(worldPos);
    float height = 0.f;
    for (int biomeIdx = 0; biomeIdx < numBiomes; ++biomeIdx) {
        Biome biome = (Biome)biomeIdx;
        float weight = getBiomeWeight(biome, biomeNoise);
        if (weight > 0.f) height += weight * getHeight(biome, worldPos);
        if (weights24) weights24[biomeIdx] = weight;
    }
    return height;
}

void generateHeightfield(ivec2 chunkWorldBlockPos, float* heightfield, float* biomeWeights)
{
    for (int z = 0; z < 16; ++z) {
        for (int x = 0; x < 16; ++x) {
            const int idx = posTo2dIndex16(x, z);
            float w[numBiomes];
            heightfield[idx] = columnHeight(chunkWorldBlockPos + ivec2{x, z}, w);
            for (int b = 0; b < numBiomes; ++b) biomeWeights[256 * b + idx] = w[b];
        }
    }
}

// ===================================================================================================
// G1 — gathered 18x18 heightfield (chunk.cu:237-293).  The reference copies the border rows/columns/corners of the
// 8 neighbouring chunks; a neighbour's heightfield entry is columnHeight() of that world position, so the gather is
// restated as a pure function of position (no neighbour objects needed).
// ===================================================================================================
void gatherHeightfield(ivec2 chunkWorldBlockPos, const float* heightfield, float* gathered)
{
    for (int gz = 0; gz < 18; ++gz) {
        for (int gx = 0; gx < 18; ++gx) {
            const int x = gx - 1, z = gz - 1;
            float h;
            if (x >= 0 && x < 16 && z >= 0 && z < 16) h = heightfield[posTo2dIndex16(x, z)];
            else h = columnHeight(chunkWorldBlockPos + ivec2{x, z}, nullptr);
            gathered[posTo2dIndex18(gx, gz)] = h;
        }
    }
}

// ===================================================================================================
// K2 — getStratifiedMaterialThickness chunk.cu:308-320, kernGenerateLayers chunk.cu:322-415
// ===================================================================================================
static float getStratifiedMaterialThickness(int layerIdx, float materialWeight, vec2 worldPos)
{
    if (materialWeight > 0) {
        const auto& materialInfo = dev_materialInfos[layerIdx];
        vec2 noisePos = worldPos * materialInfo.noiseScaleOrMaxSlope + vec2((float)layerIdx * 5283.64f);
        return g_max(0.f, materialInfo.thickness + materialInfo.noiseAmplitudeOrTanAngleOfRepose * fbm(noisePos)) * materialWeight;
    }
    else return 0;
}

void generateLayers(ivec2 chunkWorldBlockPos, const float* gatheredHeightfield, const float* biomeWeights, float* layers)
{
    const Tables& t = T();
    for (int z = 0; z < 16; ++z) {
        for (int x = 0; x < 16; ++x) {
            const int idx = posTo2dIndex16(x, z);
            const vec2 worldPos = vec2(chunkWorldBlockPos + ivec2{x, z});

            float totalMaterialWeights[numMaterials];
            for (int m = 0; m < numMaterials; ++m) totalMaterialWeights[m] = 0;
            const float* columnBiomeWeights = biomeWeights + idx;
            for (int b = 0; b < numBiomes; ++b) {
                const float biomeWeight = columnBiomeWeights[256 * b];
                for (int m = 0; m < numMaterials; ++m)
                    totalMaterialWeights[m] += biomeWeight * t.biomeMaterialWeights[m + numMaterials * b];
            }

            const ivec2 pos18 = {x + 1, z + 1};
            const float maxHeight = gatheredHeightfield[posTo2dIndex18(pos18.x, pos18.y)];

            float slope = 0;
            for (int i = 0; i < 8; ++i) {
                const ivec2 p = pos18 + t.dirVecs2d[i];
                float neighborHeight = gatheredHeightfield[posTo2dIndex18(p.x, p.y)];
                slope = g_max(slope, fabsf(neighborHeight - maxHeight) * (i % 2 == 1 ? SQRT_2 : 1.f));
            }

            float* columnLayers = layers + idx;

            float height = 0;
            int layerIdx = 0;
            for (; layerIdx < numForwardMaterials; ++layerIdx) {
                columnLayers[256 * layerIdx] = height;
                if (height > maxHeight || layerIdx == numForwardMaterials - 1) break;
                height += getStratifiedMaterialThickness(layerIdx, totalMaterialWeights[layerIdx], worldPos);
            }
            // CANONICAL: forward layers after the early break are never written by the reference (stale device memory,
            // chunk.cu:383-393); they cannot influence blocks (the layer search in fill finds an earlier layer first), so the
            // canonical value is the running height.
            for (++layerIdx; layerIdx < numForwardMaterials; ++layerIdx) columnLayers[256 * layerIdx] = height;

            height = 0;
            for (int l = numStratifiedMaterials - 1; l >= numForwardMaterials; --l) {
                height += getStratifiedMaterialThickness(l, totalMaterialWeights[l], worldPos);
                columnLayers[256 * l] = height;
            }

            height = maxHeight;
            for (int l = numMaterials - 1; l >= numStratifiedMaterials; --l) {
                const auto& materialInfo = t.materialInfos[l];
                float materialWeight = totalMaterialWeights[l];
                float layerHeight = g_max(0.f, materialInfo.thickness * ((materialInfo.noiseScaleOrMaxSlope - slope) / materialInfo.noiseScaleOrMaxSlope)) * materialWeight;
                height -= layerHeight;
                columnLayers[256 * l] = height;
            }
        }
    }
}

// ===================================================================================================
// K3 — kernDoErosion chunk.cu:477-601 + host loop Chunk::erodeZone chunk.cu:682-705
// CANONICAL: every relaxation pass is a synchronous Jacobi step on a snapshot of (layer start plane, accumulated heights).
// The reference updates both in place while other thread blocks are still reading their halos (chunk.cu:544-554 vs
// :578,585), so its result depends on block scheduling; the snapshot semantics is the schedule-free reading of the same
// arithmetic.
// ===================================================================================================
int erodeZonePlanes(float* g)
{
    const Tables& t = T();
    const int N = EROSION_GRID_NUM_COLS, S = EROSION_GRID_SIDE;
    std::vector<float> acc(N, 0.f), startOld(N), accOld(N);
    int passes = 0;

    for (int layerIdx = numErodedMaterials - 1; layerIdx >= 0; --layerIdx) {
        float* startPlane = g + (size_t)N * layerIdx;
        const float* endPlane = g + (size_t)N * (layerIdx + 1);
        const float tanAngleOfRepose = t.materialInfos[numStratifiedMaterials + layerIdx].noiseAmplitudeOrTanAngleOfRepose;
        bool isFirst = true;
        bool changed;
        do {
            changed = false;
            std::memcpy(startOld.data(), startPlane, sizeof(float) * N);
            std::memcpy(accOld.data(), acc.data(), sizeof(float) * N);
            for (int gz = 0; gz < S; ++gz) {
                for (int gx = 0; gx < S; ++gx) {
                    const int c = gx + S * gz;
                    const float thisAcc = isFirst ? accOld[c] : 0;
                    const float thisLayerStart = startOld[c] + thisAcc;
                    const float thisLayerEnd = endPlane[c] + thisAcc;

                    float newLayerStart = thisLayerStart;
                    float maxThickness = thisLayerEnd - thisLayerStart;
                    for (int i = 0; i < 8; ++i) {
                        const int nx = g_clamp(gx + t.dirVecs2d[i].x, 0, S - 1);
                        const int nz = g_clamp(gz + t.dirVecs2d[i].y, 0, S - 1);
                        const int n = nx + S * nz;
                        const float nAcc = isFirst ? accOld[n] : 0;
                        const float neighborLayerStart = startOld[n] + nAcc;
                        const float neighborLayerEnd = endPlane[n] + nAcc;
                        newLayerStart = g_max(newLayerStart, neighborLayerStart - tanAngleOfRepose * (i % 2 == 1 ? SQRT_2 : 1.f));
                        maxThickness = g_max(maxThickness, neighborLayerEnd - neighborLayerStart);
                    }
                    newLayerStart = g_min(newLayerStart, thisLayerEnd);

                    if (maxThickness > 0) {
                        startPlane[c] = newLayerStart;
                        if (newLayerStart != thisLayerStart) {
                            changed = true;
                            acc[c] += newLayerStart - thisLayerStart;
                        }
                    }
                }
            }
            isFirst = false;
            ++passes;
        } while (changed);
    }
    return passes;
}

// E3 — fixBackwardStratifiedLayers chunk.cu:725-749
void fixBackwardStratifiedLayers(float* layers)
{
    for (int idx2d = 0; idx2d < 256; ++idx2d) {
        const float erodedStart = layers[256 * numStratifiedMaterials + idx2d];
        for (int l = numForwardMaterials; l < numStratifiedMaterials; ++l)
            layers[256 * l + idx2d] = erodedStart - layers[256 * l + idx2d];
    }
}

// ===================================================================================================
// C1 — shouldGenerateCaveAtBlock chunk.cu:755-810
// ===================================================================================================
bool shouldGenerateCaveAtBlock(ivec3 worldPos, float maxHeight, float oceanAndBeachWeight)
{
    if (worldPos.y == 0) return false;
    if (worldPos.y > (g_max((int)maxHeight, SEA_LEVEL))) return true;

    vec3 noisePos = vec3(worldPos) * 0.0050f;
    float topRatioYOffset = oceanAndBeachWeight * 50.f;
    float topHeightRatio = g_smoothstep(142.f, 95.f, (float)worldPos.y + topRatioYOffset);
    float bottomHeightRatio = g_smoothstep(5.f, 20.f, (float)worldPos.y);

    vec3 noiseOffset = fbm3From3<5>(noisePos * 0.8000f) * 1.8f;
    float caveNoise = specialCaveNoise(noisePos * vec3(1.f, 1.6f, 1.f) + noiseOffset);

    float worleyEdgeThreshold = 0.24f + 0.12f * fbm<4>(noisePos * 4.f);
    float hugeCaveNoise = g_smoothstep(0.2f, 0.4f, fbm<4>(noisePos * 0.0700f));
    worleyEdgeThreshold *= (1.f + 1.4f * hugeCaveNoise);
    worleyEdgeThreshold *= (topHeightRatio) * (0.3f + 0.7f * bottomHeightRatio);

    if (worleyEdgeThreshold > 0.04f && caveNoise < worleyEdgeThreshold) return true;

    vec2 ravineNoisePos = vec2((float)worldPos.x, (float)worldPos.z) * 0.0015f;
    vec2 ravineWorleyOffset = 0.03f * fbm2From2<4>(ravineNoisePos * 10.f);
    vec3 ravineWorleyColor;
    float ravineWorley = worley(ravineNoisePos + ravineWorleyOffset, &ravineWorleyColor);
    const float ravineWorleyThreshold = 0.12f * (1.f - oceanAndBeachWeight);
    if (ravineWorley < ravineWorleyThreshold) {
        float ravineTop = 120.f + 24.f * ravineWorleyColor.x;
        float ravineRatio = 1.f - (ravineWorley / ravineWorleyThreshold);

        float ravineDepth = 60.f + 26.f * fbm<4>(ravineNoisePos * 8.f + vec2(8391.32f, 4821.39f));
        ravineDepth *= g_smoothstep(0.f, 0.3f, ravineRatio);

        float ravineWaveNoiseOffset = 4.f * fbm<4>(ravineNoisePos * 3.f + vec2(5129.32f, 1392.49f));
        float ravineWaveNoise = mm_sinf((ravineNoisePos.x + ravineNoisePos.y) * 15.f + ravineWaveNoiseOffset);
        ravineWaveNoise = g_smoothstep(0.4f, 0.6f, ravineWaveNoise);
        ravineDepth *= ravineWaveNoise;

        if (ravineDepth > 0.0001f && (float)worldPos.y > ravineTop - ravineDepth) return true;
    }
    return false;
}

// ===================================================================================================
// K4 — kernGenerateCaves chunk.cu:812-937
// ===================================================================================================
void generateCaves(ivec2 chunkWorldBlockPos, const float* heightfield, const float* biomeWeights, CaveLayer* caveLayers)
{
    for (int z = 0; z < 16; ++z) {
        for (int x = 0; x < 16; ++x) {
            const int idx2d = posTo2dIndex16(x, z);
            const float maxHeight = heightfield[idx2d];
            // CANONICAL: the reference sums the 8 ocean+beach weights with shared-memory atomicAdd (order unspecified,
            // chunk.cu:846-850); canonical order is ascending biome index starting from 0.f.
            float oceanAndBeachWeight = 0.f;
            for (int b = 0; b < numOceanAndBeachBiomes; ++b) oceanAndBeachWeight += biomeWeights[256 * b + idx2d];

            const ivec2 wp2 = chunkWorldBlockPos + ivec2{x, z};
            int isFilled[385];
            for (int y = 0; y < 384; ++y)
                isFilled[y] = shouldGenerateCaveAtBlock(ivec3{wp2.x, y, wp2.y}, maxHeight, oceanAndBeachWeight) ? 0 : 1;
            isFilled[384] = 0;

            CaveLayer* col = caveLayers + MAX_CAVE_LAYERS_PER_COLUMN * idx2d;
            for (int k = 0; k < MAX_CAVE_LAYERS_PER_COLUMN; ++k) {   // default {384, 384} (chunk.cu:970-972)
                std::memset(&col[k], 0, sizeof(CaveLayer));
                col[k].start = 384;
                col[k].end = 384;
            }
            int numFlips = 0;
            for (int y = 0; y < 384; ++y) {
                if (isFilled[y] != isFilled[y + 1]) {
                    // CANONICAL: flips beyond 32 layers overflow into the next column's slot in the reference
                    // (chunk.cu:902-907); canonical = truncate, counted.
                    if (numFlips >= 2 * MAX_CAVE_LAYERS_PER_COLUMN) { ++g_ub.caveLayerOverflow; break; }
                    if ((numFlips & 1) == 0) col[numFlips >> 1].start = y;
                    else col[numFlips >> 1].end = y;
                    ++numFlips;
                }
            }

            for (int k = 0; k < MAX_CAVE_LAYERS_PER_COLUMN; ++k) {
                CaveLayer& caveLayer = col[k];
                if (caveLayer.start != 384)
                    caveLayer.bottomBiome = getCaveBiome(ivec3{wp2.x, caveLayer.start, wp2.y}, maxHeight, 329271348);
                if (caveLayer.end == 384) caveLayer.topBiome = CaveBiome::NONE;
                else caveLayer.topBiome = getCaveBiome(ivec3{wp2.x, caveLayer.end + 1, wp2.y}, maxHeight, 4982921);
            }
        }
    }
}

// ===================================================================================================
// F1 — feature placements chunk.cu:999-1156
// ===================================================================================================
bool isFeaturePos(ivec2 worldBlockPos2d, int gridCellSize, int gridCellPadding, int seed)     // chunk.cu:999-1008
{
    const ivec2 gridCornerWorldPos = ivec2(g_floor(vec2(worldBlockPos2d) / (float)gridCellSize) * (float)gridCellSize);
    const int gridCellInternalSideLength = gridCellSize - (2 * gridCellPadding);
    vec2 randPos = rand2From3(vec3(gridCornerWorldPos, seed));
    const ivec2 gridPlaceWorldPos = gridCornerWorldPos
        + ivec2(gridCellPadding)
        + ivec2(g_floor(randPos * (float)gridCellInternalSideLength));
    return worldBlockPos2d == gridPlaceWorldPos;
}

// The reference's Chunk as the host stages see it: the same member names over the caller's per-chunk staging arrays, so that the
// member functions below can be stated as they are written (this->caveLayers.data(), this->featurePlacements.push_back({...})).
template <class E> struct Staging {
    E* p;
    E* data() const { return p; }
    E& operator[](size_t i) const { return p[i]; }
};
struct Chunk {
    ivec3 worldBlockPos;
    Staging<const float> heightfield, biomeWeights, layers;
    Staging<const CaveLayer> caveLayers;
    Staging<Block> blocks;
    std::vector<FeaturePlacement>& featurePlacements;
    std::vector<CaveFeaturePlacement>& caveFeaturePlacements;

    bool tryGenerateCaveFeaturePlacement(const CaveFeatureGen& caveFeatureGen, const CaveLayer& caveLayer, bool top, int caveFeaturePlacementSeed,
                                         float rand, ivec2 worldBlockPos2d);
    void generateColumnFeaturePlacements(int localX, int localZ);
    void tryPlaceSingleDecorator(ivec3 pos, const DecoratorGen& gen);
    void placeDecorators();
};

bool Chunk::tryGenerateCaveFeaturePlacement(const CaveFeatureGen& caveFeatureGen, const CaveLayer& caveLayer, bool top,
                                            int caveFeaturePlacementSeed, float rand, ivec2 worldBlockPos2d)    // chunk.cu:1010-1038
{
    int layerHeight = caveLayer.end - caveLayer.start;

    if (rand >= caveFeatureGen.chancePerGridCell
        || (top != caveFeatureGen.generatesFromCeiling)
        || (!caveFeatureGen.canGenerateInLava && (top ? caveLayer.end : (caveLayer.start + 1)) <= LAVA_LEVEL)
        || layerHeight < caveFeatureGen.minLayerHeight)
    {
        return false;
    }

    if (isFeaturePos(worldBlockPos2d, caveFeatureGen.gridCellSize, caveFeatureGen.gridCellPadding, caveFeaturePlacementSeed))
    {
        this->caveFeaturePlacements.push_back({
            caveFeatureGen.caveFeature,
            ivec3(worldBlockPos2d.x, caveLayer.start + 1, worldBlockPos2d.y),
            layerHeight,
            caveFeatureGen.canReplaceBlocks
        });
        return true;
    }
    CANONICAL_RETURN_FALSE;      // the reference falls off the end of the function here (chunk.cu:1028-1038); canonical: false
}

void Chunk::generateColumnFeaturePlacements(int localX, int localZ)     // chunk.cu:1041-1145
{
    const int idx2d = posTo2dIndex(localX, localZ);

    const float* columnBiomeWeights = biomeWeights.data() + idx2d;

    const float height = heightfield[idx2d];
    const int groundHeight = (int)height;

    const ivec2 localBlockPos2d = ivec2(localX, localZ);
    const ivec2 worldBlockPos2d = ivec2(this->worldBlockPos.x, this->worldBlockPos.z) + localBlockPos2d;

    auto blockRng = makeSeededRandomEngine(worldBlockPos2d.x, worldBlockPos2d.y, 329828101);
    uniform_real_distribution<float> u01(0, 1);

    bool surfaceIsCave = false;
    const auto columnCaveLayers = this->caveLayers.data() + (idx2d * MAX_CAVE_LAYERS_PER_COLUMN);
    for (int caveLayerIdx = 0; caveLayerIdx < MAX_CAVE_LAYERS_PER_COLUMN; ++caveLayerIdx)
    {
        const auto& caveLayer = columnCaveLayers[caveLayerIdx];

        if (caveLayer.start == 384 || groundHeight <= caveLayer.start)
        {
            break;
        }

        for (const auto& caveFeatureGen : host_caveBiomeFeatureGens[(int)caveLayer.bottomBiome])
        {
            int caveFeaturePlacementSeed = (int)caveFeatureGen.caveFeature * 98239 + caveLayerIdx * 191702;
            if (tryGenerateCaveFeaturePlacement(caveFeatureGen, caveLayer, false, caveFeaturePlacementSeed, u01(blockRng), worldBlockPos2d))
            {
                break;
            }
        }

        if (caveLayer.end != 384)
        {
            for (const auto& caveFeatureGen : host_caveBiomeFeatureGens[(int)caveLayer.topBiome])
            {
                int caveFeaturePlacementSeed = (int)caveFeatureGen.caveFeature * 58321 + caveLayerIdx * 871503;
                if (tryGenerateCaveFeaturePlacement(caveFeatureGen, caveLayer, true, caveFeaturePlacementSeed, u01(blockRng), worldBlockPos2d))
                {
                    break;
                }
            }
        }

        if (groundHeight > caveLayer.start && groundHeight <= caveLayer.end)
        {
            surfaceIsCave = true;
            break;
        }
    }

    if (!surfaceIsCave)
    {
        Biome biome = getRandomBiome<256>(columnBiomeWeights, u01(blockRng));
        const auto& featureGens = host_biomeFeatureGens[(int)biome];

        const float* columnLayers = this->layers.data() + idx2d;

        for (const auto& featureGen : featureGens)
        {
            if (u01(blockRng) >= featureGen.chancePerGridCell)
            {
                continue;
            }

            if (!featureGen.possibleTopLayers.empty())
            {
                bool canPlace = false;
                for (const auto& possibleTopLayer : featureGen.possibleTopLayers)
                {
                    // layerIdx + 1 == numMaterials (SNOW) would read past `layers`: no gen table lists SNOW as a top layer (the
                    // sanitizer job would see the read)
                    int layerIdx = (int)possibleTopLayer.material;
                    float layerStart = columnLayers[256 * layerIdx];
                    float layerEnd = columnLayers[256 * (layerIdx + 1)];

                    if (layerStart > height || layerEnd < height || g_min(layerEnd, height) - layerStart < possibleTopLayer.minThickness)
                    {
                        continue;
                    }

                    canPlace = true;
                    break;
                }

                if (!canPlace)
                {
                    continue;
                }
            }

            if (isFeaturePos(worldBlockPos2d, featureGen.gridCellSize, featureGen.gridCellPadding, (int)featureGen.feature * 518721))
            {
                this->featurePlacements.push_back({
                    featureGen.feature,
                    ivec3(worldBlockPos2d.x, groundHeight + 1, worldBlockPos2d.y),
                    featureGen.canReplaceBlocks
                });
                break;
            }
        }
    }
}

// the 20 / 24-byte records are compared and shipped as bytes: the padding after the bool is zeroed (aggregate initialisation leaves it unspecified)
template <class P> static void zeroPadding(std::vector<P>& v, size_t from)
{
    for (size_t i = from; i < v.size(); ++i) {
        P q;
        std::memset((void*)&q, 0, sizeof(q));
        q.feature = v[i].feature; q.pos = v[i].pos; q.canReplaceBlocks = v[i].canReplaceBlocks;
        if constexpr (sizeof(P) == 24) q.layerHeight = v[i].layerHeight;
        std::memcpy((void*)&v[i], &q, sizeof(q));
    }
}

void generateFeaturePlacements(ivec2 chunkWorldBlockPos, const float* heightfield, const float* biomeWeights, const float* layers,
                               const CaveLayer* caveLayers, std::vector<FeaturePlacement>& out, std::vector<CaveFeaturePlacement>& caveOut)
{
    const size_t n0 = out.size(), c0 = caveOut.size();
    Chunk chunk{ivec3(chunkWorldBlockPos.x, 0, chunkWorldBlockPos.y), {heightfield}, {biomeWeights}, {layers}, {caveLayers}, {nullptr}, out, caveOut};
    for (int localZ = 0; localZ < 16; ++localZ)                           // chunk.cu:1147-1156
        for (int localX = 0; localX < 16; ++localX)
            chunk.generateColumnFeaturePlacements(localX, localZ);
    zeroPadding(out, n0);
    zeroPadding(caveOut, c0);
}

// F2 — chunk.cu:1158-1167
const ivec2 gatherFeaturePlacementsChunkOffsets[49] = {
    {0, 0}, {0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, -1},
    {-1, 0}, {-1, 1}, {2, 0}, {2, 1}, {2, 2}, {1, 2}, {0, 2},
    {-1, 2}, {-2, 2}, {-2, 1}, {-2, 0}, {-2, -1}, {-2, -2},
    {-1, -2}, {0, -2}, {1, -2}, {2, -2}, {2, -1},
    {-3, -3}, {-2, -3}, {-1, -3}, {0, -3}, {1, -3}, {2, -3}, {3, -3},
    {3, -2}, {3, -1}, {3, 0}, {3, 1}, {3, 2}, {3, 3},
    {2, 3}, {1, 3}, {0, 3}, {-1, 3}, {-2, 3}, {-3, 3},
    {-3, 2}, {-3, 1}, {-3, 0}, {-3, -1}, {-3, -2}};

// ===================================================================================================
// L1 — chunkFillPlaceBlock chunk.cu:1202-1380
// ===================================================================================================
static void chunkFillPlaceBlock(
    Block* blockPtr,
    const float* shared_biomeWeights,
    const float* shared_layersAndHeight,
    const CaveLayer* shared_caveLayers,
    int y,
    float height,
    ivec3 worldBlockPos,
    Rng& rng)
{
    if (y == 0)
    {
        *blockPtr = Block::BEDROCK;
        return;
    }

    if (y > height && y > SEA_LEVEL)
    {
        *blockPtr = Block::AIR;
        return;
    }

    bool isOcean = false;
    for (int biomeIdx = 0; biomeIdx < numOceanBiomes; ++biomeIdx)
    {
        if (shared_biomeWeights[biomeIdx] > 0.f)
        {
            isOcean = true;
            break;
        }
    }

    uniform_real_distribution<float> u01(0, 1);

    Biome randBiome = getRandomBiome(shared_biomeWeights, u01(rng));
    bool isTopBlock = y >= height - 1.f;

#define doBlockPostProcess() biomeBlockPostProcess(blockPtr, randBiome, worldBlockPos, height, isTopBlock)
#define postProcessCaveBiome getCaveBiome(worldBlockPos, height, 190249401)
#define doCaveBlockPostProcess() caveBiomeBlockPostProcess(blockPtr, postProcessCaveBiome, worldBlockPos, caveBottomDepth, caveTopDepth)

    if (y > height && y <= SEA_LEVEL)
    {
        *blockPtr = Block::WATER;
        doBlockPostProcess();

        if (isOcean)
        {
            return;
        }
    }

    int caveBottomDepth = -384;
    int caveTopDepth = -384;
    int caveLayerIdx = 0;
    for ( ; caveLayerIdx < MAX_CAVE_LAYERS_PER_COLUMN; ++caveLayerIdx)
    {
        const auto& caveLayer = shared_caveLayers[caveLayerIdx];
        if (caveLayer.start == 384)
        {
            caveBottomDepth = -384;
            break;
        }

        caveBottomDepth = caveLayer.start - y;

        if (y <= caveLayer.start)
        {
            break;
        }

        if (y <= caveLayer.end)
        {
            caveBottomDepth = caveLayer.start - y;
            caveTopDepth = y - (caveLayer.end + 1);
            *blockPtr = (y <= LAVA_LEVEL) ? Block::LAVA : Block::AIR;
            doCaveBlockPostProcess();
            return;
        }

        caveTopDepth = y - (caveLayer.end + 1);
    }

    if (y > height)
    {
        return;
    }

    bool wasBlockPreProcessed = biomeBlockPreProcess(blockPtr, randBiome, worldBlockPos, height);
    if (wasBlockPreProcessed)
    {
        doBlockPostProcess();
        return;
    }

    int layerIdxStart;
    if (y >= shared_layersAndHeight[numForwardMaterials])
    {
        layerIdxStart = numForwardMaterials;
    }
    else
    {
        layerIdxStart = 0;
    }

    int thisLayerIdx = -1;
    for (int layerIdx = layerIdxStart; layerIdx < numMaterials; ++layerIdx)
    {
        float layerStart = shared_layersAndHeight[layerIdx];
        float layerEnd = shared_layersAndHeight[layerIdx + 1];

        if (layerStart <= y && y < layerEnd)
        {
            thisLayerIdx = layerIdx;
            break;
        }
    }

    // CANONICAL: the reference reads dev_materialInfos[-1] when no layer contains y (chunk.cu:1349-1363), which happens when
    // y == height exactly; canonical block is STONE. Counted.
    CANONICAL_NO_LAYER_FOUND(thisLayerIdx, blockPtr)
    *blockPtr = dev_materialInfos[thisLayerIdx].block;

    if (isTopBlock)
    {
        if (*blockPtr == Block::DIRT)
        {
            *blockPtr = dev_biomeBlocks[(int)randBiome].grassBlock;
        }
    }

    doBlockPostProcess();
    doCaveBlockPostProcess();

#undef doBlockPostProcess
#undef doCaveBlockPostProcess
#undef postProcessCaveBiome
}

// ===================================================================================================
// K6 — kernFill chunk.cu:1382-1510 + host side of Chunk::fill chunk.cu:1555-1601
// ===================================================================================================
void fillChunk(ivec3 chunkWorldBlockPos, const float* heightfield, const float* biomeWeights, const float* layers, const CaveLayer* caveLayers,
               const FeaturePlacement* features, int nFeatures, const CaveFeaturePlacement* caveFeatures, int nCaveFeatures, Block* blocks)
{
    const Tables& t = T();
    // host: union of height bounds over the un-truncated gathered lists (chunk.cu:1555-1570)
    ivec2 allFeaturesHeightBounds = {384, -1};
    for (int i = 0; i < nFeatures; ++i) {
        const ivec2 b = t.featureHeightBounds[(int)features[i].feature];
        allFeaturesHeightBounds.x = g_min(allFeaturesHeightBounds.x, features[i].pos.y + b.x);
        allFeaturesHeightBounds.y = g_max(allFeaturesHeightBounds.y, features[i].pos.y + b.y);
    }
    ivec2 allCaveFeaturesHeightBounds = {384, -1};
    for (int i = 0; i < nCaveFeatures; ++i) {
        const ivec2 b = t.caveFeatureHeightBounds[(int)caveFeatures[i].feature];
        const int featureY = caveFeatures[i].pos.y;
        allCaveFeaturesHeightBounds.x = g_min(allCaveFeaturesHeightBounds.x, featureY + b.x);
        allCaveFeaturesHeightBounds.y = g_max(allCaveFeaturesHeightBounds.y, featureY + caveFeatures[i].layerHeight + b.y);
    }
    // truncation (chunk.cu:1573-1601): at most 2048 / 4096 entries are visible to the kernel
    const int nF = g_min(nFeatures, MAX_GATHERED_FEATURES_PER_CHUNK);
    const int nCF = g_min(nCaveFeatures, MAX_GATHERED_CAVE_FEATURES_PER_CHUNK);

    for (int z = 0; z < 16; ++z) {
        for (int x = 0; x < 16; ++x) {
            const int idx2d = posTo2dIndex16(x, z);
            float colBiomeWeights[numBiomes];
            float layersAndHeight[numMaterials + 1];
            for (int b = 0; b < numBiomes; ++b) colBiomeWeights[b] = biomeWeights[256 * b + idx2d];
            for (int l = 0; l < numMaterials; ++l) layersAndHeight[l] = layers[256 * l + idx2d];
            layersAndHeight[numMaterials] = heightfield[idx2d];
            const CaveLayer* colCaveLayers = caveLayers + MAX_CAVE_LAYERS_PER_COLUMN * idx2d;
            const float height = layersAndHeight[numMaterials];

            for (int y = 0; y < 384; ++y) {
                const ivec3 worldBlockPos = chunkWorldBlockPos + ivec3{x, y, z};
                Rng rng = makeSeededRandomEngine(worldBlockPos.x, worldBlockPos.y, worldBlockPos.z);

                Block block = Block::AIR;
                chunkFillPlaceBlock(&block, colBiomeWeights, layersAndHeight, colCaveLayers, y, height, worldBlockPos, rng);

                bool isInFeatureBounds = y >= allFeaturesHeightBounds.x && y <= allFeaturesHeightBounds.y;
                bool isInCaveFeatureBounds = y >= allCaveFeaturesHeightBounds.x && y <= allCaveFeaturesHeightBounds.y;

                Block featureBlock = Block::AIR;
                bool placedFeature = false;
                if (isInFeatureBounds) {
                    for (int i = 0; i < nF; ++i) {
                        const FeaturePlacement& fp = features[i];
                        if (fp.feature == Feature::NONE) break;
                        if (block != Block::AIR && !fp.canReplaceBlocks) continue;
                        const ivec2 b = t.featureHeightBounds[(int)fp.feature];
                        if (y < b.x + fp.pos.y || y > b.y + fp.pos.y) continue;
                        if (placeFeature(fp, worldBlockPos, &featureBlock)) { placedFeature = true; break; }
                    }
                }
                if (isInCaveFeatureBounds && !placedFeature) {
                    for (int i = 0; i < nCF; ++i) {
                        const CaveFeaturePlacement& cfp = caveFeatures[i];
                        if (cfp.feature == CaveFeature::NONE) break;
                        if (block != Block::AIR && !cfp.canReplaceBlocks) continue;
                        const int featureY = cfp.pos.y;
                        const ivec2 b = t.caveFeatureHeightBounds[(int)cfp.feature];
                        if (y < featureY + b.x || y > featureY + cfp.layerHeight + b.y) continue;
                        if (placeCaveFeature(cfp, worldBlockPos, &featureBlock)) { placedFeature = true; break; }
                    }
                }
                if (placedFeature) block = featureBlock;
                blocks[y + 384 * idx2d] = block;
            }
        }
    }
}

// ===================================================================================================
// D1 — tryPlaceSingleDecorator chunk.cu:1634-1677, placeDecorators chunk.cu:1679-1747
// ===================================================================================================
void Chunk::tryPlaceSingleDecorator(ivec3 pos, const DecoratorGen& gen)
{
    // CANONICAL: a ceiling decorator of a cave layer that is open to the sky has pos.y == 384 (chunk.cu:1728 with
    // caveLayer.end == 384); the reference then indexes the next column's bedrock (rejected: not replaceable) or, for
    // the last column, reads past the array. Canonical = no-op. Counted.
    CANONICAL_DECORATOR_RANGE(pos);

    const int decoratorIdx = posTo3dIndex(pos);
    Block& currentBlock = this->blocks[decoratorIdx];
    if (!gen.possibleReplaceBlocks.empty()
        && gen.possibleReplaceBlocks.find(currentBlock) == gen.possibleReplaceBlocks.end())
    {
        return;
    }

    int underBlockOffset = gen.generatesFromCeiling ? 1 : -1;
    if (!isInRange(pos.y + underBlockOffset, 0, 383))
    {
        return;
    }

    const Block underBlock = blocks[decoratorIdx + underBlockOffset];
    if ((int)underBlock < numNonSolidBlocks
        || (!gen.possibleUnderBlocks.empty() && gen.possibleUnderBlocks.find(underBlock) == gen.possibleUnderBlocks.end()))
    {
        return;
    }

    if (gen.secondDecoratorBlock != Block::AIR)
    {
        int overBlockOffset = -underBlockOffset;
        if (!isInRange(pos.y + overBlockOffset, 0, 383))
        {
            return;
        }

        Block& overBlock = this->blocks[decoratorIdx + overBlockOffset];
        if (!gen.possibleReplaceBlocks.empty() && gen.possibleReplaceBlocks.find(overBlock) == gen.possibleReplaceBlocks.end())
        {
            return;
        }

        overBlock = gen.secondDecoratorBlock;
    }

    currentBlock = gen.decoratorBlock;
}

void Chunk::placeDecorators()
{
    auto rng = makeSeededRandomEngine(this->worldBlockPos.x, this->worldBlockPos.y, this->worldBlockPos.z, 7589341);
    uniform_real_distribution<float> u01(0, 1);

    for (int z = 0; z < 16; ++z)
    {
        for (int x = 0; x < 16; ++x)
        {
            const int idx2d = posTo2dIndex(x, z);

            const float* columnBiomeWeights = biomeWeights.data() + idx2d;
            Biome biome = getRandomBiome<256>(columnBiomeWeights, u01(rng));

            float rand = u01(rng);
            const auto& biomeDecoratorGens = host_biomeDecoratorGens[(int)biome];
            for (int genIdx = 0; genIdx < biomeDecoratorGens.size(); ++genIdx)
            {
                const auto& gen = biomeDecoratorGens[genIdx];

                if ((rand -= gen.chance) < 0.f)
                {
                    tryPlaceSingleDecorator(ivec3(x, ((int)this->heightfield[idx2d]) + 1, z), gen);
                    break;
                }
            }

            const CaveLayer* columnCaveLayers = this->caveLayers.data() + (MAX_CAVE_LAYERS_PER_COLUMN * idx2d);
            for (int caveLayerIdx = 0; caveLayerIdx < MAX_CAVE_LAYERS_PER_COLUMN; ++caveLayerIdx)
            {
                const auto& caveLayer = columnCaveLayers[caveLayerIdx];

                if (caveLayer.start == 384)
                {
                    break;
                }

                float bottomRand = u01(rng);
                float topRand = u01(rng);
                // placedBottom / placedTop are never set (chunk.cu:1718-1743): every gen whose cumulative chance is passed fires
                bool placedBottom = false;
                bool placedTop = false;
                const auto& caveBiomeDecoratorGens = host_caveBiomeDecoratorGens[(int)caveLayer.bottomBiome];
                for (int genIdx = 0; genIdx < caveBiomeDecoratorGens.size(); ++genIdx)
                {
                    const auto& gen = caveBiomeDecoratorGens[genIdx];
                    if (gen.generatesFromCeiling)
                    {
                        if (!placedTop && (topRand -= gen.chance) < 0.f)
                        {
                            tryPlaceSingleDecorator(ivec3(x, caveLayer.end, z), gen);
                        }
                    }
                    else
                    {
                        if (!placedBottom && (bottomRand -= gen.chance) < 0.f)
                        {
                            tryPlaceSingleDecorator(ivec3(x, caveLayer.start + 1, z), gen);
                        }
                    }

                    if (placedTop && placedBottom)
                    {
                        break;
                    }
                }
            }
        }
    }
}

void placeDecorators(ivec3 chunkWorldBlockPos, const float* heightfield, const float* biomeWeights, const CaveLayer* caveLayers, Block* blocks)
{
    std::vector<FeaturePlacement> none;
    std::vector<CaveFeaturePlacement> caveNone;
    Chunk chunk{chunkWorldBlockPos, {heightfield}, {biomeWeights}, {nullptr}, {caveLayers}, {blocks}, none, caveNone};
    chunk.placeDecorators();
}

}  // namespace mmo
