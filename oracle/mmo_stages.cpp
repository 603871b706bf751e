// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// Restatement of the kernels and host stages of src/terrain/chunk.cu (file:line cited per function).
#include "mmo_stages.h"
#include <cstring>
#include <algorithm>
#include <array>
#include <memory>
#include <cmath>

namespace mmo {

UbCounters g_ub = {0, 0, 0};

static inline int posTo2dIndex16(int x, int z) { return x + 16 * z; }
template <int xSize = 16> static inline int posTo2dIndex(const int x, const int z) { return x + xSize * z; }          // biomeFuncs.hpp:11-16
template <int xSize = 16> static inline int posTo2dIndex(const ivec2 pos) { return posTo2dIndex<xSize>(pos.x, pos.y); }   // biomeFuncs.hpp:18-23
template <int xSize = 16, int ySize = 384> static inline int posTo3dIndex(const int x, const int y, const int z) { return y + ySize * posTo2dIndex<xSize>(x, z); }
template <int xSize = 16, int ySize = 384> static inline int posTo3dIndex(const ivec3 pos) { return posTo3dIndex<xSize, ySize>(pos.x, pos.y, pos.z); }   // biomeFuncs.hpp:25-37
// The bodies of the reference's CUDA kernels are restated below as they are written, per thread: the thread / block indices are plain
// values handed in by the caller (which plays the launch), __shared__ arrays are the caller's, and a barrier is the boundary between
// two functions of the same kernel (every thread of the block runs the first, then every thread runs the second).
struct dim3 { int x = 1, y = 1, z = 1; };
#define __syncthreads() ((void)0)
// staging sizes of terrain.hpp:40-51
static constexpr int devHeightfieldSize = 18 * 18;
static constexpr int devBiomeWeightsSize = 256 * numBiomes;
static constexpr int devLayersSize = 256 * numMaterials;
#define EROSION_GRID_SIDE_LENGTH_BLOCKS EROSION_GRID_SIDE
// the three places where the canonical semantics add a statement to a member function of the reference; all are named
// macros so that tools/extract_ref_literals.py can list and drop exactly these
#define CANONICAL_RETURN_FALSE return false
#define CANONICAL_NO_LAYER_FOUND(idx, blockPtr) if ((idx) < 0) { ++g_ub.noLayerFound; *(blockPtr) = Block::STONE; } else
#define CANONICAL_DECORATOR_RANGE(pos) if ((pos).y < 0 || (pos).y > 383) { ++g_ub.decoratorOutOfRange; return; }
// the reference's __constant__ / host copies of BiomeUtils' tables (biomeFuncs.hpp:709-723)
#define dev_materialInfos (T().materialInfos)
#define dev_biomeMaterialWeights (T().biomeMaterialWeights)
#define dev_dirVecs2d (T().dirVecs2d)
#define dev_featureHeightBounds (T().featureHeightBounds)
#define dev_caveFeatureHeightBounds (T().caveFeatureHeightBounds)
#define host_featureHeightBounds (T().featureHeightBounds)
#define host_caveFeatureHeightBounds (T().caveFeatureHeightBounds)
#define dev_biomeBlocks (T().biomeBlocks)
#define host_biomeFeatureGens (T().biomeFeatureGens)
#define host_caveBiomeFeatureGens (T().caveBiomeFeatureGens)
#define host_biomeDecoratorGens (T().biomeDecoratorGens)
#define host_caveBiomeDecoratorGens (T().caveBiomeDecoratorGens)
static inline int posTo2dIndex18(int x, int z) { return x + 18 * z; }

// ===================================================================================================
// K1 — kernGenerateHeightfield chunk.cu:150-185
// ===================================================================================================
// the body of the kernel for the thread of column (x, z) of chunk chunkIdx (chunk.cu:156-184; threads are 1 x 16 x 16 per chunk)
static void kernGenerateHeightfield(const ivec2* chunkWorldBlockPositions, float* heightfield, float* biomeWeights, const int chunkIdx, const int x, const int z)
{
    const int idx = posTo2dIndex(x, z);

    const vec2 worldPos = chunkWorldBlockPositions[chunkIdx] + ivec2(x, z);
    const auto biomeNoise = getBiomeNoise(worldPos);

    float* columnBiomeWeights = biomeWeights + (devBiomeWeightsSize * chunkIdx) + (idx);
    float height = 0.f;
    for (int biomeIdx = 0; biomeIdx < numBiomes; ++biomeIdx)
    {
        Biome biome = (Biome)biomeIdx;

        float weight = getBiomeWeight(biome, biomeNoise);
        if (weight > 0.f)
        {
            height += weight * getHeight(biome, worldPos);
        }

        columnBiomeWeights[256 * biomeIdx] = weight;
    }

    heightfield[(256 * chunkIdx) + idx] = height;
}

// one column on its own (the slope ring of G1: a neighbour's height is the kernel's value for that world position)
float columnHeight(ivec2 worldPosI, float* weights24)
{
    float height, weights[256 * (numBiomes - 1) + 1];
    kernGenerateHeightfield(&worldPosI, &height, weights, 0, 0, 0);
    if (weights24) for (int biomeIdx = 0; biomeIdx < numBiomes; ++biomeIdx) weights24[biomeIdx] = weights[256 * biomeIdx];
    return height;
}

void generateHeightfield(ivec2 chunkWorldBlockPos, float* heightfield, float* biomeWeights)
{
    for (int z = 0; z < 16; ++z)
        for (int x = 0; x < 16; ++x) kernGenerateHeightfield(&chunkWorldBlockPos, heightfield, biomeWeights, 0, x, z);
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
        vec2 noisePos = worldPos * materialInfo.noiseScaleOrMaxSlope + vec2(layerIdx * 5283.64f);
        return g_max(0.f, materialInfo.thickness + materialInfo.noiseAmplitudeOrTanAngleOfRepose * fbm(noisePos)) * materialWeight;
    }
    else return 0;
}

// the body of the kernel after its barrier, for the thread of column (x, z) (chunk.cu:346-414); shared_heightfield = the chunk's 18 x 18
// gathered heights, staged by the block before the barrier (chunk.cu:337-344: a copy)
static void kernGenerateLayers(const float* shared_heightfield, const float* biomeWeights, const ivec2* chunkWorldBlockPositions, float* layers,
                               const int chunkIdx, const int x, const int z)
{
    const int idx = posTo2dIndex(x, z);

    const vec2 worldPos = chunkWorldBlockPositions[chunkIdx] + ivec2(x, z);

    float totalMaterialWeights[numMaterials];
    for (int materialIdx = 0; materialIdx < numMaterials; ++materialIdx)
    {
        totalMaterialWeights[materialIdx] = 0;
    }

    const float* columnBiomeWeights = biomeWeights + (devBiomeWeightsSize * chunkIdx) + (idx);
    for (int biomeIdx = 0; biomeIdx < numBiomes; ++biomeIdx)
    {
        const float biomeWeight = columnBiomeWeights[256 * biomeIdx];

        for (int materialIdx = 0; materialIdx < numMaterials; ++materialIdx)
        {
            totalMaterialWeights[materialIdx] += biomeWeight * dev_biomeMaterialWeights[posTo2dIndex<numMaterials>(materialIdx, biomeIdx)];
        }
    }

    const ivec2 pos18 = ivec2(x + 1, z + 1);
    const float maxHeight = shared_heightfield[posTo2dIndex<18>(pos18)];

    float slope = 0;
    for (int i = 0; i < 8; ++i)
    {
        float neighborHeight = shared_heightfield[posTo2dIndex<18>(pos18 + dev_dirVecs2d[i])];
        slope = g_max(slope, fabsf(neighborHeight - maxHeight) * (i % 2 == 1 ? SQRT_2 : 1));
    }

    float* columnLayers = layers + (devLayersSize * chunkIdx) + (idx);

    float height = 0;
    for (int layerIdx = 0; layerIdx < numForwardMaterials; ++layerIdx)
    {
        columnLayers[256 * layerIdx] = height;

        if (height > maxHeight || layerIdx == numForwardMaterials - 1)
        {
            break;
        }

        height += getStratifiedMaterialThickness(layerIdx, totalMaterialWeights[layerIdx], worldPos);
    }

    height = 0;
    for (int layerIdx = numStratifiedMaterials - 1; layerIdx >= numForwardMaterials; --layerIdx)
    {
        height += getStratifiedMaterialThickness(layerIdx, totalMaterialWeights[layerIdx], worldPos);
        columnLayers[256 * layerIdx] = height;
    }

    height = maxHeight;
    for (int layerIdx = numMaterials - 1; layerIdx >= numStratifiedMaterials; --layerIdx)
    {
        const auto& materialInfo = dev_materialInfos[layerIdx];

        float materialWeight = totalMaterialWeights[layerIdx];
        float layerHeight = g_max(0.f, materialInfo.thickness * ((materialInfo.noiseScaleOrMaxSlope - slope) / materialInfo.noiseScaleOrMaxSlope)) * materialWeight;

        height -= layerHeight;
        columnLayers[256 * layerIdx] = height;
    }
}

void generateLayers(ivec2 chunkWorldBlockPos, const float* gatheredHeightfield, const float* biomeWeights, float* layers)
{
    for (int z = 0; z < 16; ++z) {
        for (int x = 0; x < 16; ++x) {
            // CANONICAL: forward layers after the early break are never written by the reference (stale device memory, chunk.cu:383-393);
            // they cannot influence blocks (the layer search in fill finds an earlier layer first), so the canonical value is the running
            // height, i.e. the last value the loop did write.  The kernel body is run as it is written on a column of "not written" marks.
            float* columnLayers = layers + posTo2dIndex(x, z);
            for (int layerIdx = 0; layerIdx < numForwardMaterials; ++layerIdx) columnLayers[256 * layerIdx] = NAN;
            kernGenerateLayers(gatheredHeightfield, biomeWeights, &chunkWorldBlockPos, layers, 0, x, z);
            for (int layerIdx = 1; layerIdx < numForwardMaterials; ++layerIdx)
                if (std::isnan(columnLayers[256 * layerIdx])) columnLayers[256 * layerIdx] = columnLayers[256 * (layerIdx - 1)];
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
static constexpr int gatheredLayersBaseSize = EROSION_GRID_NUM_COLS * (numErodedMaterials + 1); // +1 for heightfield

// kernDoErosion up to its first barrier, for one thread of a 32 x 32 block (chunk.cu:487-556): the thread's indices, its own cell and
// (threads 0 - 131) one cell of the 34 x 34 tile's border into the block's shared tiles.  CANONICAL: gatheredLayers / accumulatedHeights
// are the SNAPSHOT of the state before the pass.
static void kernDoErosion_stage(const float* gatheredLayers, const float* accumulatedHeights, int layerIdx, bool isFirst,
                                float* shared_layerStart, float* shared_layerEnd, bool& shared_didChange,
                                const dim3& threadIdx, const dim3& blockIdx, const dim3& blockDim)
{
    const int localX = threadIdx.x;
    const int localZ = threadIdx.y;
    const int localIdx2d = posTo2dIndex<32>(localX, localZ);

    const int blockStartX = (blockIdx.x * blockDim.x);
    const int blockStartZ = (blockIdx.y * blockDim.y);

    const int globalX = blockStartX + localX;
    const int globalZ = blockStartZ + localZ;
    const int globalIdx2d = posTo2dIndex<EROSION_GRID_SIDE_LENGTH_BLOCKS>(globalX, globalZ);

    if (localIdx2d == 0)
    {
        shared_didChange = false;
    }

    const ivec2 sharedLayerPos = ivec2(localX + 1, localZ + 1);
    const int sharedLayerIdx = posTo2dIndex<34>(sharedLayerPos);
    const int gatheredLayersIdx = globalIdx2d + (EROSION_GRID_NUM_COLS * layerIdx);

    float thisAccumulatedHeight = isFirst ? accumulatedHeights[globalIdx2d] : 0;

    const float thisLayerStart = gatheredLayers[gatheredLayersIdx] + thisAccumulatedHeight;
    const float thisLayerEnd = gatheredLayers[gatheredLayersIdx + EROSION_GRID_NUM_COLS] + thisAccumulatedHeight;
    shared_layerStart[sharedLayerIdx] = thisLayerStart;
    shared_layerEnd[sharedLayerIdx] = thisLayerEnd;

    ivec2 storePos = ivec2(-1);
    if (localIdx2d < 64)
    {
        storePos = ivec2((localIdx2d % 32) + 1, localIdx2d < 32 ? 0 : 33);
    }
    else if (localIdx2d < 128)
    {
        storePos = ivec2(localIdx2d < 96 ? 0 : 33, (localIdx2d % 32) + 1);
    }
    else
    {
        switch (localIdx2d)
        {
        case 128:
            storePos = ivec2(0, 0);
            break;
        case 129:
            storePos = ivec2(33, 0);
            break;
        case 130:
            storePos = ivec2(0, 33);
            break;
        case 131:
            storePos = ivec2(33, 33);
            break;
        }
    }

    if (storePos.x != -1)
    {
        ivec2 loadPos = ivec2(blockStartX - 1, blockStartZ - 1) + storePos;
        loadPos = g_clamp(loadPos, 0, EROSION_GRID_SIDE_LENGTH_BLOCKS - 1);

        const int loadIdx2d = posTo2dIndex<EROSION_GRID_SIDE_LENGTH_BLOCKS>(loadPos);
        const int loadIdx = loadIdx2d + (EROSION_GRID_NUM_COLS * layerIdx);
        const int storeIdx = posTo2dIndex<34>(storePos);

        thisAccumulatedHeight = isFirst ? accumulatedHeights[loadIdx2d] : 0;

        shared_layerStart[storeIdx] = gatheredLayers[loadIdx] + thisAccumulatedHeight;
        shared_layerEnd[storeIdx] = gatheredLayers[loadIdx + EROSION_GRID_NUM_COLS] + thisAccumulatedHeight;
    }

    __syncthreads();
}

// kernDoErosion between its two barriers (chunk.cu:558-591): the relaxation of the thread's own cell against the shared tiles, written
// to the live planes.  (The thread's indices are derived again; its own start / end are what it staged.)
static void kernDoErosion_relax(float* gatheredLayers, float* accumulatedHeights, int layerIdx, const float* shared_layerStart, const float* shared_layerEnd,
                                bool& shared_didChange, const dim3& threadIdx, const dim3& blockIdx, const dim3& blockDim)
{
    const int localX = threadIdx.x, localZ = threadIdx.y;
    const int globalIdx2d = posTo2dIndex<EROSION_GRID_SIDE_LENGTH_BLOCKS>((blockIdx.x * blockDim.x) + localX, (blockIdx.y * blockDim.y) + localZ);
    const ivec2 sharedLayerPos = ivec2(localX + 1, localZ + 1);
    const int gatheredLayersIdx = globalIdx2d + (EROSION_GRID_NUM_COLS * layerIdx);
    const float thisLayerStart = shared_layerStart[posTo2dIndex<34>(sharedLayerPos)];
    const float thisLayerEnd = shared_layerEnd[posTo2dIndex<34>(sharedLayerPos)];

    float newLayerStart = thisLayerStart;
    float maxThickness = thisLayerEnd - thisLayerStart;
    const float tanAngleOfRepose = dev_materialInfos[numStratifiedMaterials + layerIdx].noiseAmplitudeOrTanAngleOfRepose;

    for (int i = 0; i < 8; ++i)
    {
        const auto& neighborDir = dev_dirVecs2d[i];
        int neighborIdx = posTo2dIndex<34>(sharedLayerPos + neighborDir);

        float neighborLayerStart = shared_layerStart[neighborIdx];
        newLayerStart = g_max(newLayerStart, neighborLayerStart - tanAngleOfRepose * (i % 2 == 1 ? SQRT_2 : 1));

        maxThickness = g_max(maxThickness, shared_layerEnd[neighborIdx] - neighborLayerStart);
    }

    newLayerStart = g_min(newLayerStart, thisLayerEnd);

    if (maxThickness > 0)
    {
        gatheredLayers[gatheredLayersIdx] = newLayerStart;

        if (newLayerStart != thisLayerStart)
        {
            shared_didChange = true;

            accumulatedHeights[globalIdx2d] += newLayerStart - thisLayerStart;
        }
    }

    __syncthreads();
}

// The host loop of Chunk::erodeZone (chunk.cu:672-705) over the kernel above, one launch = one pass; the launch is played block by
// block (12 x 12 blocks of 32 x 32 threads), every block staging from the snapshot taken before the pass.
int erodeZonePlanes(float* g)
{
    const int N = EROSION_GRID_NUM_COLS;
    std::vector<float> accumulatedHeights(N, 0.f), snapshot(g, g + (size_t)gatheredLayersBaseSize), accumulatedSnapshot(N);
    std::vector<float> shared_layerStart(34 * 34), shared_layerEnd(34 * 34);
    const dim3 blockSize2d{32, 32, 1};
    constexpr int blocksPerGrid = (ZONE_SIZE * 2 * 16) / 32;
    int passes = 0;

    for (int layerIdx = numErodedMaterials - 1; layerIdx >= 0; --layerIdx) {
        bool isFirst = true;
        bool flagDidChange;
        do {
            flagDidChange = false;
            // the two planes this pass reads: the layer's own start plane and its end plane (the eroded start plane of the layer above)
            std::memcpy(snapshot.data() + (size_t)N * layerIdx, g + (size_t)N * layerIdx, sizeof(float) * 2 * N);
            accumulatedSnapshot = accumulatedHeights;
            for (int by = 0; by < blocksPerGrid; ++by) {
                for (int bx = 0; bx < blocksPerGrid; ++bx) {
                    const dim3 blockIdx{bx, by, 0};
                    bool shared_didChange = false;
                    for (int ty = 0; ty < 32; ++ty)
                        for (int tx = 0; tx < 32; ++tx)
                            kernDoErosion_stage(snapshot.data(), accumulatedSnapshot.data(), layerIdx, isFirst, shared_layerStart.data(), shared_layerEnd.data(),
                                                shared_didChange, dim3{tx, ty, 0}, blockIdx, blockSize2d);
                    for (int ty = 0; ty < 32; ++ty)
                        for (int tx = 0; tx < 32; ++tx)
                            kernDoErosion_relax(g, accumulatedHeights.data(), layerIdx, shared_layerStart.data(), shared_layerEnd.data(), shared_didChange,
                                                dim3{tx, ty, 0}, blockIdx, blockSize2d);
                    if (shared_didChange) flagDidChange = true;          // chunk.cu:596-599: the block's flag into the launch's
                }
            }
            isFirst = false;
            ++passes;
        } while (flagDidChange);
    }
    return passes;
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
// K4 — kernGenerateCaves chunk.cu:812-937, restated per thread: one block = one column (blockDim = (1, 384, 1), thread y = voxel y), one
// function per barrier-separated phase; the caller runs every thread of the block through a phase before the next one starts.
// CANONICAL where the reference is schedule dependent: the threads run in ascending y, so the shared-memory atomicAdd of
// the 8 ocean + beach weights (chunk.cu:846-850) sums in ascending biome order from 0.f; the layer slots are complete before the biome
// phase reads them (the reference has no barrier there and relies on the 32 threads being one warp); flips beyond 32 layers, which the
// reference stores into the next column's slots (chunk.cu:902-907), are dropped and counted.
// ===================================================================================================
struct CaveBlock {
    float shared_maxHeight, shared_oceanAndBeachWeight;
    int shared_isFilled[384], shared_flipHeights[384];
    int warp_numFlips[32];                                   // what __shfl_sync hands the first warp's lanes
};
#define atomicAdd(ptr, value) (*(ptr) += (value))
#define __shfl_sync(mask, var, srcLane) (warp_##var[srcLane])
#define CANONICAL_CAVE_LAYER_OVERFLOW(storeIdx) if ((storeIdx) >= 3 * MAX_CAVE_LAYERS_PER_COLUMN) { ++g_ub.caveLayerOverflow; continue; }
// the thread's indices (chunk.cu:824-835), for the phases whose section starts behind them
#define KGC_THREAD                                                                      \
    const int globalX = (blockIdx.x * blockDim.x) + threadIdx.x;                        \
    const int chunkIdx = globalX / 16;                                                  \
    const int x = globalX - (chunkIdx * 16);                                            \
    const int y = (blockIdx.y * blockDim.y) + threadIdx.y;                              \
    const int z = (blockIdx.z * blockDim.z) + threadIdx.z;                              \
    const int idx2d = posTo2dIndex(x, z);                                               \
    const ivec2 chunkWorldBlockPos2d = chunkWorldBlockPositions[chunkIdx];              \
    const ivec3 worldPos = ivec3(chunkWorldBlockPos2d.x + x, y, chunkWorldBlockPos2d.y + z); \
    (void)idx2d; (void)worldPos;
#define KGC_SHARED                                                                      \
    float& shared_maxHeight = blk.shared_maxHeight;                                     \
    float& shared_oceanAndBeachWeight = blk.shared_oceanAndBeachWeight;                 \
    int* shared_isFilled = blk.shared_isFilled;                                         \
    int* shared_flipHeights = blk.shared_flipHeights;                                   \
    int* warp_numFlips = blk.warp_numFlips;                                             \
    (void)shared_maxHeight; (void)shared_oceanAndBeachWeight; (void)shared_isFilled; (void)shared_flipHeights; (void)warp_numFlips;

static void kernGenerateCaves_init(const float* heightfield, const ivec2* chunkWorldBlockPositions, CaveBlock& blk, const dim3& threadIdx, const dim3& blockIdx,
                                   const dim3& blockDim)
{
    KGC_SHARED
    const int globalX = (blockIdx.x * blockDim.x) + threadIdx.x;

    const int chunkIdx = globalX / 16;
    const int x = globalX - (chunkIdx * 16);
    const int y = (blockIdx.y * blockDim.y) + threadIdx.y;
    const int z = (blockIdx.z * blockDim.z) + threadIdx.z;

    const int idx2d = posTo2dIndex(x, z);

    const ivec2 chunkWorldBlockPos2d = chunkWorldBlockPositions[chunkIdx];
    const ivec3 worldPos = ivec3(chunkWorldBlockPos2d.x + x, y, chunkWorldBlockPos2d.y + z);

    if (y == 0)
    {
        shared_oceanAndBeachWeight = 0.f;
    } else if (y == 1)
    {
        shared_maxHeight = heightfield[256 * chunkIdx + idx2d];
    }

    __syncthreads();
    (void)worldPos;
}

static void kernGenerateCaves_weights(const float* biomeWeights, const ivec2* chunkWorldBlockPositions, CaveBlock& blk, const dim3& threadIdx, const dim3& blockIdx,
                                      const dim3& blockDim)
{
    KGC_SHARED
    KGC_THREAD
    if (y < numOceanAndBeachBiomes)
    {
        float biomeWeight = biomeWeights[devBiomeWeightsSize * chunkIdx + 256 * y + idx2d];
        atomicAdd(&shared_oceanAndBeachWeight, biomeWeight);
    }

    __syncthreads();
}

static void kernGenerateCaves_filled(const ivec2* chunkWorldBlockPositions, CaveBlock& blk, const dim3& threadIdx, const dim3& blockIdx, const dim3& blockDim)
{
    KGC_SHARED
    KGC_THREAD
    int isThisFilled = shouldGenerateCaveAtBlock(worldPos, shared_maxHeight, shared_oceanAndBeachWeight) ? 0 : 1;
    shared_isFilled[y] = isThisFilled;

    __syncthreads();
}

static void kernGenerateCaves_flips(const ivec2* chunkWorldBlockPositions, CaveBlock& blk, const dim3& threadIdx, const dim3& blockIdx, const dim3& blockDim)
{
    KGC_SHARED
    KGC_THREAD
    const int isThisFilled = shared_isFilled[y];
    int isNextFilled = y < 383 ? shared_isFilled[y + 1] : 0;
    shared_flipHeights[y] = (isThisFilled != isNextFilled) ? y : -1;

    __syncthreads();
}

// threads y < 384 / 32: the flips of the thread's 32 voxels moved to the front of its slice; how many there are goes to the shuffle
static void kernGenerateCaves_compact(const ivec2* chunkWorldBlockPositions, CaveBlock& blk, const dim3& threadIdx, const dim3& blockIdx, const dim3& blockDim)
{
    KGC_SHARED
    KGC_THREAD
    if (y >= 384 / 32) return;
        const int startLoadIdx = 32 * y;
        int endLoadIdx = startLoadIdx;
        for (int i = startLoadIdx; i < startLoadIdx + 32; ++i)
        {
            int flipHeight = shared_flipHeights[i];
            if (flipHeight != -1)
            {
                shared_flipHeights[endLoadIdx] = flipHeight;
                ++endLoadIdx;
            }
        }

        const int numFlips = endLoadIdx - startLoadIdx;
    warp_numFlips[y] = numFlips;
}

// threads y < 384 / 32: where the thread's flips go (the flips of the threads below it come first), and the stores into the column's
// layer slots, two ints per layer with the biome word skipped
static void kernGenerateCaves_store(const ivec2* chunkWorldBlockPositions, CaveLayer* caveLayers, CaveBlock& blk, const dim3& threadIdx, const dim3& blockIdx,
                                    const dim3& blockDim)
{
    KGC_SHARED
    KGC_THREAD
    if (y >= 384 / 32) return;
    CaveLayer* columnCaveLayers = caveLayers + (256 * MAX_CAVE_LAYERS_PER_COLUMN * chunkIdx) + (MAX_CAVE_LAYERS_PER_COLUMN * idx2d);
    const int startLoadIdx = 32 * y, numFlips = warp_numFlips[y];
        int startStoreIdx = 0;

        for (int srcLane = 0; srcLane < 12; ++srcLane)
        {
            int srcLaneNumFlips = __shfl_sync(0x00000fffu, numFlips, srcLane);
            if (srcLane < y)
            {
                startStoreIdx += srcLaneNumFlips;
            }
        }

        int* columnCaveLayersInts = (int*)columnCaveLayers;

        for (int i = 0; i < numFlips; ++i)
        {
            int storeIdx = startStoreIdx + i;
            storeIdx += (storeIdx >> 1);
            CANONICAL_CAVE_LAYER_OVERFLOW(storeIdx)
            columnCaveLayersInts[storeIdx] = shared_flipHeights[startLoadIdx + i];
        }
}

// threads y < 32: the cave biomes at the two ends of layer y of the column
static void kernGenerateCaves_biomes(const ivec2* chunkWorldBlockPositions, CaveLayer* caveLayers, CaveBlock& blk, const dim3& threadIdx, const dim3& blockIdx,
                                     const dim3& blockDim)
{
    KGC_SHARED
    KGC_THREAD
    if (y >= MAX_CAVE_LAYERS_PER_COLUMN) return;
    CaveLayer* columnCaveLayers = caveLayers + (256 * MAX_CAVE_LAYERS_PER_COLUMN * chunkIdx) + (MAX_CAVE_LAYERS_PER_COLUMN * idx2d);
    {
        CaveLayer& caveLayer = columnCaveLayers[y];
        const ivec2 worldBlockPos2d = chunkWorldBlockPos2d + ivec2(x, z);

        if (caveLayer.start != 384)
        {
            caveLayer.bottomBiome = getCaveBiome(ivec3(worldBlockPos2d.x, caveLayer.start, worldBlockPos2d.y), shared_maxHeight, 329271348);
        }

        if (caveLayer.end == 384)
        {
            caveLayer.topBiome = CaveBiome::NONE;
        }
        else
        {
            caveLayer.topBiome = getCaveBiome(ivec3(worldBlockPos2d.x, caveLayer.end + 1, worldBlockPos2d.y), shared_maxHeight, 4982921);
        }
    }
}
#undef atomicAdd

// Chunk::generateCaves for one chunk (chunk.cu:970-980): every slot {384, 384}, then the launch, one block per column
void generateCaves(ivec2 chunkWorldBlockPos, const float* heightfield, const float* biomeWeights, CaveLayer* caveLayers)
{
    for (int k = 0; k < 256 * MAX_CAVE_LAYERS_PER_COLUMN; ++k) {
        std::memset(&caveLayers[k], 0, sizeof(CaveLayer));
        caveLayers[k].start = 384;
        caveLayers[k].end = 384;
    }
    const dim3 blockSize3d{1, 384, 1};
    CaveBlock blk;
    for (int z = 0; z < 16; ++z) {
        for (int x = 0; x < 16; ++x) {
            const dim3 blockIdx{x, 0, z};
            auto threads = [&](auto&& phase, int n) { for (int y = 0; y < n; ++y) phase(dim3{0, y, 0}); };
            threads([&](const dim3& t) { kernGenerateCaves_init(heightfield, &chunkWorldBlockPos, blk, t, blockIdx, blockSize3d); }, 384);
            threads([&](const dim3& t) { kernGenerateCaves_weights(biomeWeights, &chunkWorldBlockPos, blk, t, blockIdx, blockSize3d); }, 384);
            threads([&](const dim3& t) { kernGenerateCaves_filled(&chunkWorldBlockPos, blk, t, blockIdx, blockSize3d); }, 384);
            threads([&](const dim3& t) { kernGenerateCaves_flips(&chunkWorldBlockPos, blk, t, blockIdx, blockSize3d); }, 384);
            threads([&](const dim3& t) { kernGenerateCaves_compact(&chunkWorldBlockPos, blk, t, blockIdx, blockSize3d); }, 384 / 32);
            threads([&](const dim3& t) { kernGenerateCaves_store(&chunkWorldBlockPos, caveLayers, blk, t, blockIdx, blockSize3d); }, 384 / 32);
            threads([&](const dim3& t) { kernGenerateCaves_biomes(&chunkWorldBlockPos, caveLayers, blk, t, blockIdx, blockSize3d); }, MAX_CAVE_LAYERS_PER_COLUMN);
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
    Staging<float> heightfield, biomeWeights, layers;            // (copyLayers and fixBackwardStratifiedLayers write the layers)
    Staging<const CaveLayer> caveLayers;
    Staging<Block> blocks;
    std::vector<FeaturePlacement>& featurePlacements;
    std::vector<CaveFeaturePlacement>& caveFeaturePlacements;
    std::vector<FeaturePlacement> gatheredFeaturePlacements;
    std::vector<CaveFeaturePlacement> gatheredCaveFeaturePlacements;

    void fixBackwardStratifiedLayers();

    bool tryGenerateCaveFeaturePlacement(const CaveFeatureGen& caveFeatureGen, const CaveLayer& caveLayer, bool top, int caveFeaturePlacementSeed,
                                         float rand, ivec2 worldBlockPos2d);
    void generateColumnFeaturePlacements(int localX, int localZ);
    void tryPlaceSingleDecorator(ivec3 pos, const DecoratorGen& gen);
    void placeDecorators();
};

// E3 — Chunk::fixBackwardStratifiedLayers chunk.cu:725-749
void Chunk::fixBackwardStratifiedLayers()
{
    std::array<float, 256> erodedStartHeights;

    for (int layerIdx = numForwardMaterials; layerIdx < numStratifiedMaterials; ++layerIdx)
    {
        const int layerIdx256 = 256 * layerIdx;

        for (int localZ = 0; localZ < 16; ++localZ)
        {
            for (int localX = 0; localX < 16; ++localX)
            {
                const int idx2d = posTo2dIndex(localX, localZ);
                float* columnLayers = this->layers.data() + idx2d;

                if (layerIdx == numForwardMaterials)
                {
                    erodedStartHeights[idx2d] = columnLayers[256 * numStratifiedMaterials];
                }

                columnLayers[layerIdx256] = erodedStartHeights[idx2d] - columnLayers[layerIdx256];
            }
        }
    }
}

static std::vector<FeaturePlacement> g_noPlacements;
static std::vector<CaveFeaturePlacement> g_noCavePlacements;
static Chunk layersView(float* heightfield, float* layers)
{
    return Chunk{ivec3(0), {heightfield}, {nullptr}, {layers}, {nullptr}, {nullptr}, g_noPlacements, g_noCavePlacements, {}, {}};
}

void fixBackwardStratifiedLayers(float* layers)
{
    Chunk chunk = layersView(nullptr, layers);
    chunk.fixBackwardStratifiedLayers();
}

// E1 / E3 — copyLayers chunk.cu:603-656.  The reference's Zone as copyLayers sees it (terrain.hpp:28-37).
struct Zone {
    std::vector<Chunk*> gatheredChunks;
    std::array<std::unique_ptr<Chunk>, ZONE_SIZE * ZONE_SIZE> chunks;
};

void copyLayers(Zone* zonePtr, float* gatheredLayers, bool toGatheredLayers)
{
    const int maxDim = toGatheredLayers ? ZONE_SIZE * 2 : ZONE_SIZE;
    const int maxLayerIdx = toGatheredLayers ? numMaterials + 1 : numMaterials;

    for (int chunkZ = 0; chunkZ < maxDim; ++chunkZ)
    {
        for (int chunkX = 0; chunkX < maxDim; ++chunkX)
        {
            Chunk* chunkPtr;
            ivec2 chunkBlockPos;
            if (toGatheredLayers)
            {
                chunkPtr = zonePtr->gatheredChunks[posTo2dIndex<ZONE_SIZE * 2>(chunkX, chunkZ)];
                chunkBlockPos = ivec2(chunkX, chunkZ) * 16;
            }
            else
            {
                chunkPtr = zonePtr->chunks[posTo2dIndex<ZONE_SIZE>(chunkX, chunkZ)].get();
                chunkBlockPos = (ivec2(chunkX, chunkZ) + ivec2(ZONE_SIZE / 2)) * 16;
            }

            for (int layerIdx = numStratifiedMaterials; layerIdx < maxLayerIdx; ++layerIdx)
            {
                for (int blockZ = 0; blockZ < 16; ++blockZ)
                {
                    const int globalBlockZ = chunkBlockPos.y + blockZ;

                    float* srcLayers;
                    if (toGatheredLayers && layerIdx == maxLayerIdx - 1)
                    {
                        srcLayers = chunkPtr->heightfield.data() + (16 * blockZ);
                    }
                    else
                    {
                        srcLayers = chunkPtr->layers.data() + (16 * blockZ) + (256 * layerIdx);
                    }

                    float* dstLayers = gatheredLayers
                        + (chunkBlockPos.x)
                        + (EROSION_GRID_SIDE_LENGTH_BLOCKS * globalBlockZ)
                        + (EROSION_GRID_NUM_COLS * (layerIdx - numStratifiedMaterials));

                    if (!toGatheredLayers)
                    {
                        std::swap(srcLayers, dstLayers);
                    }

                    std::memcpy(dstLayers, srcLayers, 16 * sizeof(float));
                }
            }
        }
    }
}

// copyLayers for a zone whose 24 x 24 gathered chunks / 12 x 12 own chunks live in chunk-major staging arrays (chunkIdx: index of each
// chunk in those arrays, row-major over the zone's grid): toGatheredLayers packs raw layers + heightfields into the zone planes, the
// other direction writes the eroded planes of the centre chunks back into `layers`
void zoneCopyLayers(float* layers, float* heightfields, const int* chunkIdx, float* gatheredLayers, bool toGatheredLayers)
{
    Zone zone;
    std::vector<Chunk> views;
    const int n = toGatheredLayers ? 4 * ZONE_SIZE * ZONE_SIZE : ZONE_SIZE * ZONE_SIZE;
    views.reserve(n);
    for (int i = 0; i < n; ++i)
        views.push_back(layersView(heightfields ? heightfields + (size_t)256 * chunkIdx[i] : nullptr, layers + (size_t)devLayersSize * chunkIdx[i]));
    if (toGatheredLayers) for (int i = 0; i < n; ++i) zone.gatheredChunks.push_back(&views[i]);
    else for (int i = 0; i < n; ++i) zone.chunks[i] = std::make_unique<Chunk>(views[i]);
    copyLayers(&zone, gatheredLayers, toGatheredLayers);
}

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
    Chunk chunk{ivec3(chunkWorldBlockPos.x, 0, chunkWorldBlockPos.y), {(float*)heightfield}, {(float*)biomeWeights}, {(float*)layers}, {caveLayers}, {nullptr}, out, caveOut, {}, {}};
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
// K6 — kernFill chunk.cu:1382-1510 + heightBoundsMinMax :1512-1516 + host side of Chunk::fill :1518-1632
// ===================================================================================================
// the body of the kernel after its barrier, for the thread of voxel (x, y, z) (chunk.cu:1427-1509); the three shared arrays are the
// column's biome weights, its layers + height and its cave layers, staged by the block before the barrier (chunk.cu:1404-1423: copies)
static void kernFill(
    Block* blocks,
    const float* shared_biomeWeights,
    const float* shared_layersAndHeight,
    const CaveLayer* shared_caveLayers,
    const FeaturePlacement* featurePlacements,
    ivec2 allFeaturesHeightBounds,
    const CaveFeaturePlacement* caveFeaturePlacements,
    ivec2 allCaveFeaturesHeightBounds,
    ivec3 chunkWorldBlockPos,
    const int x, const int y, const int z)
{
    const int idx = posTo3dIndex(x, y, z);

    const float height = shared_layersAndHeight[numMaterials];

    const ivec3 worldBlockPos = chunkWorldBlockPos + ivec3(x, y, z);
    auto rng = makeSeededRandomEngine(worldBlockPos.x, worldBlockPos.y, worldBlockPos.z);

    Block block;
    chunkFillPlaceBlock(&block, shared_biomeWeights, shared_layersAndHeight, shared_caveLayers, y, height, worldBlockPos, rng);

    bool isInFeatureBounds = y >= allFeaturesHeightBounds[0] && y <= allFeaturesHeightBounds[1];
    bool isInCaveFeatureBounds = y >= allCaveFeaturesHeightBounds[0] && y <= allCaveFeaturesHeightBounds[1];

    Block featureBlock;
    bool placedFeature = false;
    if (isInFeatureBounds)
    {
        for (int featureIdx = 0; featureIdx < MAX_GATHERED_FEATURES_PER_CHUNK; ++featureIdx)
        {
            const auto& featurePlacement = featurePlacements[featureIdx];

            if (featurePlacement.feature == Feature::NONE)
            {
                break;
            }

            if (block != Block::AIR && !featurePlacement.canReplaceBlocks)
            {
                continue;
            }

            ivec2 featureHeightBounds = dev_featureHeightBounds[(int)featurePlacement.feature] + ivec2(featurePlacement.pos.y);
            if (y < featureHeightBounds[0] || y > featureHeightBounds[1])
            {
                continue;
            }

            if (placeFeature(featurePlacement, worldBlockPos, &featureBlock))
            {
                placedFeature = true;
                break;
            }
        }
    }

    if (isInCaveFeatureBounds && !placedFeature)
    {
        for (int caveFeatureIdx = 0; caveFeatureIdx < MAX_GATHERED_CAVE_FEATURES_PER_CHUNK; ++caveFeatureIdx)
        {
            const auto& caveFeaturePlacement = caveFeaturePlacements[caveFeatureIdx];

            if (caveFeaturePlacement.feature == CaveFeature::NONE)
            {
                break;
            }

            if (block != Block::AIR && !caveFeaturePlacement.canReplaceBlocks)
            {
                continue;
            }

            const int featureY = caveFeaturePlacement.pos.y;
            ivec2 caveFeatureHeightBounds = ivec2(featureY, featureY + caveFeaturePlacement.layerHeight) + dev_caveFeatureHeightBounds[(int)caveFeaturePlacement.feature];
            if (y < caveFeatureHeightBounds[0] || y > caveFeatureHeightBounds[1])
            {
                continue;
            }

            if (placeCaveFeature(caveFeaturePlacement, worldBlockPos, &featureBlock))
            {
                placedFeature = true;
                break;
            }
        }
    }

    if (placedFeature)
    {
        block = featureBlock;
    }

    blocks[idx] = block;
}

void heightBoundsMinMax(ivec2& in, const ivec2& v)
{
    in[0] = g_min(in[0], v[0]);
    in[1] = g_max(in[1], v[1]);
}

// the copies of the reference's host code are plain memory copies here
enum cudaMemcpyKind { cudaMemcpyHostToDevice };
typedef int cudaStream_t;
static inline void cudaMemcpyAsync(void* dst, const void* src, size_t count, cudaMemcpyKind, cudaStream_t) { std::memcpy(dst, src, count); }

// Chunk::fill for chunk i of a batch, from the gathered lists to the launch (chunk.cu:1555-1616): the union of the lists' height bounds
// over the UN-truncated lists, the truncation to 2 048 / 4 096 entries with the NONE terminator, then the kernel over the chunk's voxels
static void Chunk_fill(Chunk* chunkPtr, int i, FeaturePlacement* dev_featurePlacements, CaveFeaturePlacement* dev_caveFeaturePlacements, Block* dev_blocks,
                       cudaStream_t stream)
{
        ivec2 allFeaturesHeightBounds = ivec2(384, -1);
        for (const auto& featurePlacement : chunkPtr->gatheredFeaturePlacements)
        {
            const auto& featureHeightBounds = host_featureHeightBounds[(int)featurePlacement.feature];
            const ivec2 thisFeatureHeightBounds = ivec2(featurePlacement.pos.y) + featureHeightBounds;
            heightBoundsMinMax(allFeaturesHeightBounds, thisFeatureHeightBounds);
        }

        ivec2 allCaveFeaturesHeightBounds = ivec2(384, -1);
        for (const auto& caveFeaturePlacement : chunkPtr->gatheredCaveFeaturePlacements)
        {
            const auto& caveFeatureHeightBounds = host_caveFeatureHeightBounds[(int)caveFeaturePlacement.feature];
            const int featureY = caveFeaturePlacement.pos.y;
            const ivec2 thisCaveFeatureHeightBounds = ivec2(featureY, featureY + caveFeaturePlacement.layerHeight) + caveFeatureHeightBounds;
            heightBoundsMinMax(allCaveFeaturesHeightBounds, thisCaveFeatureHeightBounds);
        }

        int numFeaturePlacements = g_min((int)chunkPtr->gatheredFeaturePlacements.size(), MAX_GATHERED_FEATURES_PER_CHUNK);
        if (numFeaturePlacements < MAX_GATHERED_FEATURES_PER_CHUNK)
        {
            chunkPtr->gatheredFeaturePlacements.push_back({ Feature::NONE });
            ++numFeaturePlacements;
        }
        cudaMemcpyAsync(
            dev_featurePlacements + (i * MAX_GATHERED_FEATURES_PER_CHUNK),
            chunkPtr->gatheredFeaturePlacements.data(),
            numFeaturePlacements * sizeof(FeaturePlacement),
            cudaMemcpyHostToDevice,
            stream
        );
        chunkPtr->gatheredFeaturePlacements.clear();

        int numCaveFeaturePlacements = g_min((int)chunkPtr->gatheredCaveFeaturePlacements.size(), MAX_GATHERED_CAVE_FEATURES_PER_CHUNK);
        if (numCaveFeaturePlacements < MAX_GATHERED_CAVE_FEATURES_PER_CHUNK)
        {
            chunkPtr->gatheredCaveFeaturePlacements.push_back({ CaveFeature::NONE });
            ++numCaveFeaturePlacements;
        }
        cudaMemcpyAsync(
            dev_caveFeaturePlacements + (i * MAX_GATHERED_CAVE_FEATURES_PER_CHUNK),
            chunkPtr->gatheredCaveFeaturePlacements.data(),
            numCaveFeaturePlacements * sizeof(CaveFeaturePlacement),
            cudaMemcpyHostToDevice,
            stream
        );
        chunkPtr->gatheredCaveFeaturePlacements.clear();

        const dim3 blockSize3d{1, 128, 1};
        const dim3 blocksPerGrid3d{16, 3, 16};
        // the launch: one block = 128 voxels of one column; its shared arrays are staged once per column here
        for (int z = 0; z < blocksPerGrid3d.z; ++z) {
            for (int x = 0; x < blocksPerGrid3d.x; ++x) {
                const int idx2d = posTo2dIndex(x, z);
                float shared_biomeWeights[numBiomes];
                float shared_layersAndHeight[numMaterials + 1];
                for (int loadIdx = 0; loadIdx < numBiomes; ++loadIdx) shared_biomeWeights[loadIdx] = chunkPtr->biomeWeights[idx2d + 256 * loadIdx];
                for (int loadIdx = 0; loadIdx < numMaterials; ++loadIdx) shared_layersAndHeight[loadIdx] = chunkPtr->layers[idx2d + 256 * loadIdx];
                shared_layersAndHeight[numMaterials] = chunkPtr->heightfield[idx2d];
                const CaveLayer* shared_caveLayers = chunkPtr->caveLayers.data() + (MAX_CAVE_LAYERS_PER_COLUMN * idx2d);
                for (int y = 0; y < blocksPerGrid3d.y * blockSize3d.y; ++y)
                    kernFill(dev_blocks, shared_biomeWeights, shared_layersAndHeight, shared_caveLayers, dev_featurePlacements + (i * MAX_GATHERED_FEATURES_PER_CHUNK),
                             allFeaturesHeightBounds, dev_caveFeaturePlacements + (i * MAX_GATHERED_CAVE_FEATURES_PER_CHUNK), allCaveFeaturesHeightBounds,
                             chunkPtr->worldBlockPos, x, y, z);
            }
        }
}

void fillChunk(ivec3 chunkWorldBlockPos, const float* heightfield, const float* biomeWeights, const float* layers, const CaveLayer* caveLayers,
               const FeaturePlacement* features, int nFeatures, const CaveFeaturePlacement* caveFeatures, int nCaveFeatures, Block* blocks)
{
    std::vector<FeaturePlacement> none;
    std::vector<CaveFeaturePlacement> caveNone;
    Chunk chunk{chunkWorldBlockPos, {(float*)heightfield}, {(float*)biomeWeights}, {(float*)layers}, {caveLayers}, {blocks}, none, caveNone,
                std::vector<FeaturePlacement>(features, features + nFeatures), std::vector<CaveFeaturePlacement>(caveFeatures, caveFeatures + nCaveFeatures)};
    std::vector<FeaturePlacement> dev_featurePlacements(MAX_GATHERED_FEATURES_PER_CHUNK);
    std::vector<CaveFeaturePlacement> dev_caveFeaturePlacements(MAX_GATHERED_CAVE_FEATURES_PER_CHUNK);
    Chunk_fill(&chunk, 0, dev_featurePlacements.data(), dev_caveFeaturePlacements.data(), blocks, 0);
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
    Chunk chunk{chunkWorldBlockPos, {(float*)heightfield}, {(float*)biomeWeights}, {nullptr}, {caveLayers}, {blocks}, none, caveNone, {}, {}};
    chunk.placeDecorators();
}

}  // namespace mmo
