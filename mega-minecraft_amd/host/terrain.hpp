// mmgen host side — C++ mirror of the reference's `Terrain` world-streaming / action-time scheduler for the generation path
// (src/terrain/terrain.hpp:53-120, src/terrain/terrain.cpp).  Same public interface (init / tick / setCurrentChunkPos /
// getCurrentChunkPos / getDrawableChunks / getMaxNumDrawableChunks), same nine work queues drained latest-stage-first under the
// same action-time budget and costs (terrain.cpp:65-82), same zone bookkeeping (12x12-chunk zones, 8 neighbour links, erosion
// readiness with the 6-chunk padding), same staging-buffer ownership (initCuda, terrain.cpp:111-185 → initHip).  The renderer
// hooks (OptixRenderer, draw, GL buffer upload / destruction) are outside the path; createVBOs builds verts / idx on the GPU.
#pragma once
#include <map>
#include <queue>
#include <set>
#include <unordered_set>
#include "chunk.hpp"

namespace mmhost {

class Terrain {
public:
    // terrain.cpp:65-82
    static constexpr int chunkVbosGenRadius = 16;
    static constexpr int chunkMaxGenRadius = chunkVbosGenRadius + (ZONE_SIZE * 2);
    static constexpr int maxActionTimePerFrame = 500;
    static constexpr int totalActionTimePerSecond = 60 * maxActionTimePerFrame;
    static constexpr int actionTimeGenerateHeightfield = 3;
    static constexpr int actionTimeGatherHeightfield = 2;
    static constexpr int actionTimeGenerateLayers = 5;
    static constexpr int actionTimeErodeZone = maxActionTimePerFrame;
    static constexpr int actionTimeGenerateCaves = 8;
    static constexpr int actionTimeGenerateFeaturePlacements = 3;
    static constexpr int actionTimeGatherFeaturePlacements = 5;
    static constexpr int actionTimeFill = 8;
    static constexpr int actionTimeCreateAndBufferVbos = maxActionTimePerFrame / 3;

    Terrain();
    ~Terrain();
    void init();
    void tick(float deltaTime);
    std::unordered_set<Chunk*> getDrawableChunks();
    ivec2 getCurrentChunkPos() const;
    void setCurrentChunkPos(ivec2 newCurrentChunkPos);
    static int getMaxNumDrawableChunks();

    // headless helpers (the reference's equivalent is DEBUG_TIME_CHUNK_FILL, terrain.cpp:939-959)
    bool allQueuesEmpty() const;
    Chunk* findChunk(ivec2 worldChunkPos);
    size_t numChunks() const;

private:
    std::vector<ivec2> spiral;
    std::map<std::pair<int, int>, std::unique_ptr<Zone>> zones;

    std::queue<Chunk*> chunksToGenerateHeightfield, chunksToGatherHeightfield, chunksToGenerateLayers;
    std::set<Zone*> zonesToTryErosion;
    std::queue<Zone*> zonesToErode;
    std::queue<Chunk*> chunksToGenerateCaves, chunksToGenerateFeaturePlacements, chunksToGatherFeaturePlacements, chunksToFill,
        chunksToCreateAndBufferVbos;
    std::unordered_set<Chunk*> drawableChunks;

    ivec2 currentChunkPos{0, 0}, lastChunkPos{0, 0};
    bool needsUpdateChunks{true};
    Zone* lastUpdateZonePtr{nullptr};
    int actionTimeLeft{0};

    void initHip();
    void freeHip();
    void generateSpiral();
    Zone* createZone(ivec2 zoneWorldChunkPos);
    void updateChunk(int dx, int dz);
    void updateChunks();
    void addZonesToTryErosionSet(Chunk* chunkPtr);
    void updateZones();

    // staging, sized by budget / cost exactly like terrain.cpp:111-129
    Block *host_blocks{}, *dev_blocks{};
    FeaturePlacement* dev_featurePlacements{};
    CaveFeaturePlacement* dev_caveFeaturePlacements{};
    float *host_heightfields{}, *dev_heightfields{}, *host_biomeWeights{}, *dev_biomeWeights{};
    ivec2 *host_chunkWorldBlockPositions{}, *dev_chunkWorldBlockPositions{};
    float *host_layers{}, *dev_layers{};
    CaveLayer *host_caveLayers{}, *dev_caveLayers{};
    float *host_gatheredLayers{}, *dev_gatheredLayers{}, *dev_accumulatedHeights{};
    std::array<hipStream_t, 5> streams{};
};

}  // namespace mmhost
