// mmgen host side — C++ mirror of the reference's `Chunk` / `Zone` interface for the chunk-generation path
// (src/terrain/chunk.hpp:18-176, src/terrain/terrain.hpp:17-50), implemented over the C ABI of libmmgen (include/mmgen.h).
//
// Same class shape, same public data members and layouts, same static stage functions with the same argument order and
// meaning, same ownership (the CALLER owns every pinned-host and device staging buffer and the streams, exactly like
// Terrain::initCuda, terrain.cpp:154-185), same synchronous-on-return behaviour, same error convention (print + exit,
// src/cuda/cuda_utils.cpp:5-17).  Differences, all outside the generation path:
//   * renderer coupling is dropped: no `Drawable` base and no GL buffers (bufferVBOs, chunk.cu:2005-2021); createVBOs() keeps its
//     name, its `verts` / `idx` outputs and their exact contents, but is built on the GPU (mmgen_mesh_count / mmgen_mesh_fill);
//   * stream type is hipStream_t; glm's ivec2/ivec3 are replaced by layout-identical PODs (glm is not a dependency);
//     with -DMMHOST_REFERENCE_TREE the header is instead built inside the reference's tree on the reference's own types (below);
//   * generateFeaturePlacements() and placeDecorators() keep their member signatures but run as device kernels.
#pragma once
#include <array>
#include <vector>
#include <memory>
#include <functional>
#include <hip/hip_runtime.h>
#include "../../include/mmgen.h"

#ifdef MMHOST_REFERENCE_TREE
// ---------------------------------------------------------------------------------------------------------------------------------
// Built INSIDE the reference's source tree in place of its chunk.hpp (tests/refdrop/ builds the reference's own
// unmodified terrain.cpp against this header and runs its Terrain::tick on the MI355X).  The vocabulary is the reference's own:
// glm vectors, `Block` / `CaveLayer` / `FeaturePlacement` / `Vertex` from its block.hpp / biome.hpp / rendering/structs.hpp, `Zone` and
// the dev*Size constants from its terrain.hpp, `Drawable` as the base class, cudaStream_t (= hipStream_t through the name map), the
// global namespace.  Their layouts equal the C ABI's PODs (pinned by oracle/ref_block_probe.cpp -> tests/golden/block_data.npz and
// static_asserts in chunk.cpp), so the ABI calls reinterpret the pointers.
// ---------------------------------------------------------------------------------------------------------------------------------
#include <glm/glm.hpp>
#include "block.hpp"
#include "rendering/drawable.hpp"
#include "rendering/structs.hpp"
#include "biome.hpp"
#include "cuda/cudaUtils.hpp"
#define MMHOST_NS_BEGIN
#define MMHOST_NS_END
using namespace glm;
typedef cudaStream_t mmhostStream;
struct Zone;                                 // terrain.hpp:26-37 (the reference's own definition is used)
namespace HipUtils { void checkError(const char* msg, int code = 0, int line = -1); }
#else
#define MMHOST_NS_BEGIN namespace mmhost {
#define MMHOST_NS_END }
MMHOST_NS_BEGIN

struct ivec2 { int x, y; ivec2() = default; constexpr ivec2(int x, int y) : x(x), y(y) {} };
struct ivec3 { int x, y, z; ivec3() = default; constexpr ivec3(int x, int y, int z) : x(x), y(y), z(z) {} };
inline ivec2 operator+(ivec2 a, ivec2 b) { return {a.x + b.x, a.y + b.y}; }
inline ivec2 operator-(ivec2 a, ivec2 b) { return {a.x - b.x, a.y - b.y}; }
inline bool operator==(ivec2 a, ivec2 b) { return a.x == b.x && a.y == b.y; }
typedef hipStream_t mmhostStream;
typedef unsigned int GLuint;

using Block = uint8_t;                       // enum Block : unsigned char (block.hpp:5-154), ids MMB_*
using CaveLayer = mmgen_cave_layer;          // biome.hpp:106-115
using FeaturePlacement = mmgen_feature_placement;
using CaveFeaturePlacement = mmgen_cave_feature_placement;
using Vertex = mmgen_vertex;                 // rendering/structs.hpp:25-31

constexpr int numMaterials = MMGEN_NUM_MATERIALS, numBiomes = MMGEN_NUM_BIOMES;
constexpr int numStratifiedMaterials = MMGEN_NUM_STRATIFIED_MATERIALS, numForwardMaterials = MMGEN_NUM_FORWARD_MATERIALS;
constexpr int numErodedMaterials = MMGEN_NUM_ERODED_MATERIALS;
#define MAX_CAVE_LAYERS_PER_COLUMN MMGEN_MAX_CAVE_LAYERS_PER_COLUMN
#define ZONE_SIZE MMGEN_ZONE_SIZE
#define EROSION_GRID_SIDE_LENGTH_BLOCKS MMGEN_EROSION_GRID_SIDE
#define EROSION_GRID_NUM_COLS MMGEN_EROSION_GRID_NUM_COLS
constexpr int devBlocksSize = MMGEN_BLOCKS_PER_CHUNK;
constexpr int devFeaturePlacementsSize = MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK;
constexpr int devCaveFeaturePlacementsSize = MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK;
constexpr int devHeightfieldSize = MMGEN_GATHERED_HEIGHTFIELD_SIZE;
constexpr int devBiomeWeightsSize = MMGEN_BIOME_WEIGHTS_SIZE;
constexpr int devLayersSize = MMGEN_LAYERS_SIZE;
constexpr int devCaveLayersSize = MMGEN_CAVE_LAYERS_SIZE;
constexpr int devGatheredLayersSize = MMGEN_GATHERED_LAYERS_SIZE;
constexpr int devAccumulatedHeightsSize = MMGEN_EROSION_GRID_NUM_COLS;

namespace BiomeUtils { void init(); }        // biome.hpp:299-305 → mmgen_init(current device)
namespace HipUtils { void checkError(const char* msg, int code = 0, int line = -1); }   // CudaUtils::checkCUDAError
MMHOST_NS_END
#endif

MMHOST_NS_BEGIN

enum class ChunkState : unsigned char {      // chunk.hpp:18-32
    EMPTY, HAS_HEIGHTFIELD, NEEDS_LAYERS, HAS_LAYERS, NEEDS_EROSION, NEEDS_CAVES, NEEDS_FEATURE_PLACEMENTS,
    NEEDS_GATHER_FEATURE_PLACEMENTS, READY_TO_FILL, FILLED, NEEDS_VBOS, DRAWABLE
};

class Chunk;

#ifndef MMHOST_REFERENCE_TREE
struct Zone {                                // terrain.hpp:26-37
    explicit Zone(ivec2 worldChunkPos) : worldChunkPos(worldChunkPos) {}
    ivec2 worldChunkPos;
    std::array<std::unique_ptr<Chunk>, ZONE_SIZE * ZONE_SIZE> chunks{};
    std::array<Zone*, 8> neighbors{};
    std::vector<Chunk*> gatheredChunks;      // 24 x 24, filled by the scheduler (isZoneReadyForErosion, terrain.cpp:471-522)
    bool hasBeenQueuedForErosion{false};
};
#endif

#ifdef MMHOST_REFERENCE_TREE
class Chunk : public Drawable {
#else
class Chunk {
#endif
    template <std::size_t diameter>
    using ChunkProcessorFunc = std::function<void(Chunk* chunkPtr, Chunk* const (&neighborChunks)[diameter][diameter], int centerX, int centerZ)>;

private:
    ChunkState state{ChunkState::EMPTY};
    bool readyForQueue{true};
    std::vector<FeaturePlacement> featurePlacements;
    std::vector<FeaturePlacement> gatheredFeaturePlacements;
    std::vector<CaveFeaturePlacement> caveFeaturePlacements;
    std::vector<CaveFeaturePlacement> gatheredCaveFeaturePlacements;

public:
    const ivec2 worldChunkPos;
    const ivec3 worldBlockPos;
    Zone* zonePtr{nullptr};
    std::array<Chunk*, 4> neighbors{};       // N (+z), E (+x), S (-z), W (-x)

    std::array<float, 256> heightfield;                                   // iteration order z, x
    std::vector<float> gatheredHeightfield;
    std::array<float, 256 * numMaterials> layers;                         // y, z, x
    // RAW (pre-erosion) copy of the 8 eroded-layer planes, kept so that a neighbouring zone's erosion padding never sees this
    // chunk's eroded result (canonical raw-padding semantics; the reference reads `layers`, chunk.cu:638, and is
    // therefore dependent on the order in which the player's movement erodes zones)
    std::array<float, 256 * numErodedMaterials> rawErodedLayers;
    std::array<CaveLayer, 256 * MAX_CAVE_LAYERS_PER_COLUMN> caveLayers;   // z, x, y
    std::array<float, 256 * numBiomes> biomeWeights;                      // y, z, x
    std::array<Block, 98304> blocks;                                      // z, x, y

    Chunk(ivec2 worldChunkPos);

    ChunkState getState() const;
    void setState(ChunkState newState);
    bool isReadyForQueue();
    void setNotReadyForQueue();

private:
    template <std::size_t diameter> void floodFill(Chunk* (&neighborChunks)[diameter][diameter], ChunkState minState);
    template <std::size_t diameter>
    static void iterateNeighborChunks(Chunk* const (&neighborChunks)[diameter][diameter], ChunkState currentState, ChunkState nextState,
                                      ChunkProcessorFunc<diameter> chunkProcessorFunc);
    template <std::size_t diameter>
    void floodFillAndIterateNeighbors(ChunkState currentState, ChunkState nextState, ChunkProcessorFunc<diameter> chunkProcessorFunc);
    static void otherChunkGatherHeightfield(Chunk* chunkPtr, Chunk* const (&neighborChunks)[5][5], int centerX, int centerZ);
    void fixBackwardStratifiedLayers();
    static void otherChunkGatherFeaturePlacements(Chunk* chunkPtr, Chunk* const (&neighborChunks)[13][13], int centerX, int centerZ);

public:
    static void generateHeightfields(std::vector<Chunk*>& chunks, ivec2* host_chunkWorldBlockPositions, ivec2* dev_chunkWorldBlockPositions,
                                     float* host_heightfields, float* dev_heightfields, float* host_biomeWeights, float* dev_biomeWeights,
                                     mmhostStream stream);
    void gatherHeightfield();
    static void generateLayers(std::vector<Chunk*>& chunks, float* host_heightfields, float* dev_heightfields, float* host_biomeWeights,
                               float* dev_biomeWeights, ivec2* host_chunkWorldBlockPositions, ivec2* dev_chunkWorldBlockPositions,
                               float* host_layers, float* dev_layers, mmhostStream stream);
    static void erodeZone(Zone* zonePtr, float* host_gatheredLayers, float* dev_gatheredLayers, float* dev_accumulatedHeights, mmhostStream stream);
    static void generateCaves(std::vector<Chunk*>& chunks, float* host_heightfields, float* dev_heightfields, float* host_biomeWeights,
                              float* dev_biomeWeights, ivec2* host_chunkWorldBlockPositions, ivec2* dev_chunkWorldBlockPositions,
                              CaveLayer* host_caveLayers, CaveLayer* dev_caveLayers, mmhostStream stream);
    void generateFeaturePlacements();
    void gatherFeaturePlacements();
    static void fill(std::vector<Chunk*>& chunks, float* host_heightfields, float* dev_heightfields, float* host_biomeWeights, float* dev_biomeWeights,
                     float* host_layers, float* dev_layers, CaveLayer* host_caveLayers, CaveLayer* dev_caveLayers,
                     FeaturePlacement* dev_featurePlacements, CaveFeaturePlacement* dev_caveFeaturePlacements, Block* host_blocks, Block* dev_blocks,
                     mmhostStream stream);
    void placeDecorators();

    // Drawable's buffers (rendering/drawable.hpp) as filled by createVBOs (chunk.cu:1778-2003): same order, same bytes
    std::vector<GLuint> idx;
    std::vector<Vertex> verts;
#ifdef MMHOST_REFERENCE_TREE
    void createVBOs();
    void bufferVBOs() override;              // chunk.cu:2005-2021: the renderer's GL upload stays the reference's (not on the generation path)
#else
    int idxCount{0};
    void createVBOs();
#endif

    // test hooks
    const std::vector<FeaturePlacement>& getFeaturePlacements() const { return featurePlacements; }
    const std::vector<CaveFeaturePlacement>& getCaveFeaturePlacements() const { return caveFeaturePlacements; }
};

MMHOST_NS_END
