// mmgen host side — implementation of the `Chunk` stage functions over the libmmgen C ABI.
// Behavioural spec: src/terrain/chunk.cu (host orchestrators :187-229, :231-302, :417-469, :603-749, :939-993, :1147-1196,
// :1518-1632, :1679-1747).  Each static stage keeps the reference's contract: pack into the caller's pinned staging slice →
// H2D → device stage → D2H → unpack into the Chunk members → stream synchronise → error check (print + exit).
#ifdef MMHOST_REFERENCE_TREE
#include "terrain/terrain.hpp"     // the reference's own (Zone, dev*Size); it includes "chunk.hpp" = this tree's header first, like terrain.cpp does
#endif
#include "chunk.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

MMHOST_NS_BEGIN

// the host-side vocabulary (the reference's own structs in MMHOST_REFERENCE_TREE builds) is layout-identical to the C ABI's PODs
static_assert(sizeof(Block) == 1 && sizeof(CaveLayer) == sizeof(mmgen_cave_layer) && sizeof(Vertex) == sizeof(mmgen_vertex), "ABI layout");
static_assert(sizeof(FeaturePlacement) == sizeof(mmgen_feature_placement) && sizeof(CaveFeaturePlacement) == sizeof(mmgen_cave_feature_placement), "ABI layout");
static_assert(sizeof(ivec2) == 8 && sizeof(ivec3) == 12 && sizeof(GLuint) == 4, "ABI layout");
static inline const mmgen_feature_placement& abi(const FeaturePlacement& p) { return reinterpret_cast<const mmgen_feature_placement&>(p); }
static inline const mmgen_cave_feature_placement& abi(const CaveFeaturePlacement& p) { return reinterpret_cast<const mmgen_cave_feature_placement&>(p); }
#define ABI_CL(p) ((mmgen_cave_layer*)(p))
#define ABI_FP(p) ((mmgen_feature_placement*)(p))
#define ABI_CFP(p) ((mmgen_cave_feature_placement*)(p))

// ---------------------------------------------------------------------------------------------------------
// error convention (src/cuda/cuda_utils.cpp:5-17): message to stderr, exit(EXIT_FAILURE)
// ---------------------------------------------------------------------------------------------------------
void HipUtils::checkError(const char* msg, int code, int line)
{
    hipError_t err = code ? (hipError_t)code : hipGetLastError();
    if (err == hipSuccess) return;
    if (line >= 0) std::fprintf(stderr, "Line %d: ", line);
    std::fprintf(stderr, "HIP error: %s: %s.\n", msg, hipGetErrorString(err));
    std::exit(EXIT_FAILURE);
}
#define MM_CALL(expr, what) HipUtils::checkError(what, (int)(expr), __LINE__)

void BiomeUtils::init()      // biomeFuncs.hpp:725 (declared by biome.hpp:299-305)
{
    int dev = 0;
    MM_CALL(hipGetDevice(&dev), "hipGetDevice");
    MM_CALL(mmgen_init(dev), "BiomeUtils::init() failed");
}

namespace {
// library-side scratch of this translation unit (the reference has no slot for these in Terrain::initCuda):
// per-chunk feature bounds for fill, and the single-chunk staging of the two member-function stages.
struct Scratch {
    void* p = nullptr; size_t cap = 0;
    void* get(size_t bytes)
    {
        if (bytes > cap) {
            if (p) MM_CALL(hipFree(p), "hipFree");
            MM_CALL(hipMalloc(&p, bytes), "hipMalloc");
            cap = bytes;
        }
        return p;
    }
};
Scratch g_bounds, g_single;
}  // namespace

Chunk::Chunk(ivec2 worldChunkPos) : worldChunkPos(worldChunkPos), worldBlockPos(worldChunkPos.x * 16, 0, worldChunkPos.y * 16) {}

ChunkState Chunk::getState() const { return state; }
void Chunk::setState(ChunkState newState) { state = newState; readyForQueue = true; }
bool Chunk::isReadyForQueue() { return readyForQueue; }
void Chunk::setNotReadyForQueue() { readyForQueue = false; }

// ---------------------------------------------------------------------------------------------------------
// neighbourhood collection (chunk.cu:52-144): breadth-first walk over the 4-neighbour links, bounded by the window
// ---------------------------------------------------------------------------------------------------------
template <std::size_t diameter>
void Chunk::floodFill(Chunk* (&grid)[diameter][diameter], ChunkState minState)
{
    constexpr int radius = diameter / 2;
    bool seen[diameter][diameter] = {};
    std::vector<Chunk*> frontier{this};
    seen[radius][radius] = true;
    for (std::size_t head = 0; head < frontier.size(); ++head) {
        Chunk* c = frontier[head];
        if (c->getState() < minState) continue;        // not ready: neither recorded nor expanded
        const int gx = c->worldChunkPos.x - worldChunkPos.x + radius, gz = c->worldChunkPos.y - worldChunkPos.y + radius;
        grid[gz][gx] = c;
        for (Chunk* n : c->neighbors) {
            if (!n) continue;
            const int nx = n->worldChunkPos.x - worldChunkPos.x + radius, nz = n->worldChunkPos.y - worldChunkPos.y + radius;
            if (nx < 0 || nz < 0 || nx >= (int)diameter || nz >= (int)diameter || seen[nz][nx]) continue;
            seen[nz][nx] = true;
            frontier.push_back(n);
        }
    }
}

template <std::size_t diameter>
void Chunk::iterateNeighborChunks(Chunk* const (&grid)[diameter][diameter], ChunkState currentState, ChunkState nextState,
                                  ChunkProcessorFunc<diameter> process)
{
    constexpr int k = diameter / 4;                    // diameter = 4k + 1: centres in [k, diameter - k), window radius k
    for (int cz = k; cz < (int)diameter - k; ++cz) {
        for (int cx = k; cx < (int)diameter - k; ++cx) {
            Chunk* c = grid[cz][cx];
            if (!c || c->getState() != currentState) continue;
            bool complete = true;
            for (int dz = -k; dz <= k && complete; ++dz)
                for (int dx = -k; dx <= k && complete; ++dx) complete = grid[cz + dz][cx + dx] != nullptr;
            if (!complete) continue;
            process(c, grid, cx, cz);
            c->setState(nextState);
        }
    }
}

template <std::size_t diameter>
void Chunk::floodFillAndIterateNeighbors(ChunkState currentState, ChunkState nextState, ChunkProcessorFunc<diameter> process)
{
    Chunk* grid[diameter][diameter] = {};
    floodFill<diameter>(grid, currentState);
    iterateNeighborChunks<diameter>(grid, currentState, nextState, process);
}

// ---------------------------------------------------------------------------------------------------------
// heightfield (chunk.cu:187-229) + gathered 18x18 ring (chunk.cu:231-302)
// ---------------------------------------------------------------------------------------------------------
void Chunk::generateHeightfields(std::vector<Chunk*>& chunks, ivec2* host_pos, ivec2* dev_pos, float* host_hf, float* dev_hf, float* host_bw,
                                 float* dev_bw, mmhostStream stream)
{
    const int n = (int)chunks.size();
    for (int i = 0; i < n; ++i) host_pos[i] = ivec2(chunks[i]->worldBlockPos.x, chunks[i]->worldBlockPos.z);
    MM_CALL(hipMemcpyAsync(dev_pos, host_pos, n * sizeof(ivec2), hipMemcpyHostToDevice, stream), "H2D positions");
    MM_CALL(mmgen_generate_heightfields((const int32_t*)dev_pos, n, dev_hf, dev_bw, stream), "Chunk::generateHeightfield() failed");
    MM_CALL(hipMemcpyAsync(host_hf, dev_hf, (size_t)n * 256 * sizeof(float), hipMemcpyDeviceToHost, stream), "D2H heightfields");
    MM_CALL(hipMemcpyAsync(host_bw, dev_bw, (size_t)n * devBiomeWeightsSize * sizeof(float), hipMemcpyDeviceToHost, stream), "D2H biome weights");
    MM_CALL(hipStreamSynchronize(stream), "Chunk::generateHeightfield() failed");
    for (int i = 0; i < n; ++i) {
        std::memcpy(chunks[i]->heightfield.data(), host_hf + 256 * (size_t)i, 256 * sizeof(float));
        std::memcpy(chunks[i]->biomeWeights.data(), host_bw + (size_t)devBiomeWeightsSize * i, devBiomeWeightsSize * sizeof(float));
    }
}

void Chunk::otherChunkGatherHeightfield(Chunk* c, Chunk* const (&grid)[5][5], int cx, int cz)
{
    c->gatheredHeightfield.assign(18 * 18, 0.f);
    for (int gz = 0; gz < 18; ++gz) {
        for (int gx = 0; gx < 18; ++gx) {
            // source chunk and local column of gathered cell (gx, gz): -1 / 0 / +1 chunk offset per axis
            const int ox = gx == 0 ? -1 : (gx == 17 ? 1 : 0), oz = gz == 0 ? -1 : (gz == 17 ? 1 : 0);
            const int lx = gx == 0 ? 15 : (gx == 17 ? 0 : gx - 1), lz = gz == 0 ? 15 : (gz == 17 ? 0 : gz - 1);
            c->gatheredHeightfield[gx + 18 * gz] = grid[cz + oz][cx + ox]->heightfield[lx + 16 * lz];
        }
    }
}

void Chunk::gatherHeightfield()
{
    floodFillAndIterateNeighbors<5>(ChunkState::HAS_HEIGHTFIELD, ChunkState::NEEDS_LAYERS, &Chunk::otherChunkGatherHeightfield);
}

// ---------------------------------------------------------------------------------------------------------
// layers (chunk.cu:417-469)
// ---------------------------------------------------------------------------------------------------------
void Chunk::generateLayers(std::vector<Chunk*>& chunks, float* host_hf, float* dev_hf, float* host_bw, float* dev_bw, ivec2* host_pos, ivec2* dev_pos,
                           float* host_layers, float* dev_layers, mmhostStream stream)
{
    const int n = (int)chunks.size();
    for (int i = 0; i < n; ++i) {
        Chunk* c = chunks[i];
        std::memcpy(host_hf + (size_t)i * devHeightfieldSize, c->gatheredHeightfield.data(), devHeightfieldSize * sizeof(float));
        c->gatheredHeightfield.clear();
        std::memcpy(host_bw + (size_t)i * devBiomeWeightsSize, c->biomeWeights.data(), devBiomeWeightsSize * sizeof(float));
        host_pos[i] = ivec2(c->worldBlockPos.x, c->worldBlockPos.z);
    }
    MM_CALL(hipMemcpyAsync(dev_hf, host_hf, (size_t)n * devHeightfieldSize * sizeof(float), hipMemcpyHostToDevice, stream), "H2D gathered heightfields");
    MM_CALL(hipMemcpyAsync(dev_bw, host_bw, (size_t)n * devBiomeWeightsSize * sizeof(float), hipMemcpyHostToDevice, stream), "H2D biome weights");
    MM_CALL(hipMemcpyAsync(dev_pos, host_pos, n * sizeof(ivec2), hipMemcpyHostToDevice, stream), "H2D positions");
    MM_CALL(mmgen_generate_layers(dev_hf, dev_bw, (const int32_t*)dev_pos, n, dev_layers, stream), "Chunk::generateLayers() failed");
    MM_CALL(hipMemcpyAsync(host_layers, dev_layers, (size_t)n * devLayersSize * sizeof(float), hipMemcpyDeviceToHost, stream), "D2H layers");
    MM_CALL(hipStreamSynchronize(stream), "Chunk::generateLayers() failed");
    for (int i = 0; i < n; ++i) {
        std::memcpy(chunks[i]->layers.data(), host_layers + (size_t)i * devLayersSize, devLayersSize * sizeof(float));
        std::memcpy(chunks[i]->rawErodedLayers.data(), chunks[i]->layers.data() + 256 * numStratifiedMaterials, 256 * numErodedMaterials * sizeof(float));
    }
}

// ---------------------------------------------------------------------------------------------------------
// erosion (chunk.cu:603-749)
// ---------------------------------------------------------------------------------------------------------
static void copyLayers(Zone* zone, float* packed, bool toPacked)
{
    const int dim = toPacked ? ZONE_SIZE * 2 : ZONE_SIZE;
    const int planes = toPacked ? numErodedMaterials + 1 : numErodedMaterials;     // + heightfield plane on the way in
    for (int cz = 0; cz < dim; ++cz) {
        for (int cx = 0; cx < dim; ++cx) {
            Chunk* c = toPacked ? zone->gatheredChunks[cx + ZONE_SIZE * 2 * cz] : zone->chunks[cx + ZONE_SIZE * cz].get();
            const int bx = (toPacked ? cx : cx + ZONE_SIZE / 2) * 16, bz = (toPacked ? cz : cz + ZONE_SIZE / 2) * 16;
            for (int p = 0; p < planes; ++p) {
                for (int z = 0; z < 16; ++z) {
                    float* chunkRow = (p == numErodedMaterials) ? c->heightfield.data() + 16 * z
                                      : toPacked    ? c->rawErodedLayers.data() + 256 * p + 16 * z          // padding AND centre start from raw planes
                                                    : c->layers.data() + 256 * (numStratifiedMaterials + p) + 16 * z;
                    float* packedRow = packed + (size_t)EROSION_GRID_NUM_COLS * p + (size_t)EROSION_GRID_SIDE_LENGTH_BLOCKS * (bz + z) + bx;
                    if (toPacked) std::memcpy(packedRow, chunkRow, 16 * sizeof(float));
                    else std::memcpy(chunkRow, packedRow, 16 * sizeof(float));
                }
            }
        }
    }
}

void Chunk::erodeZone(Zone* zone, float* host_gathered, float* dev_gathered, float* dev_acc, mmhostStream stream)
{
    copyLayers(zone, host_gathered, true);
    zone->gatheredChunks.clear();
    const size_t bytes = (size_t)EROSION_GRID_NUM_COLS * (numErodedMaterials + 1) * sizeof(float);
    MM_CALL(hipMemcpyAsync(dev_gathered, host_gathered, bytes, hipMemcpyHostToDevice, stream), "H2D gathered layers");
    MM_CALL(mmgen_erode_zone(dev_gathered, dev_acc, stream), "Chunk::erodeZone() failed");
    MM_CALL(hipMemcpyAsync(host_gathered, dev_gathered, bytes, hipMemcpyDeviceToHost, stream), "D2H gathered layers");
    MM_CALL(hipStreamSynchronize(stream), "Chunk::erodeZone() failed");
    copyLayers(zone, host_gathered, false);
    for (auto& c : zone->chunks) c->fixBackwardStratifiedLayers();
}

void Chunk::fixBackwardStratifiedLayers()
{
    for (int col = 0; col < 256; ++col) {
        const float erodedStart = layers[256 * numStratifiedMaterials + col];
        for (int l = numForwardMaterials; l < numStratifiedMaterials; ++l) layers[256 * l + col] = erodedStart - layers[256 * l + col];
    }
}

// ---------------------------------------------------------------------------------------------------------
// caves (chunk.cu:939-993)
// ---------------------------------------------------------------------------------------------------------
void Chunk::generateCaves(std::vector<Chunk*>& chunks, float* host_hf, float* dev_hf, float* host_bw, float* dev_bw, ivec2* host_pos, ivec2* dev_pos,
                          CaveLayer* host_cl, CaveLayer* dev_cl, mmhostStream stream)
{
    const int n = (int)chunks.size();
    for (int i = 0; i < n; ++i) {
        Chunk* c = chunks[i];
        std::memcpy(host_hf + (size_t)i * 256, c->heightfield.data(), 256 * sizeof(float));
        std::memcpy(host_bw + (size_t)i * devBiomeWeightsSize, c->biomeWeights.data(), devBiomeWeightsSize * sizeof(float));
        host_pos[i] = ivec2(c->worldBlockPos.x, c->worldBlockPos.z);
    }
    MM_CALL(hipMemcpyAsync(dev_hf, host_hf, (size_t)n * 256 * sizeof(float), hipMemcpyHostToDevice, stream), "H2D heightfields");
    MM_CALL(hipMemcpyAsync(dev_bw, host_bw, (size_t)n * devBiomeWeightsSize * sizeof(float), hipMemcpyHostToDevice, stream), "H2D biome weights");
    MM_CALL(hipMemcpyAsync(dev_pos, host_pos, n * sizeof(ivec2), hipMemcpyHostToDevice, stream), "H2D positions");
    MM_CALL(mmgen_generate_caves(dev_hf, dev_bw, (const int32_t*)dev_pos, n, ABI_CL(dev_cl), stream), "Chunk::generateCaves() failed");
    MM_CALL(hipMemcpyAsync(host_cl, dev_cl, (size_t)n * devCaveLayersSize * sizeof(CaveLayer), hipMemcpyDeviceToHost, stream), "D2H cave layers");
    MM_CALL(hipStreamSynchronize(stream), "Chunk::generateCaves() failed");
    for (int i = 0; i < n; ++i) std::memcpy(chunks[i]->caveLayers.data(), host_cl + (size_t)i * devCaveLayersSize, devCaveLayersSize * sizeof(CaveLayer));
}

// ---------------------------------------------------------------------------------------------------------
// feature placements (chunk.cu:1147-1196)
// ---------------------------------------------------------------------------------------------------------
void Chunk::generateFeaturePlacements()
{
    // single-chunk device stage with library-side staging (CPU loop in the reference)
    const size_t oHf = 0, oBw = oHf + 256 * 4, oL = oBw + devBiomeWeightsSize * 4, oCl = oL + devLayersSize * 4,
                 oPos = oCl + devCaveLayersSize * sizeof(CaveLayer), oFp = oPos + 16, oCfp = oFp + MMGEN_FP_CAP * sizeof(FeaturePlacement),
                 oCnt = oCfp + MMGEN_CFP_CAP * sizeof(CaveFeaturePlacement), total = oCnt + 16;
    char* d = (char*)g_single.get(total);
    const int32_t pos[2] = {worldBlockPos.x, worldBlockPos.z};
    MM_CALL(hipMemcpy(d + oHf, heightfield.data(), 256 * 4, hipMemcpyHostToDevice), "H2D");
    MM_CALL(hipMemcpy(d + oBw, biomeWeights.data(), devBiomeWeightsSize * 4, hipMemcpyHostToDevice), "H2D");
    MM_CALL(hipMemcpy(d + oL, layers.data(), devLayersSize * 4, hipMemcpyHostToDevice), "H2D");
    MM_CALL(hipMemcpy(d + oCl, caveLayers.data(), devCaveLayersSize * sizeof(CaveLayer), hipMemcpyHostToDevice), "H2D");
    MM_CALL(hipMemcpy(d + oPos, pos, 8, hipMemcpyHostToDevice), "H2D");
    MM_CALL(mmgen_generate_feature_placements((float*)(d + oHf), (float*)(d + oBw), (float*)(d + oL), ABI_CL(d + oCl), (int32_t*)(d + oPos), 1,
                                              ABI_FP(d + oFp), ABI_CFP(d + oCfp), (int32_t*)(d + oCnt), nullptr),
            "Chunk::generateFeaturePlacements() failed");
    int32_t counts[2];
    MM_CALL(hipMemcpy(counts, d + oCnt, 8, hipMemcpyDeviceToHost), "D2H");
    featurePlacements.resize(counts[0]);
    caveFeaturePlacements.resize(std::min(counts[1], MMGEN_CFP_CAP));
    if (counts[0]) MM_CALL(hipMemcpy(featurePlacements.data(), d + oFp, counts[0] * sizeof(FeaturePlacement), hipMemcpyDeviceToHost), "D2H");
    if (!caveFeaturePlacements.empty())
        MM_CALL(hipMemcpy(caveFeaturePlacements.data(), d + oCfp, caveFeaturePlacements.size() * sizeof(CaveFeaturePlacement), hipMemcpyDeviceToHost), "D2H");
}

static const struct { int x, y; } kGatherOffsets[49] = {      // chunk.cu:1158-1167 — the order is observable (first match wins in fill)
    {0, 0}, {0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, -1}, {-1, 0}, {-1, 1}, {2, 0}, {2, 1}, {2, 2}, {1, 2}, {0, 2}, {-1, 2}, {-2, 2},
    {-2, 1}, {-2, 0}, {-2, -1}, {-2, -2}, {-1, -2}, {0, -2}, {1, -2}, {2, -2}, {2, -1}, {-3, -3}, {-2, -3}, {-1, -3}, {0, -3}, {1, -3}, {2, -3},
    {3, -3}, {3, -2}, {3, -1}, {3, 0}, {3, 1}, {3, 2}, {3, 3}, {2, 3}, {1, 3}, {0, 3}, {-1, 3}, {-2, 3}, {-3, 3}, {-3, 2}, {-3, 1}, {-3, 0},
    {-3, -1}, {-3, -2}};

void Chunk::otherChunkGatherFeaturePlacements(Chunk* c, Chunk* const (&grid)[13][13], int cx, int cz)
{
    c->gatheredFeaturePlacements.clear();
    for (const auto& o : kGatherOffsets) {
        const Chunk* n = grid[cz + o.y][cx + o.x];
        c->gatheredFeaturePlacements.insert(c->gatheredFeaturePlacements.end(), n->featurePlacements.begin(), n->featurePlacements.end());
        c->gatheredCaveFeaturePlacements.insert(c->gatheredCaveFeaturePlacements.end(), n->caveFeaturePlacements.begin(), n->caveFeaturePlacements.end());
    }
}

void Chunk::gatherFeaturePlacements()
{
    floodFillAndIterateNeighbors<13>(ChunkState::NEEDS_GATHER_FEATURE_PLACEMENTS, ChunkState::READY_TO_FILL, &Chunk::otherChunkGatherFeaturePlacements);
}

// ---------------------------------------------------------------------------------------------------------
// fill (chunk.cu:1518-1632) — ONE device launch for the whole batch (the reference launches kernFill per chunk), decorators on
// the device before the D2H (the reference runs placeDecorators on the host after it; same blocks).
// ---------------------------------------------------------------------------------------------------------
static const int kFeatureBounds[MMGEN_NUM_FEATURES][2] = {{0, 0}, {-6, 6}, {-3, 12}, {0, 20}, {0, 110}, {0, 15}, {-5, 75}, {-3, 50}, {0, 30}, {0, 15},
                                                          {0, 8}, {0, 10}, {0, 38}, {0, 17}, {0, 5}, {0, 6}, {0, 120}, {-3, 32}, {-6, 64}, {0, 28}, {0, 15}};
static const int kCaveFeatureBounds[MMGEN_NUM_CAVE_FEATURES][2] = {{0, 0}, {-3, 3}, {-3, 3}, {0, 0}, {0, 6}, {-12, 12}, {-12, 12}, {-8, 8}, {-2, 3}, {-2, 5}};

void Chunk::fill(std::vector<Chunk*>& chunks, float* host_hf, float* dev_hf, float* host_bw, float* dev_bw, float* host_layers, float* dev_layers,
                 CaveLayer* host_cl, CaveLayer* dev_cl, FeaturePlacement* dev_fp, CaveFeaturePlacement* dev_cfp, Block* host_blocks, Block* dev_blocks,
                 mmhostStream stream)
{
    const int n = (int)chunks.size();
    std::vector<int32_t> pos(2 * (size_t)n), bounds(4 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        Chunk* c = chunks[i];
        std::memcpy(host_hf + (size_t)i * 256, c->heightfield.data(), 256 * sizeof(float));
        std::memcpy(host_bw + (size_t)i * devBiomeWeightsSize, c->biomeWeights.data(), devBiomeWeightsSize * sizeof(float));
        std::memcpy(host_layers + (size_t)i * devLayersSize, c->layers.data(), devLayersSize * sizeof(float));
        std::memcpy(host_cl + (size_t)i * devCaveLayersSize, c->caveLayers.data(), devCaveLayersSize * sizeof(CaveLayer));
        pos[2 * i] = c->worldBlockPos.x; pos[2 * i + 1] = c->worldBlockPos.z;
    }
    MM_CALL(hipMemcpyAsync(dev_hf, host_hf, (size_t)n * 256 * sizeof(float), hipMemcpyHostToDevice, stream), "H2D heightfields");
    MM_CALL(hipMemcpyAsync(dev_bw, host_bw, (size_t)n * devBiomeWeightsSize * sizeof(float), hipMemcpyHostToDevice, stream), "H2D biome weights");
    MM_CALL(hipMemcpyAsync(dev_layers, host_layers, (size_t)n * devLayersSize * sizeof(float), hipMemcpyHostToDevice, stream), "H2D layers");
    MM_CALL(hipMemcpyAsync(dev_cl, host_cl, (size_t)n * devCaveLayersSize * sizeof(CaveLayer), hipMemcpyHostToDevice, stream), "H2D cave layers");

    for (int i = 0; i < n; ++i) {
        Chunk* c = chunks[i];
        int lo0 = 384, hi0 = -1, lo1 = 384, hi1 = -1;     // unions over the un-truncated lists (chunk.cu:1555-1570)
        for (const auto& e : c->gatheredFeaturePlacements) {
            const mmgen_feature_placement& p = abi(e);
            lo0 = std::min(lo0, p.pos[1] + kFeatureBounds[p.feature][0]); hi0 = std::max(hi0, p.pos[1] + kFeatureBounds[p.feature][1]);
        }
        for (const auto& e : c->gatheredCaveFeaturePlacements) {
            const mmgen_cave_feature_placement& p = abi(e);
            lo1 = std::min(lo1, p.pos[1] + kCaveFeatureBounds[p.feature][0]);
            hi1 = std::max(hi1, p.pos[1] + p.layer_height + kCaveFeatureBounds[p.feature][1]);
        }
        bounds[4 * i] = lo0; bounds[4 * i + 1] = hi0; bounds[4 * i + 2] = lo1; bounds[4 * i + 3] = hi1;

        int nf = std::min((int)c->gatheredFeaturePlacements.size(), MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK);
        if (nf < MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK) { c->gatheredFeaturePlacements.push_back(FeaturePlacement{}); ++nf; }     // NONE sentinel
        MM_CALL(hipMemcpyAsync(dev_fp + (size_t)i * MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK, c->gatheredFeaturePlacements.data(),
                               nf * sizeof(FeaturePlacement), hipMemcpyHostToDevice, stream), "H2D feature placements");
        int nc = std::min((int)c->gatheredCaveFeaturePlacements.size(), MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK);
        if (nc < MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK) { c->gatheredCaveFeaturePlacements.push_back(CaveFeaturePlacement{}); ++nc; }
        MM_CALL(hipMemcpyAsync(dev_cfp + (size_t)i * MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK, c->gatheredCaveFeaturePlacements.data(),
                               nc * sizeof(CaveFeaturePlacement), hipMemcpyHostToDevice, stream), "H2D cave feature placements");
    }
    MM_CALL(hipStreamSynchronize(stream), "Chunk::fill() failed");      // pageable list sources must outlive the copies
    for (Chunk* c : chunks) { c->gatheredFeaturePlacements.clear(); c->gatheredCaveFeaturePlacements.clear(); }

    char* d = (char*)g_bounds.get((size_t)n * 24);
    int32_t* dev_bounds = (int32_t*)d;
    int32_t* dev_pos = (int32_t*)(d + (size_t)n * 16);
    MM_CALL(hipMemcpyAsync(dev_bounds, bounds.data(), (size_t)n * 16, hipMemcpyHostToDevice, stream), "H2D bounds");
    MM_CALL(hipMemcpyAsync(dev_pos, pos.data(), (size_t)n * 8, hipMemcpyHostToDevice, stream), "H2D positions");
    MM_CALL(mmgen_fill(dev_hf, dev_bw, dev_layers, ABI_CL(dev_cl), dev_pos, n, ABI_FP(dev_fp), ABI_CFP(dev_cfp), dev_bounds, (uint8_t*)dev_blocks, stream), "Chunk::fill() failed");
    MM_CALL(mmgen_place_decorators((uint8_t*)dev_blocks, dev_hf, dev_bw, ABI_CL(dev_cl), dev_pos, n, stream), "Chunk::fill() failed");
    MM_CALL(hipMemcpyAsync(host_blocks, dev_blocks, (size_t)n * devBlocksSize, hipMemcpyDeviceToHost, stream), "D2H blocks");
    MM_CALL(hipStreamSynchronize(stream), "Chunk::fill() failed");
    for (int i = 0; i < n; ++i) std::memcpy(chunks[i]->blocks.data(), host_blocks + (size_t)i * devBlocksSize, devBlocksSize);
}

void Chunk::placeDecorators()
{
    const size_t oB = 0, oHf = oB + devBlocksSize, oBw = oHf + 256 * 4, oCl = oBw + devBiomeWeightsSize * 4,
                 oPos = oCl + devCaveLayersSize * sizeof(CaveLayer), total = oPos + 16;
    char* d = (char*)g_single.get(total);
    const int32_t pos[2] = {worldBlockPos.x, worldBlockPos.z};
    MM_CALL(hipMemcpy(d + oB, blocks.data(), devBlocksSize, hipMemcpyHostToDevice), "H2D");
    MM_CALL(hipMemcpy(d + oHf, heightfield.data(), 256 * 4, hipMemcpyHostToDevice), "H2D");
    MM_CALL(hipMemcpy(d + oBw, biomeWeights.data(), devBiomeWeightsSize * 4, hipMemcpyHostToDevice), "H2D");
    MM_CALL(hipMemcpy(d + oCl, caveLayers.data(), devCaveLayersSize * sizeof(CaveLayer), hipMemcpyHostToDevice), "H2D");
    MM_CALL(hipMemcpy(d + oPos, pos, 8, hipMemcpyHostToDevice), "H2D");
    MM_CALL(mmgen_place_decorators((uint8_t*)(d + oB), (float*)(d + oHf), (float*)(d + oBw), ABI_CL(d + oCl), (int32_t*)(d + oPos), 1, nullptr),
            "Chunk::placeDecorators() failed");
    MM_CALL(hipMemcpy(blocks.data(), d + oB, devBlocksSize, hipMemcpyDeviceToHost), "D2H");
}

// chunk.cu:1778-2003.  The reference walks the voxels on the host; here the chunk and its (up to) four neighbours' blocks go to the
// device, the mesher counts, the buffers are sized exactly, and the vertices / indices come back in the reference's order.
void Chunk::createVBOs()
{
    idx.clear();
    verts.clear();
    idxCount = 0;
    const size_t oB = 0, oN = oB + (size_t)5 * devBlocksSize, oPos = oN + 16, oCol = oPos + 8, oCnt = oCol + 256 * 4, oOff = oCnt + 8, total = oOff + 8;
    char* d = (char*)g_single.get(total);
    int32_t nidx[4] = {-1, -1, -1, -1};
    MM_CALL(hipMemcpy(d + oB, blocks.data(), devBlocksSize, hipMemcpyHostToDevice), "H2D");
    int used = 1;
    for (int k = 0; k < 4; ++k) {                 // neighbors: N (+z), E (+x), S (-z), W (-x), the order createVBOs indexes them in
        if (!neighbors[k]) continue;
        MM_CALL(hipMemcpy(d + oB + (size_t)used * devBlocksSize, neighbors[k]->blocks.data(), devBlocksSize, hipMemcpyHostToDevice), "H2D");
        nidx[k] = used++;
    }
    const int32_t pos[2] = {worldBlockPos.x, worldBlockPos.z};
    const uint64_t zero = 0;
    MM_CALL(hipMemcpy(d + oN, nidx, 16, hipMemcpyHostToDevice), "H2D");
    MM_CALL(hipMemcpy(d + oPos, pos, 8, hipMemcpyHostToDevice), "H2D");
    MM_CALL(hipMemcpy(d + oOff, &zero, 8, hipMemcpyHostToDevice), "H2D");
    MM_CALL(mmgen_mesh_count((uint8_t*)(d + oB), nullptr, (int32_t*)(d + oN), 1, (uint32_t*)(d + oCol), (uint32_t*)(d + oCnt), nullptr), "Chunk::createVBOs() count failed");
    uint32_t nv = 0;
    MM_CALL(hipMemcpy(&nv, d + oCnt, 4, hipMemcpyDeviceToHost), "D2H");
    if (nv == 0) return;
    static Scratch out;
    const size_t vb = (size_t)nv * sizeof(Vertex), ib = (size_t)nv / 4 * 6 * sizeof(unsigned int);
    char* o = (char*)out.get(vb + ib);
    MM_CALL(mmgen_mesh_fill((uint8_t*)(d + oB), nullptr, (int32_t*)(d + oN), (int32_t*)(d + oPos), 1, (uint32_t*)(d + oCol), (uint64_t*)(d + oOff), (mmgen_vertex*)o,
                            (uint32_t*)(o + vb), nullptr),
            "Chunk::createVBOs() fill failed");
    verts.resize(nv);
    idx.resize((size_t)nv / 4 * 6);
    MM_CALL(hipMemcpy(verts.data(), o, vb, hipMemcpyDeviceToHost), "D2H");
    MM_CALL(hipMemcpy(idx.data(), o + vb, ib, hipMemcpyDeviceToHost), "D2H");
    idxCount = (int)idx.size();
}

MMHOST_NS_END
