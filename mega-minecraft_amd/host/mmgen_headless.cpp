// Headless driver for the C++ `Chunk` mirror: plays the part of Terrain::tick (src/terrain/terrain.cpp:587-960) for ONE erosion zone —
// same stage order, same caller-owned staging buffers as Terrain::initCuda (terrain.cpp:154-185), state set before each stage —
// and checks the blocks it obtains through the per-stage drop-in API against the device-resident region fast path.
//
//   mmgen_headless [zoneChunkX zoneChunkZ]      exit code 0 = the two paths agree bit for bit
#include "chunk.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

using namespace mmhost;

template <class T> static T* pinned(size_t n) { void* p = nullptr; HipUtils::checkError("hipHostMalloc", (int)hipHostMalloc(&p, n * sizeof(T))); return (T*)p; }
template <class T> static T* device(size_t n) { void* p = nullptr; HipUtils::checkError("hipMalloc", (int)hipMalloc(&p, n * sizeof(T))); return (T*)p; }

int main(int argc, char** argv)
{
    const int zx = argc > 2 ? std::atoi(argv[1]) : 0, zz = argc > 2 ? std::atoi(argv[2]) : 0;     // zone origin in chunks (multiple of 12)
    HipUtils::checkError("hipSetDevice", (int)hipSetDevice(0));
    BiomeUtils::init();

    // ---- staging, sized like Terrain::initCuda but for whole-zone batches
    const int maxBatch = 26 * 26;
    ivec2* h_pos = pinned<ivec2>(maxBatch); ivec2* d_pos = device<ivec2>(maxBatch);
    float* h_hf = pinned<float>((size_t)maxBatch * devHeightfieldSize); float* d_hf = device<float>((size_t)maxBatch * devHeightfieldSize);
    float* h_bw = pinned<float>((size_t)maxBatch * devBiomeWeightsSize); float* d_bw = device<float>((size_t)maxBatch * devBiomeWeightsSize);
    float* h_layers = pinned<float>((size_t)576 * devLayersSize); float* d_layers = device<float>((size_t)576 * devLayersSize);
    CaveLayer* h_cl = pinned<CaveLayer>((size_t)144 * devCaveLayersSize); CaveLayer* d_cl = device<CaveLayer>((size_t)144 * devCaveLayersSize);
    float* h_gathered = pinned<float>(devGatheredLayersSize); float* d_gathered = device<float>(devGatheredLayersSize);
    float* d_acc = device<float>(devAccumulatedHeightsSize);
    const int fillBatch = 36;
    Block* h_blocks = pinned<Block>((size_t)fillBatch * devBlocksSize); Block* d_blocks = device<Block>((size_t)fillBatch * devBlocksSize);
    FeaturePlacement* d_fp = device<FeaturePlacement>((size_t)fillBatch * devFeaturePlacementsSize);
    CaveFeaturePlacement* d_cfp = device<CaveFeaturePlacement>((size_t)fillBatch * devCaveFeaturePlacementsSize);
    hipStream_t stream;
    HipUtils::checkError("hipStreamCreate", (int)hipStreamCreate(&stream));

    // ---- world: the zone, its 6-chunk erosion padding, and one more ring so that every padded chunk has its 8 neighbours
    Zone zone({zx, zz});
    std::map<std::pair<int, int>, Chunk*> world;
    std::vector<std::unique_ptr<Chunk>> others;
    for (int cz = zz - 7; cz < zz + 19; ++cz) {
        for (int cx = zx - 7; cx < zx + 19; ++cx) {
            auto c = std::make_unique<Chunk>(ivec2{cx, cz});
            world[{cx, cz}] = c.get();
            const int lx = cx - zx, lz = cz - zz;
            if (lx >= 0 && lx < ZONE_SIZE && lz >= 0 && lz < ZONE_SIZE) { c->zonePtr = &zone; zone.chunks[lx + ZONE_SIZE * lz] = std::move(c); }
            else others.push_back(std::move(c));
        }
    }
    const int ndx[4] = {0, 1, 0, -1}, ndz[4] = {1, 0, -1, 0};       // N, E, S, W (util/enums.hpp:8-16)
    for (auto& kv : world)
        for (int d = 0; d < 4; ++d) {
            auto it = world.find({kv.first.first + ndx[d], kv.first.second + ndz[d]});
            kv.second->neighbors[d] = it == world.end() ? nullptr : it->second;
        }

    // ---- heightfields for all 26x26, gather + layers for the 24x24 gathered area
    std::vector<Chunk*> all, area;
    for (auto& kv : world) all.push_back(kv.second);
    for (Chunk* c : all) c->setState(ChunkState::HAS_HEIGHTFIELD);
    Chunk::generateHeightfields(all, h_pos, d_pos, h_hf, d_hf, h_bw, d_bw, stream);
    for (Chunk* c : all) c->gatherHeightfield();
    for (int cz = zz - 6; cz < zz + 18; ++cz) for (int cx = zx - 6; cx < zx + 18; ++cx) area.push_back(world[{cx, cz}]);
    for (Chunk* c : area) if (c->getState() != ChunkState::NEEDS_LAYERS) { std::fprintf(stderr, "gatherHeightfield did not reach a padded chunk\n"); return 2; }
    for (Chunk* c : area) c->setState(ChunkState::HAS_LAYERS);
    Chunk::generateLayers(area, h_hf, d_hf, h_bw, d_bw, h_pos, d_pos, h_layers, d_layers, stream);

    // ---- erosion of the zone (isZoneReadyForErosion fills gatheredChunks, terrain.cpp:471-522)
    zone.gatheredChunks = area;
    Chunk::erodeZone(&zone, h_gathered, d_gathered, d_acc, stream);
    std::vector<Chunk*> own;
    for (auto& c : zone.chunks) { c->setState(ChunkState::NEEDS_FEATURE_PLACEMENTS); own.push_back(c.get()); }

    // ---- caves, placements, gather, fill for the 6x6 interior whose 7x7 neighbourhoods lie inside the zone
    Chunk::generateCaves(own, h_hf, d_hf, h_bw, d_bw, h_pos, d_pos, h_cl, d_cl, stream);
    for (Chunk* c : own) { c->generateFeaturePlacements(); c->setState(ChunkState::NEEDS_GATHER_FEATURE_PLACEMENTS); }
    for (Chunk* c : own) c->gatherFeaturePlacements();
    std::vector<Chunk*> interior;
    for (int lz = 3; lz < 9; ++lz) for (int lx = 3; lx < 9; ++lx) interior.push_back(zone.chunks[lx + ZONE_SIZE * lz].get());
    for (Chunk* c : interior) {
        if (c->getState() != ChunkState::READY_TO_FILL) { std::fprintf(stderr, "gatherFeaturePlacements did not reach an interior chunk\n"); return 2; }
        c->setState(ChunkState::FILLED);
    }
    Chunk::fill(interior, h_hf, d_hf, h_bw, d_bw, h_layers, d_layers, h_cl, d_cl, d_fp, d_cfp, h_blocks, d_blocks, stream);

    // ---- the same 6x6 chunks through the region fast path
    mmgen_region* region = nullptr;
    HipUtils::checkError("mmgen_region_create", mmgen_region_create(&region));
    Block* d_ref = device<Block>((size_t)36 * devBlocksSize);
    HipUtils::checkError("mmgen_region_generate", mmgen_region_generate(region, zx + 3, zz + 3, 6, 6, MMGEN_REGION_EROSION | MMGEN_REGION_FEATURES | MMGEN_REGION_DECORATORS,
                                                                         d_ref, nullptr, stream));
    std::vector<Block> ref((size_t)36 * devBlocksSize);
    HipUtils::checkError("D2H", (int)hipMemcpy(ref.data(), d_ref, ref.size(), hipMemcpyDeviceToHost));
    mmgen_region_destroy(region);

    size_t diff = 0, placements = 0;
    for (int i = 0; i < 36; ++i) {
        diff += std::memcmp(interior[i]->blocks.data(), ref.data() + (size_t)i * devBlocksSize, devBlocksSize) != 0;
        placements += interior[i]->getFeaturePlacements().size() + interior[i]->getCaveFeaturePlacements().size();
    }
    std::printf("mmgen_headless: zone (%d,%d): 36 chunks through the Chunk API, %zu placements in them, %zu chunks differ from the region path\n",
                zx, zz, placements, diff);
    return diff == 0 ? 0 : 1;
}
