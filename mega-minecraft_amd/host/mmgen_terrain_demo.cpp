// Headless replacement of the reference's main loop for the terrain path (src/main.cpp:80-99 init, :661-716 tick): constructs
// `Terrain`, ticks it at a fixed dt until every generation queue has drained (the reference's DEBUG_TIME_CHUNK_FILL measurement,
// terrain.cpp:939-959), then checks chunks produced by the streaming scheduler against the device-resident region path.
//
//   mmgen_terrain_demo [playerChunkX playerChunkZ [digests.txt]]      exit code 0 = all sampled chunks identical
// digests.txt: one line per drawable chunk (chunk_digest.hpp) - the tests hold them to the CPU oracle's chunks
#include "terrain.hpp"
#include "chunk_digest.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace mmhost;

int main(int argc, char** argv)
{
    const ivec2 player = {argc > 2 ? std::atoi(argv[1]) : 0, argc > 2 ? std::atoi(argv[2]) : 0};
    HipUtils::checkError("hipSetDevice", (int)hipSetDevice(0));
    BiomeUtils::init();
    Terrain terrain;
    terrain.init();
    terrain.setCurrentChunkPos(player);

    const auto t0 = std::chrono::steady_clock::now();
    int ticks = 0, idle = 0;
    while (idle < 3 && ticks < 100000) {
        terrain.tick(1.f / 60.f);
        ++ticks;
        idle = terrain.allQueuesEmpty() ? idle + 1 : 0;
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const size_t drawable = terrain.getDrawableChunks().size();
    std::printf("mmgen_terrain_demo: player chunk (%d,%d): %d ticks, %.2f s wall, %zu chunks created, %zu drawable (max %d)\n", player.x, player.y, ticks,
                secs, terrain.numChunks(), drawable, Terrain::getMaxNumDrawableChunks());

    // sample drawable chunks (centre, corners of the drawable square, a few interior ones) against the region fast path
    mmgen_region* region = nullptr;
    HipUtils::checkError("mmgen_region_create", mmgen_region_create(&region));
    uint8_t* d_ref = nullptr;
    HipUtils::checkError("hipMalloc", (int)hipMalloc((void**)&d_ref, devBlocksSize));
    std::vector<uint8_t> ref(devBlocksSize);
    const ivec2 samples[] = {{0, 0}, {16, 16}, {-16, 16}, {16, -16}, {-16, -16}, {5, -9}, {-11, 3}, {12, 12}};
    int bad = 0, checked = 0;
    for (const ivec2& s : samples) {
        const ivec2 c = {player.x + s.x, player.y + s.y};
        Chunk* chunk = terrain.findChunk(c);
        if (!chunk || chunk->getState() != ChunkState::DRAWABLE) { std::printf("  chunk (%d,%d) is not drawable\n", c.x, c.y); ++bad; continue; }
        HipUtils::checkError("mmgen_region_generate", mmgen_region_generate(region, c.x, c.y, 1, 1, MMGEN_REGION_EROSION | MMGEN_REGION_FEATURES | MMGEN_REGION_DECORATORS,
                                                                             d_ref, nullptr, nullptr));
        HipUtils::checkError("D2H", (int)hipMemcpy(ref.data(), d_ref, devBlocksSize, hipMemcpyDeviceToHost));
        const bool same = std::memcmp(ref.data(), chunk->blocks.data(), devBlocksSize) == 0;
        ++checked;
        if (!same) { ++bad; std::printf("  chunk (%d,%d) differs from the region path\n", c.x, c.y); }
        // createVBOs ran for every drawable chunk: a surface chunk has faces, 4 vertices and 6 indices per quad
        if (chunk->verts.empty() || chunk->verts.size() % 4 != 0 || chunk->idx.size() != chunk->verts.size() / 4 * 6 || chunk->idxCount != (int)chunk->idx.size()) {
            ++bad; std::printf("  chunk (%d,%d): bad mesh (%zu vertices, %zu indices)\n", c.x, c.y, chunk->verts.size(), chunk->idx.size());
        }
    }
    mmgen_region_destroy(region);
    if (argc > 3) {
        FILE* f = std::fopen(argv[3], "w");
        if (!f) return 2;
        for (Chunk* c : terrain.getDrawableChunks()) mmhostWriteChunkDigest(f, c);
        std::fclose(f);
    }
    size_t totalVerts = 0;
    for (Chunk* c : terrain.getDrawableChunks()) totalVerts += c->verts.size();
    std::printf("mmgen_terrain_demo: %zu mesh vertices over the drawable chunks (%.0f per chunk)\n", totalVerts, drawable ? (double)totalVerts / drawable : 0.0);
    std::printf("mmgen_terrain_demo: %d sampled chunks checked against the region path, %d bad\n", checked, bad);
    return (bad == 0 && (int)drawable == Terrain::getMaxNumDrawableChunks()) ? 0 : 1;
}
