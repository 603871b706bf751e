// tests/refdrop: a headless main for the REFERENCE's own Terrain (its unmodified terrain.cpp + terrain.hpp, compiled where they lie)
// driving mmgen's Chunk (mega-minecraft_amd/host/chunk.{hpp,cpp} built with -DMMHOST_REFERENCE_TREE in place of the reference's
// chunk.hpp / chunk.cu).  Plays src/main.cpp:80-99 (init) and :661-716 (tick at a fixed dt) until the drawable set is complete, then
// writes one line per drawable chunk (mega-minecraft_amd/host/chunk_digest.hpp): position, digest of the blocks - the value
// tests/golden/world_digests.npz holds for the ORACLE's chunks - vertex count, digests of the vertex and index bytes.
//   ref_terrain_dropin <out.txt> [playerChunkX playerChunkZ]
#include "terrain/terrain.hpp"
#include "rendering/optixRenderer.hpp"
#include "../../mega-minecraft_amd/host/chunk_digest.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: ref_terrain_dropin <out.txt> [playerChunkX playerChunkZ]\n"); return 2; }
    const ivec2 player(argc > 3 ? std::atoi(argv[2]) : 0, argc > 3 ? std::atoi(argv[3]) : 0);
    hipSetDevice(GPU_DEVICE);                                 // src/main.cpp:82 (GPU_DEVICE from the reference's defines.hpp)
    BiomeUtils::init();
    OptixRenderer renderer;
    Terrain terrain;
    terrain.setOptixRenderer(&renderer);
    terrain.init();
    terrain.setCurrentChunkPos(player);
    const size_t want = 33 * 33;                              // chunkVbosGenRadius = 16 (terrain.cpp:64)
    const auto t0 = std::chrono::steady_clock::now();
    int ticks = 0;
    while (terrain.getDrawableChunks().size() < want && ticks < 200000) { terrain.tick(1.f / 60.f); ++ticks; }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const auto drawable = terrain.getDrawableChunks();
    std::printf("ref_terrain_dropin: the reference's Terrain::tick x %d (%.1f s of frames at 60 fps, %.2f s wall): %zu drawable chunks around (%d,%d), %d accel builds\n",
                ticks, ticks / 60.0, secs, drawable.size(), player.x, player.y, renderer.built);
    FILE* f = std::fopen(argv[1], "w");
    if (!f) return 2;
    for (Chunk* c : drawable) mmhostWriteChunkDigest(f, c);
    std::fclose(f);
    return drawable.size() == want ? 0 : 1;
}
