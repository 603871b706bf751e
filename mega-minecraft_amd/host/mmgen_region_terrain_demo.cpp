// Headless comparison of the two schedulers over the same C ABI:
//   Terrain        the drop-in mirror of the reference's action-time loop (nine per-stage queues through host Chunk objects)
//   RegionTerrain  the region-batched streaming scheduler (one device-resident region call per missing rectangle, pool meshing)
// Both stream the world around a player position, then around a second position 13 / -5 chunks away (new strips only).  Every
// chunk that is drawable for the player must exist in both with identical blocks, vertices and indices.
//
//   mmgen_region_terrain_demo [playerChunkX playerChunkZ [digests.txt]]      exit code 0 = identical
//       digests.txt: one line per chunk RegionTerrain holds drawable after the second leg (chunk_digest.hpp) - the tests hold them to
//       the CPU oracle's chunks
//   mmgen_region_terrain_demo --bench                          the streaming figures as ONE JSON line (bench.py's `streaming` record):
//       initial load of the radius-16 world and a 32-step walk of one chunk per tick through RegionTerrain - device resident, and with
//       the packed blocks + meshes copied into the host Chunk objects - beside the action-time mirror's initial load
#include "terrain.hpp"
#include "region_terrain.hpp"
#include "chunk_digest.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace mmhost;
using Clock = std::chrono::steady_clock;

static double secondsSince(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

template <class T> static int drain(T& terrain, int maxTicks)
{
    int ticks = 0, idle = 0;
    while (idle < 3 && ticks < maxTicks) {
        terrain.tick(1.f / 60.f);
        ++ticks;
        idle = terrain.allQueuesEmpty() ? idle + 1 : 0;
    }
    return ticks;
}

static int compare(Terrain& a, RegionTerrain& b, ivec2 player)
{
    int bad = 0, checked = 0;
    size_t verts = 0;
    const int r = Terrain::chunkVbosGenRadius;
    for (int dz = -r; dz <= r; ++dz)
        for (int dx = -r; dx <= r; ++dx) {
            const ivec2 c = {player.x + dx, player.y + dz};
            Chunk* ca = a.findChunk(c);
            Chunk* cb = b.findChunk(c);
            ++checked;
            if (!ca || !cb || ca->getState() != ChunkState::DRAWABLE || cb->getState() != ChunkState::DRAWABLE) {
                if (++bad <= 5) std::printf("  chunk (%d,%d): not drawable in %s\n", c.x, c.y, (!ca || ca->getState() != ChunkState::DRAWABLE) ? "Terrain" : "RegionTerrain");
                continue;
            }
            const bool blocksSame = std::memcmp(ca->blocks.data(), cb->blocks.data(), devBlocksSize) == 0;
            const bool vertsSame = ca->verts.size() == cb->verts.size() && std::memcmp(ca->verts.data(), cb->verts.data(), ca->verts.size() * sizeof(Vertex)) == 0;
            const bool idxSame = ca->idx.size() == cb->idx.size() && std::memcmp(ca->idx.data(), cb->idx.data(), ca->idx.size() * sizeof(unsigned int)) == 0;
            verts += ca->verts.size();
            if (!blocksSame || !vertsSame || !idxSame) {
                if (++bad <= 5) std::printf("  chunk (%d,%d): blocks %s, verts %s (%zu vs %zu), idx %s\n", c.x, c.y, blocksSame ? "same" : "DIFFER", vertsSame ? "same" : "DIFFER",
                                            ca->verts.size(), cb->verts.size(), idxSame ? "same" : "DIFFER");
            }
        }
    std::printf("  %d drawable chunks compared (blocks, %zu vertices, indices): %d bad\n", checked, verts, bad);
    return bad;
}

// one scheduler configuration through the streaming scenario: everything around `home`, then `steps` ticks with the player one chunk
// further along +x each (a 35 x 1 strip of new drawable chunks + its neighbour ring per tick)
struct StreamFigures { int loadChunks = 0, loadMeshed = 0; double loadMs = 0; int walkChunks = 0, walkMeshed = 0, ringComputed = 0, ringReused = 0; double walkMs = 0; size_t d2hBytes = 0;
                       long long zoneHits = 0, zoneMisses = 0; };
static StreamFigures stream_scenario(bool copyToHost, ivec2 home, int steps, int zoneCacheZones = 128)
{
    StreamFigures f;
    RegionTerrain t(4096);
    t.zoneCacheZones = zoneCacheZones;
    t.copyToHost = copyToHost;
    t.packedTransfer = true;
    t.dropRadius = 24;
    t.init();
    t.setCurrentChunkPos(home);
    auto t0 = Clock::now();
    do { t.tick(1.f / 60.f); f.loadChunks += t.lastGenerated; f.loadMeshed += t.lastMeshed; f.d2hBytes += t.lastBlockBytesD2H; } while (!t.allQueuesEmpty());
    { const int booked = t.lastMeshed; t.finish(); f.loadMeshed += t.lastMeshed - booked; }      // (a device-resident strip's mesh is booked by the next call)
    HipUtils::checkError("hipDeviceSynchronize", (int)hipDeviceSynchronize());
    f.loadMs = 1e3 * secondsSince(t0);
    t0 = Clock::now();
    for (int k = 1; k <= steps; ++k) {
        t.setCurrentChunkPos({home.x + k, home.y});
        do {
            t.tick(1.f / 60.f);
            f.walkChunks += t.lastGenerated; f.walkMeshed += t.lastMeshed; f.ringComputed += t.lastRingComputed; f.ringReused += t.lastRingReused;
        } while (!t.allQueuesEmpty());
    }
    { const int booked = t.lastMeshed; t.finish(); f.walkMeshed += t.lastMeshed - booked; }
    HipUtils::checkError("hipDeviceSynchronize", (int)hipDeviceSynchronize());
    f.walkMs = 1e3 * secondsSince(t0);
    t.zoneCacheStats(f.zoneHits, f.zoneMisses);
    return f;
}

static int bench_main()
{
    const ivec2 home = {0, 0};
    const int steps = 32;
    stream_scenario(false, {100, 100}, 2);                    // warm-up: allocations, first-launch costs
    const StreamFigures dev = stream_scenario(false, home, steps);
    const StreamFigures devNoZones = stream_scenario(false, home, steps, 0);      // what the zone cache is worth: the same walk relaxing every zone it grazes
    const StreamFigures host = stream_scenario(true, home, steps);
    Terrain stage;
    stage.init();
    stage.setCurrentChunkPos(home);
    const auto t0 = Clock::now();
    const int ticks = drain(stage, 200000);
    const double mirrorS = secondsSince(t0);
    auto rec = [&](const char* name, const StreamFigures& f) {
        std::printf("\"%s\": {\"initial_load\": {\"chunks\": %d, \"meshed\": %d, \"ms\": %.2f, \"chunks_per_s\": %.0f}, "
                    "\"walk\": {\"steps\": %d, \"chunks\": %d, \"meshed\": %d, \"ms_per_step\": %.3f, \"chunks_per_s\": %.0f, \"ring_cells_computed\": %d, "
                    "\"ring_cells_from_cache\": %d}, \"zones_from_cache\": %lld, \"zones_relaxed\": %lld, \"block_bytes_to_host\": %zu}",
                    name, f.loadChunks, f.loadMeshed, f.loadMs, f.loadChunks / (f.loadMs * 1e-3), steps, f.walkChunks, f.walkMeshed, f.walkMs / steps,
                    f.walkChunks / (f.walkMs * 1e-3), f.ringComputed, f.ringReused, f.zoneHits, f.zoneMisses, f.d2hBytes);
    };
    std::printf("{\"scenario\": \"RegionTerrain (host/region_terrain.cpp) around chunk (0,0): everything within radius 16 + the mesh neighbour ring, then %d ticks with the "
                "player one chunk further along +x each; chunks/s = generated chunks incl. their meshing\", ", steps);
    rec("device_resident", dev);
    std::printf(", ");
    rec("device_resident_without_zone_cache", devNoZones);
    std::printf(", ");
    rec("host_chunks_packed_d2h", host);
    std::printf(", \"action_time_mirror\": {\"what\": \"host/terrain.cpp, the reference's scheduler (terrain.cpp:587-960) over the same C ABI, initial load only\", "
                "\"chunks\": %zu, \"ticks\": %d, \"frame_seconds_at_60fps\": %.1f, \"wall_s\": %.3f}}\n", stage.numChunks(), ticks, ticks / 60.0, mirrorS);
    return 0;
}

int main(int argc, char** argv)
{
    ivec2 player = {argc > 2 ? std::atoi(argv[1]) : 0, argc > 2 ? std::atoi(argv[2]) : 0};
    HipUtils::checkError("hipSetDevice", (int)hipSetDevice(0));
    BiomeUtils::init();
    if (argc > 1 && std::strcmp(argv[1], "--bench") == 0) return bench_main();
    Terrain stage;
    stage.init();
    RegionTerrain batched;
    batched.init();
    int bad = 0;
    for (int leg = 0; leg < 2; ++leg) {
        if (leg == 1) player = {player.x + 13, player.y - 5};
        stage.setCurrentChunkPos(player);
        batched.setCurrentChunkPos(player);
        auto t0 = Clock::now();
        const int ticksA = drain(stage, 200000);
        const double sA = secondsSince(t0);
        t0 = Clock::now();
        int generated = 0, regions = 0, meshed = 0, ticksB = 0;
        size_t blockBytes = 0;
        batched.packedTransfer = leg == 0;            // leg 0: blocks cross PCIe in the run-length wire format; leg 1: raw
        while (true) {
            batched.tick(1.f / 60.f);
            ++ticksB;
            generated += batched.lastGenerated; regions += batched.lastRegions; meshed += batched.lastMeshed; blockBytes += batched.lastBlockBytesD2H;
            if (batched.allQueuesEmpty()) break;
        }
        const double sB = secondsSince(t0);
        std::printf("leg %d, player chunk (%d,%d):\n  Terrain        %6d ticks (= %.1f s of frames at 60 fps) %8.3f s wall   %zu chunks exist\n  RegionTerrain  %6d ticks %8.3f s   %d chunks generated in %d regions, %d meshed"
                    "   (%.0fx)\n",
                    leg, player.x, player.y, ticksA, ticksA / 60.0, sA, stage.numChunks(), ticksB, sB, generated, regions, meshed, sB > 0 ? sA / sB : 0.0);
        std::printf("  block data copied to the host: %.1f MB (%s, %.1f KB per chunk)\n", blockBytes / 1e6, batched.packedTransfer ? "wire format" : "raw",
                    generated ? blockBytes / 1e3 / generated : 0.0);
        bad += compare(stage, batched, player);
    }
    if (argc > 3) {
        FILE* f = std::fopen(argv[3], "w");
        if (!f) return 2;
        for (Chunk* c : batched.getDrawableChunks()) mmhostWriteChunkDigest(f, c);
        std::fclose(f);
    }
    {   // the same two legs with everything left on the device (renderer interop): what the GPU path itself costs
        RegionTerrain resident;
        resident.copyToHost = false;
        resident.init();
        ivec2 p = {player.x - 13, player.y + 5};
        for (int leg = 0; leg < 2; ++leg) {
            if (leg == 1) p = player;
            resident.setCurrentChunkPos(p);
            const auto t0 = Clock::now();
            int generated = 0, meshed = 0, reused = 0, computed = 0;
            do {
                resident.tick(1.f / 60.f);
                generated += resident.lastGenerated; meshed += resident.lastMeshed; reused += resident.lastRingReused; computed += resident.lastRingComputed;
            } while (!resident.allQueuesEmpty());
            { const int booked = resident.lastMeshed; resident.finish(); meshed += resident.lastMeshed - booked; }
            const double s = secondsSince(t0);
            std::printf("device-resident leg %d: %d chunks generated, %d meshed in %.1f ms (%.0f generated chunks/s incl. meshing); ring cells: %d computed, %d from the placement cache\n",
                        leg, generated, meshed, 1e3 * s, generated / s, computed, reused);
        }
    }
    {   // several lanes ("multi-GPU streaming"): one lane per GPU of the box, or two handles on the one GPU there is.  The
        // same two legs as above; every drawable chunk must equal the mirror's (blocks, vertices, indices - across the lanes' borders too,
        // where a mesh looks at a neighbour that another lane generated), then the same again device-resident against lane-free digests
        int nDev = 0;
        HipUtils::checkError("hipGetDeviceCount", (int)hipGetDeviceCount(&nDev));
        std::vector<int> devices;
        if (nDev >= 2) for (int d = 0; d < nDev && d < 8; ++d) devices.push_back(d); else devices = {0, 0};
        const ivec2 first = {player.x - 13, player.y + 5};
        RegionTerrain lanes2(4096, devices);
        lanes2.init();
        std::vector<int> share(devices.size(), 0);
        for (int leg = 0; leg < 2; ++leg) {
            lanes2.setCurrentChunkPos(leg == 0 ? first : player);
            const auto t0 = Clock::now();
            int generated = 0;
            do { lanes2.tick(1.f / 60.f); generated += lanes2.lastGenerated; for (size_t l = 0; l < devices.size(); ++l) share[l] += lanes2.lastGeneratedOnLane((int)l); } while (!lanes2.allQueuesEmpty());
            std::printf("%zu lanes (%s), leg %d: %d chunks generated in %.1f ms\n", devices.size(), nDev >= 2 ? "one per GPU" : "two handles on one GPU", leg, generated, 1e3 * secondsSince(t0));
            bad += compare(stage, lanes2, leg == 0 ? first : player);
        }
        std::printf("  chunks per lane:");
        for (size_t l = 0; l < devices.size(); ++l) { std::printf(" %d", share[l]); if (share[l] == 0) ++bad; }
        std::printf("\n");
        if (argc > 3) {
            const std::string path = std::string(argv[3]) + ".lanes";
            FILE* f = std::fopen(path.c_str(), "w");
            if (!f) return 2;
            for (Chunk* c : lanes2.getDrawableChunks()) mmhostWriteChunkDigest(f, c);
            std::fclose(f);
        }
        RegionTerrain residentLanes(4096, devices), residentOne;
        residentLanes.copyToHost = false; residentOne.copyToHost = false;
        residentLanes.init(); residentOne.init();
        residentLanes.setCurrentChunkPos(player); residentOne.setCurrentChunkPos(player);
        do residentLanes.tick(1.f / 60.f); while (!residentLanes.allQueuesEmpty());
        do residentOne.tick(1.f / 60.f); while (!residentOne.allQueuesEmpty());
        int differ = 0;
        const int r = Terrain::chunkVbosGenRadius;
        for (int dz = -r; dz <= r; ++dz)
            for (int dx = -r; dx <= r; ++dx) {
                const ivec2 c = {player.x + dx, player.y + dz};
                Chunk *a = residentLanes.findChunk(c), *b = residentOne.findChunk(c), *m = stage.findChunk(c);
                if (!a || !b || !m || a->getState() != ChunkState::DRAWABLE || b->getState() != ChunkState::DRAWABLE) { ++differ; continue; }
                const uint64_t da = residentLanes.deviceBlocksDigest(c), db = residentOne.deviceBlocksDigest(c);
                if (da != db || da != mmhostDigest(m->blocks.data(), m->blocks.size()) || a->idxCount != b->idxCount || a->idxCount != m->idxCount) ++differ;
            }
        std::printf("device-resident, %zu lanes vs one lane vs the mirror: %d of %d drawable chunks differ (pool digests, index counts)\n", devices.size(), differ, (2 * r + 1) * (2 * r + 1));
        bad += differ;
        // the streaming form of the same square: loaded eight chunks to the west, then walked here one chunk per tick.  Every tick generates a
        // 35-chunk strip whose mesh is only ENQUEUED by its tick and booked by the next (RegionTerrain::completeMesh); at the end every
        // drawable chunk must be what the one-tick load above made of it
        RegionTerrain walked;
        walked.copyToHost = false;
        walked.init();
        int strips = 0, stripChunks = 0;
        for (int k = -8; k <= 0; ++k) {
            walked.setCurrentChunkPos({player.x + k, player.y});
            do { walked.tick(1.f / 60.f); if (k > -8) { ++strips; stripChunks += walked.lastGenerated; } } while (!walked.allQueuesEmpty());
        }
        walked.finish();
        int differWalk = 0;
        for (int dz = -r; dz <= r; ++dz)
            for (int dx = -r; dx <= r; ++dx) {
                const ivec2 c = {player.x + dx, player.y + dz};
                Chunk *a = walked.findChunk(c), *b = residentOne.findChunk(c);
                if (!a || !b || a->getState() != ChunkState::DRAWABLE) { ++differWalk; continue; }
                if (walked.deviceBlocksDigest(c) != residentOne.deviceBlocksDigest(c) || a->idxCount != b->idxCount) ++differWalk;
            }
        std::printf("device-resident walk (%d strip ticks, %d chunks, meshes booked by the following tick) vs the one-tick load: %d of %d drawable chunks differ\n", strips, stripChunks,
                    differWalk, (2 * r + 1) * (2 * r + 1));
        if (strips < 8 || stripChunks > 8 * 128) ++bad;          // (the walk must have gone through the strip path)
        bad += differWalk;
    }
    {   // chunk lifetime: a pool of 2 600 slots serves a 2 x 16-step walk (each step regenerates a strip, far chunks are destroyed and
        // their slots recycled); back at the first position everything was dropped and regenerated, and must equal the mirror's chunks
        const ivec2 home = {player.x - 13, player.y + 5};
        RegionTerrain walker(2600);
        walker.dropRadius = 20;
        walker.init();
        int generated = 0, dropped = 0;
        size_t peak = 0;
        const auto t0 = Clock::now();
        for (int step = 0; step <= 32; ++step) {
            const int k = step <= 16 ? step : 32 - step;
            walker.setCurrentChunkPos({home.x + 7 * k, home.y + 3 * k});
            do { walker.tick(1.f / 60.f); generated += walker.lastGenerated; dropped += walker.lastDropped; } while (!walker.allQueuesEmpty());
            peak = std::max(peak, walker.poolInUse());
        }
        std::printf("lifetime walk: 33 positions, %d chunks generated, %d destroyed, peak %zu of 2600 pool slots, %.2f s\n", generated, dropped, peak, secondsSince(t0));
        if (dropped == 0 || peak > 2600) ++bad;
        bad += compare(stage, walker, home);
    }
    std::printf("mmgen_region_terrain_demo: %s\n", bad == 0 ? "IDENTICAL" : "MISMATCH");
    return bad == 0 ? 0 : 1;
}
