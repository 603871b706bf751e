// mmgen host side — region-batched streaming scheduler (see region_terrain.hpp).
#include "region_terrain.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "chunk_digest.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>

namespace mmhost {

// MMHOST_TICK_PROFILE=1: where the host spends a tick (printed when the terrain is destroyed): planning, the region calls, the mesher's
// calls, waiting for the device
namespace {
struct TickProfile {
    double plan = 0, gen = 0, mesh = 0, wait = 0, total = 0; long long ticks = 0;
    double cPlan = 0, cGen = 0, cMesh = 0, cWait = 0;      // the current tick's (kept only if it generated a strip: at most 128 chunks)
    const bool on = std::getenv("MMHOST_TICK_PROFILE") != nullptr;
    static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
} g_tp;
}


namespace {
const ivec2 kDir4[4] = {{0, 1}, {1, 0}, {0, -1}, {-1, 0}};      // N (+z), E (+x), S (-z), W (-x): Chunk::neighbors order (util/enums.hpp:40-47)
#define RT_CALL(expr, msg) HipUtils::checkError(msg, (int)(expr), __LINE__)
}  // namespace

RegionTerrain::RegionTerrain(size_t poolChunks, std::vector<int> devices) : poolChunks(poolChunks)
{
    if (devices.empty()) {
        int dev = 0;
        RT_CALL(hipGetDevice(&dev), "hipGetDevice failed");
        devices.push_back(dev);
    }
    lanes.resize(devices.size());
    for (size_t i = 0; i < devices.size(); ++i) lanes[i].device = devices[i];
}

RegionTerrain::ChunkPtr RegionTerrain::newChunk(ivec2 worldChunkPos)
{
    void* mem;
    if (!chunkStore.free.empty()) { mem = chunkStore.free.back(); chunkStore.free.pop_back(); }
    else mem = ::operator new(sizeof(Chunk));
    return ChunkPtr(new (mem) Chunk(worldChunkPos), ChunkRecycler{&chunkStore});
}

void RegionTerrain::use(const Lane& L) { RT_CALL(hipSetDevice(L.device), "hipSetDevice failed"); }

// a blocking copy on the lane's stream (one lane: the null stream, like the hipMemcpy it replaces)
void RegionTerrain::copySync(const Lane& L, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, const char* what)
{
    RT_CALL(hipMemcpyAsync(dst, src, bytes, kind, L.stream), what);
    RT_CALL(hipStreamSynchronize(L.stream), what);
}

// host -> device without a synchronisation: staged in the lane's pinned arena, copied on the lane's stream (stream order does the rest).
// An upload that does not fit what is left of the arena falls back to the blocking copy.
void RegionTerrain::uploadAsync(Lane& L, void* dst, const void* src, size_t bytes, const char* what) { uploadAsyncOn(L, L.stream, dst, src, bytes, what); }
void RegionTerrain::uploadAsyncOn(Lane& L, hipStream_t st, void* dst, const void* src, size_t bytes, const char* what)
{
    const size_t need = (bytes + 63) / 64 * 64;
    if (!L.h_pin || L.pinUsed + need > L.pinLimit) {
        RT_CALL(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st), what);
        RT_CALL(hipStreamSynchronize(st), what);
        return;
    }
    char* stage = L.h_pin + L.pinUsed;
    L.pinUsed += need;
    std::memcpy(stage, src, bytes);
    RT_CALL(hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, st), what);
}

RegionTerrain::~RegionTerrain()
{
    if (g_tp.on && g_tp.ticks) {
        std::fprintf(stderr, "RegionTerrain: %lld ticks, us per tick: total %.1f = planning %.1f + region calls %.1f + mesher (incl. waiting %.1f for the previous strip) %.1f + rest\n",
                     g_tp.ticks, g_tp.total / g_tp.ticks, g_tp.plan / g_tp.ticks, g_tp.gen / g_tp.ticks, g_tp.wait / g_tp.ticks, g_tp.mesh / g_tp.ticks);
        g_tp.plan = g_tp.gen = g_tp.mesh = g_tp.wait = g_tp.total = 0; g_tp.ticks = 0;
    }
    for (Lane& L : lanes) {
        (void)hipSetDevice(L.device);
        if (L.evMesh) { (void)hipEventSynchronize(L.evMesh); (void)hipEventDestroy(L.evMesh); }
        if (L.h_pin) (void)hipHostFree(L.h_pin);
        if (L.h_back) (void)hipHostFree(L.h_back);
        if (L.region) mmgen_region_destroy(L.region);
        if (L.d_pool) (void)hipFree(L.d_pool);
        if (L.d_stage) (void)hipFree(L.d_stage);
        if (L.d_cacheFp) (void)hipFree(L.d_cacheFp);
        if (L.d_cacheCfp) (void)hipFree(L.d_cacheCfp);
        if (L.d_cacheCnt) (void)hipFree(L.d_cacheCnt);
        if (L.d_idxWork) (void)hipFree(L.d_idxWork);
        if (L.d_meshOut) (void)hipFree(L.d_meshOut);
        if (L.d_meshWork) (void)hipFree(L.d_meshWork);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    if (!lanes.empty()) (void)hipSetDevice(lanes[0].device);
}

void RegionTerrain::init()
{
    for (Lane& L : lanes) {
        use(L);
        RT_CALL(mmgen_init(L.device), "mmgen_init failed");
        // one lane keeps the caller's null stream (the single-GPU scheduler as it always ran); several lanes get a stream each, so that two
        // handles on ONE device overlap too
        if (lanes.size() > 1) RT_CALL(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking), "hipStreamCreate failed");
        RT_CALL(mmgen_region_create(&L.region), "mmgen_region_create failed");
        // the eroded layers of the zones the walk has touched stay on the device: a strip of new chunks then costs its own chunks, not a
        // relaxation of every zone it grazes (include/mmgen.h mmgen_region_set_zone_cache)
        if (zoneCacheZones > 0) RT_CALL(mmgen_region_set_zone_cache(L.region, zoneCacheZones), "mmgen_region_set_zone_cache failed");
        RT_CALL(hipMalloc((void**)&L.d_pool, (poolChunks + (lanes.size() > 1 ? kGhostSlots : 0)) * (size_t)devBlocksSize), "hipMalloc (chunk pool) failed");
        L.freeSlots.resize(poolChunks);
        for (size_t i = 0; i < poolChunks; ++i) L.freeSlots[i] = (int)(poolChunks - 1 - i);
        // the cache also holds ring chunks that were computed for a region but never generated themselves: 2 x the pool is ample
        L.cacheCells = 2 * poolChunks;
        RT_CALL(hipMalloc((void**)&L.d_cacheFp, L.cacheCells * MMGEN_FP_CAP * sizeof(FeaturePlacement)), "hipMalloc (placement cache) failed");
        RT_CALL(hipMalloc((void**)&L.d_cacheCfp, L.cacheCells * MMGEN_CFP_CAP * sizeof(CaveFeaturePlacement)), "hipMalloc (placement cache) failed");
        RT_CALL(hipMalloc((void**)&L.d_cacheCnt, L.cacheCells * 2 * sizeof(int32_t)), "hipMalloc (placement cache) failed");
        L.freePlacementSlots.resize(L.cacheCells);
        for (size_t i = 0; i < L.cacheCells; ++i) L.freePlacementSlots[i] = (int)(L.cacheCells - 1 - i);
        L.pinCap = 1 << 20;
        RT_CALL(hipHostMalloc((void**)&L.h_pin, L.pinCap, hipHostMallocDefault), "hipHostMalloc (upload arena) failed");
        RT_CALL(hipHostMalloc((void**)&L.h_back, 8192, hipHostMallocDefault), "hipHostMalloc (mesh read-back) failed");
        RT_CALL(hipEventCreateWithFlags(&L.evMesh, hipEventDisableTiming), "hipEventCreate failed");
    }
    use(lanes[0]);
}

void RegionTerrain::zoneCacheStats(long long& hits, long long& misses) const
{
    hits = misses = 0;
    for (const Lane& L : lanes) { long long h = 0, m = 0; mmgen_region_zone_cache_stats(L.region, &h, &m); hits += h; misses += m; }
}

size_t RegionTerrain::poolInUse() const
{
    size_t n = 0;
    for (const Lane& L : lanes) n += poolChunks - L.freeSlots.size();
    return n;
}

uint64_t RegionTerrain::deviceBlocksDigest(ivec2 c)
{
    auto it = cells.find({c.x, c.y});
    if (it == cells.end()) return 0;
    Lane& L = lanes[it->second.lane];
    use(L);
    std::vector<uint8_t> h(devBlocksSize);
    copySync(L, h.data(), L.d_pool + (size_t)it->second.slot * devBlocksSize, devBlocksSize, hipMemcpyDeviceToHost, "D2H (digest) failed");
    use(lanes[0]);
    return mmhostDigest(h.data(), h.size());
}

void* RegionTerrain::ensure(void*& p, size_t& cap, size_t bytes)
{
    if (bytes > cap) {
        if (p) RT_CALL(hipFree(p), "hipFree failed");
        RT_CALL(hipMalloc(&p, bytes), "hipMalloc failed");
        cap = bytes;
    }
    return p;
}

Chunk* RegionTerrain::findChunk(ivec2 c)
{
    auto it = cells.find({c.x, c.y});
    return it == cells.end() ? nullptr : it->second.chunk.get();
}

std::unordered_set<Chunk*> RegionTerrain::getDrawableChunks() { finish(); return drawable; }

// a strip's enqueued mesh: wait for its counts, book its chunks.  Too small a fill (rare: the buffers are sized from what earlier ticks
// needed per chunk) leaves the chunks unmeshed again with the capacity grown; the next meshReady repeats them.
bool RegionTerrain::completeMesh(Lane& L)
{
    if (!L.pm.active) return true;
    use(L);
    { const double tw = g_tp.on ? TickProfile::now() : 0; RT_CALL(hipEventSynchronize(L.evMesh), "mesh build failed"); if (g_tp.on) g_tp.cWait += TickProfile::now() - tw; }
    L.pm.active = false;
    L.generationOutstanding = false;                   // (the event lies behind everything the lane's stream held when the mesh was enqueued)
    const int n = (int)L.pm.work.size();
    const uint32_t* cnt = (const uint32_t*)L.h_back;
    uint64_t totalVerts = 0;
    std::memcpy(&totalVerts, L.h_back + L.pm.totalAt, 8);
    uint32_t most = 0;
    for (int i = 0; i < n; ++i) most = std::max(most, cnt[i]);
    if (totalVerts > L.pm.capacity) {
        L.meshVertsPerChunkCap = std::max<uint64_t>(2 * L.meshVertsPerChunkCap, (totalVerts + n - 1) / n * 2);
        for (Cell* c : L.pm.work) c->meshed = false;
        return false;
    }
    L.meshVertsPerChunkCap = std::max<uint64_t>(L.meshVertsPerChunkCap, (uint64_t)most + most / 4);
    for (int i = 0; i < n; ++i) {
        Chunk* c = L.pm.work[i]->chunk.get();
        c->idxCount = (int)(cnt[i] / 4 * 6);
        c->setState(ChunkState::DRAWABLE);
        drawable.insert(c);
    }
    lastMeshed += n;
    return true;
}

void RegionTerrain::finish()
{
    for (int round = 0; round < 8; ++round) {
        bool whole = true;
        for (Lane& L : lanes) whole = completeMesh(L) && whole;
        if (whole) break;
        meshReady();                                   // a fill that did not fit: again, with the larger buffers
    }
    for (Lane& L : lanes)
        if (L.generationOutstanding) {
            use(L);
            if (L.stream) RT_CALL(hipStreamSynchronize(L.stream), "region generation failed"); else RT_CALL(hipDeviceSynchronize(), "region generation failed");
            L.generationOutstanding = false;
        }
    use(lanes[0]);
}

// one region call: all stages for the rectangle; blocks go straight into pool slots when the next free slots are one contiguous run
// (always, until something has been dropped), else through a staging buffer and one device copy per run of consecutive slots
void RegionTerrain::generateRect(int lane, int cx0, int cz0, int nx, int nz)
{
    Lane& L = lanes[lane];
    use(L);
    hipStream_t st = L.stream;
    const size_t n = (size_t)nx * nz;
    if (L.freeSlots.size() < n) HipUtils::checkError("RegionTerrain: chunk pool exhausted (raise poolChunks or lower dropRadius)", 2, __LINE__);
    // n consecutive free slots if the pool has such a run anywhere (lowest first; the list is sorted descending, so a run of ascending
    // slots is a stretch of the vector read backwards): the region then writes its blocks straight into the pool.  Only a pool without
    // one is served from the lowest free slots through the staging buffer (one device copy per run of consecutive slots below).
    std::vector<int> slots(n);
    bool contiguous = false;
    {
        const size_t m = L.freeSlots.size();
        size_t run = 1, endAt = m;                     // L.freeSlots[endAt - 1 .. endAt - 1 + n) is the run, lowest slot at the highest index
        for (size_t i = m; i-- > 0;) {
            run = (i + 1 < m && L.freeSlots[i] == L.freeSlots[i + 1] + 1) ? run + 1 : 1;
            if (run == n) { endAt = i + 1; break; }
        }
        if (endAt != m || n == 1) {
            const size_t lo = (n == 1) ? m - 1 : endAt - 1;
            for (size_t i = 0; i < n; ++i) slots[i] = L.freeSlots[lo + n - 1 - i];
            L.freeSlots.erase(L.freeSlots.begin() + lo, L.freeSlots.begin() + lo + n);
            contiguous = true;
        } else {
            for (size_t i = 0; i < n; ++i) { slots[i] = L.freeSlots.back(); L.freeSlots.pop_back(); }
            contiguous = true;
            for (size_t i = 1; i < n && contiguous; ++i) contiguous = slots[i] == slots[i - 1] + 1;
        }
    }
    uint8_t* dst = L.d_pool + (size_t)slots[0] * devBlocksSize;
    if (!contiguous) {
        if (L.stageChunks < n) {
            if (L.d_stage) RT_CALL(hipFree(L.d_stage), "hipFree failed");
            RT_CALL(hipMalloc((void**)&L.d_stage, n * (size_t)devBlocksSize), "hipMalloc (stage) failed");
            L.stageChunks = n;
        }
        dst = L.d_stage;
    }
    const unsigned flags = MMGEN_REGION_EROSION | MMGEN_REGION_FEATURES | MMGEN_REGION_DECORATORS;
    if (!cachePlacements) {
        RT_CALL(mmgen_region_generate(L.region, cx0, cz0, nx, nz, flags, dst, nullptr, st), "mmgen_region_generate failed");
    } else {
        // two-phase region: ring cells whose placement lists are cached are masked out of the compute list (no caves, no placements
        // for them) and their lists are copied in; everything this region computed goes into the cache for the regions to come
        const int gw = nx + 6, gh = nz + 6;
        std::vector<uint8_t> mask((size_t)gw * gh, 1);
        std::vector<int32_t> impSrc, impDst, expSrc, expDst;
        for (int z = 0; z < gh; ++z)
            for (int x = 0; x < gw; ++x) {
                const int cell = x + gw * z;
                const bool inR = x >= 3 && x < nx + 3 && z >= 3 && z < nz + 3;
                auto it = L.placementSlot.find({cx0 - 3 + x, cz0 - 3 + z});
                if (it != L.placementSlot.end() && !inR) { mask[cell] = 0; impSrc.push_back(it->second); impDst.push_back(cell); ++lastRingReused; }
                else if (it == L.placementSlot.end()) {
                    if (L.freePlacementSlots.empty()) continue;              // cache full: this cell is simply recomputed next time
                    const int slot = L.freePlacementSlots.back(); L.freePlacementSlots.pop_back();
                    L.placementSlot[{cx0 - 3 + x, cz0 - 3 + z}] = slot;
                    expSrc.push_back(cell); expDst.push_back(slot);
                    if (!inR) ++lastRingComputed;
                }
            }
        // the output is known before the begin: it then issues the base fill itself, as soon as the caves' extents and the eroded layers
        // exist - beside the cave biomes, the placement pass and the host's work on the placement cache below - instead of at the finish
        RT_CALL(mmgen_region_set_output(L.region, dst), "mmgen_region_set_output failed");
        RT_CALL(mmgen_region_begin(L.region, cx0, cz0, nx, nz, flags, mask.data(), st), "mmgen_region_begin failed");
        FeaturePlacement* gfp; CaveFeaturePlacement* gcfp; int32_t* gcnt; int gx0, gz0, w, h;
        RT_CALL(mmgen_region_placement_buffers(L.region, &gfp, &gcfp, &gcnt, &gx0, &gz0, &w, &h), "mmgen_region_placement_buffers failed");
        const size_t ni = impSrc.size(), ne = expSrc.size();
        int32_t* wk = (int32_t*)ensure(L.d_idxWork, L.idxWorkCap, (2 * ni + 2 * ne + 4) * sizeof(int32_t));
        if (ni + ne) {                                          // the four index lists in one copy
            std::vector<int32_t> lists;
            lists.reserve(2 * (ni + ne));
            lists.insert(lists.end(), impSrc.begin(), impSrc.end()); lists.insert(lists.end(), impDst.begin(), impDst.end());
            lists.insert(lists.end(), expSrc.begin(), expSrc.end()); lists.insert(lists.end(), expDst.begin(), expDst.end());
            uploadAsync(L, wk, lists.data(), lists.size() * 4, "H2D failed");
        }
        if (ni) RT_CALL(mmgen_copy_placements(L.d_cacheFp, L.d_cacheCfp, L.d_cacheCnt, wk, gfp, gcfp, gcnt, wk + ni, (int)ni, st), "mmgen_copy_placements failed");
        if (ne) RT_CALL(mmgen_copy_placements(gfp, gcfp, gcnt, wk + 2 * ni, L.d_cacheFp, L.d_cacheCfp, L.d_cacheCnt, wk + 2 * ni + ne, (int)ne, st), "mmgen_copy_placements failed");
        RT_CALL(mmgen_region_finish(L.region, dst, nullptr, nullptr, nullptr, st), "mmgen_region_finish failed");
    }
    if (!contiguous)
        for (size_t i = 0; i < n;) {
            size_t j = i + 1;
            while (j < n && slots[j] == slots[j - 1] + 1) ++j;
            RT_CALL(hipMemcpyAsync(L.d_pool + (size_t)slots[i] * devBlocksSize, L.d_stage + i * (size_t)devBlocksSize, (j - i) * (size_t)devBlocksSize, hipMemcpyDeviceToDevice, st),
                    "D2D into pool slots failed");
            i = j;
        }
    std::vector<uint64_t> packOff;
    std::vector<uint32_t> packBytes;
    if (copyToHost && packedTransfer) {
        // wire format: count, prefix on the host, fill, one copy of ~10 KB per chunk, decode into the Chunk objects below
        const size_t oSlots = 0, oRuns = oSlots + n * 4, oBytes = oRuns + n * 512, oOff = (oBytes + n * 4 + 7) / 8 * 8, total = oOff + n * 8;
        char* w = (char*)ensure(L.d_meshWork, L.meshWorkCap, total);
        uploadAsync(L, w + oSlots, slots.data(), n * 4, "H2D failed");
        RT_CALL(mmgen_pack_count(L.d_pool, (int32_t*)(w + oSlots), (int)n, (uint16_t*)(w + oRuns), (uint32_t*)(w + oBytes), st), "mmgen_pack_count failed");
        packBytes.resize(n); packOff.resize(n);
        copySync(L, packBytes.data(), w + oBytes, n * 4, hipMemcpyDeviceToHost, "D2H failed");
        uint64_t totalBytes = 0;
        for (size_t i = 0; i < n; ++i) { packOff[i] = totalBytes; totalBytes += packBytes[i]; }
        uploadAsync(L, w + oOff, packOff.data(), n * 8, "H2D failed");
        char* o = (char*)ensure(L.d_meshOut, L.meshOutCap, totalBytes + 64);
        RT_CALL(mmgen_pack_fill(L.d_pool, (int32_t*)(w + oSlots), (int)n, (uint16_t*)(w + oRuns), (uint64_t*)(w + oOff), (uint8_t*)o, st), "mmgen_pack_fill failed");
        L.hostStage.resize(totalBytes);
        copySync(L, L.hostStage.data(), o, totalBytes, hipMemcpyDeviceToHost, "D2H packed blocks failed");
        lastBlockBytesD2H += totalBytes;
    } else if (copyToHost) {
        L.hostStage.resize(n * (size_t)devBlocksSize);
        for (size_t i = 0; i < n;) {
            size_t j = i + 1;
            while (j < n && slots[j] == slots[j - 1] + 1) ++j;
            copySync(L, L.hostStage.data() + i * (size_t)devBlocksSize, L.d_pool + (size_t)slots[i] * devBlocksSize, (j - i) * (size_t)devBlocksSize, hipMemcpyDeviceToHost,
                     "D2H blocks failed");
            i = j;
        }
        lastBlockBytesD2H += L.hostStage.size();
    } else {
        // device resident: nothing on the host reads these blocks.  The mesher's calls come behind the region's on the same (null)
        // stream and end with a synchronisation; a tick without meshing synchronises at its end (tick): errors surface there, and the
        // 33 Chunk objects of a strip are made while the GPU generates their blocks instead of after it
        L.generationOutstanding = true;
    }
    for (int z = 0; z < nz; ++z)
        for (int x = 0; x < nx; ++x) {
            const ivec2 c = {cx0 + x, cz0 + z};
            const size_t i = (size_t)x + (size_t)nx * z;
            Cell cell;
            cell.chunk = newChunk(c);
            cell.slot = slots[i];
            cell.meshed = false;
            cell.lane = lane;
            if (copyToHost && packedTransfer) {
                if (mmgen_unpack_chunk_host(L.hostStage.data() + packOff[i], packBytes[i], cell.chunk->blocks.data()) != 0)
                    HipUtils::checkError("RegionTerrain: malformed packed chunk", 2, __LINE__);
            } else if (copyToHost) {
                std::memcpy(cell.chunk->blocks.data(), L.hostStage.data() + i * devBlocksSize, devBlocksSize);
            }
            cell.chunk->setState(ChunkState::FILLED);
            Chunk* cp = cell.chunk.get();
            cells.emplace(std::make_pair(c.x, c.y), std::move(cell));
            for (int k = 0; k < 4; ++k) {                      // neighbour links, as Terrain::updateChunk keeps them
                auto it = cells.find({c.x + kDir4[k].x, c.y + kDir4[k].y});
                if (it == cells.end()) continue;
                cp->neighbors[k] = it->second.chunk.get();
                it->second.chunk->neighbors[(k + 2) % 4] = cp;
            }
        }
    lastGenerated += (int)n;
    L.lastGenerated += (int)n;
    L.totalGenerated += (long long)n;
    lastRegions += 1;
}

// chunk lifetime: destroy what is far from the player, recycle its pool slot, unlink it from its neighbours
void RegionTerrain::dropFarChunks()
{
    for (Lane& L : lanes) (void)completeMesh(L);       // (a pending mesh points at its cells)
    for (auto it = cells.begin(); it != cells.end();) {
        const int dx = it->first.first - plannedFor.x, dz = it->first.second - plannedFor.y;
        if (std::max(std::abs(dx), std::abs(dz)) <= dropRadius) { ++it; continue; }
        Chunk* c = it->second.chunk.get();
        for (int k = 0; k < 4; ++k)
            if (c->neighbors[k]) c->neighbors[k]->neighbors[(k + 2) % 4] = nullptr;
        drawable.erase(c);
        lanes[it->second.lane].freeSlots.push_back(it->second.slot);
        it = cells.erase(it);
        ++lastDropped;
    }
    for (Lane& L : lanes) {
        if (lastDropped) std::sort(L.freeSlots.begin(), L.freeSlots.end(), [](int a, int b) { return a > b; });
        for (auto it = L.placementSlot.begin(); it != L.placementSlot.end();) {
            const int dx = it->first.first - plannedFor.x, dz = it->first.second - plannedFor.y;
            if (std::max(std::abs(dx), std::abs(dz)) <= dropRadius + 3) { ++it; continue; }
            L.freePlacementSlots.push_back(it->second);
            it = L.placementSlot.erase(it);
        }
    }
}

// every unmeshed chunk of the drawable square whose four neighbours exist: one count + fill pair over each lane's pool
void RegionTerrain::meshReady()
{
    if (lanes.size() > 1) {
        // a lane's meshes look at neighbours that another lane may have generated in this very tick: all generation first
        for (Lane& L : lanes) { use(L); RT_CALL(hipStreamSynchronize(L.stream), "region generation failed"); L.generationOutstanding = false; }
    }
    for (int l = 0; l < (int)lanes.size(); ++l) meshLane(l);
    use(lanes[0]);
}

void RegionTerrain::meshLane(int lane)
{
    Lane& L = lanes[lane];
    (void)completeMesh(L);              // last tick's strip (long over by now: this tick's planning and generation calls lie in between)
    use(L);
    hipStream_t st = L.stream;
    std::vector<Cell*> work;
    std::vector<int32_t> meta;          // per chunk: slot, 4 neighbour slots, world block x, z
    std::map<std::pair<int, int>, int> ghostOf;      // a foreign neighbour's position -> its ghost slot in this lane's pool (this tick)
    const int r = chunkVbosGenRadius;
    for (int dz = -r; dz <= r; ++dz)
        for (int dx = -r; dx <= r; ++dx) {
            auto it = cells.find({plannedFor.x + dx, plannedFor.y + dz});
            if (it == cells.end() || it->second.meshed || it->second.lane != lane) continue;
            int32_t nb[4];
            bool all = true;
            for (int k = 0; k < 4 && all; ++k) {
                const std::pair<int, int> np = {plannedFor.x + dx + kDir4[k].x, plannedFor.y + dz + kDir4[k].y};
                auto n = cells.find(np);
                if (n == cells.end()) { all = false; break; }
                if (n->second.lane == lane) { nb[k] = n->second.slot; continue; }
                // the one-chunk overlap between lanes: the neighbour's blocks, device to device, into a ghost slot behind this lane's pool
                auto g = ghostOf.find(np);
                if (g == ghostOf.end()) {
                    if (ghostOf.size() >= kGhostSlots) { all = false; break; }      // (more foreign neighbours than ghost slots in one tick: next tick)
                    const int gs = (int)(poolChunks + ghostOf.size());
                    const Lane& O = lanes[n->second.lane];
                    RT_CALL(hipMemcpyPeerAsync(L.d_pool + (size_t)gs * devBlocksSize, L.device, O.d_pool + (size_t)n->second.slot * devBlocksSize, O.device,
                                               devBlocksSize, st), "peer copy of a neighbour chunk failed");
                    g = ghostOf.emplace(np, gs).first;
                }
                nb[k] = g->second;
            }
            if (!all) continue;         // terrain.cpp:569-585: VBOs only once all four neighbours are filled
            work.push_back(&it->second);
            meta.push_back(it->second.slot);
            for (int k = 0; k < 4; ++k) meta.push_back(nb[k]);
            meta.push_back(it->second.chunk->worldBlockPos.x);
            meta.push_back(it->second.chunk->worldBlockPos.z);
        }
    const int n = (int)work.size();
    if (n == 0) return;
    // device work area: chunk idx [n], neighbour idx [n][4], positions [n][2], column counts [n][256], chunk counts [n], offsets [n] (u64)
    const size_t oIdx = 0, oNb = oIdx + (size_t)n * 4, oPos = oNb + (size_t)n * 16, oCol = oPos + (size_t)n * 8, oCnt = oCol + (size_t)n * 1024,
                 oOff = (oCnt + (size_t)n * 4 + 7) / 8 * 8, oTot = oOff + (size_t)n * 8, total = oTot + 8;
    char* w = (char*)ensure(L.d_meshWork, L.meshWorkCap, total);
    // the three input arrays are adjacent in the work area: one copy (a blocking copy from pageable memory is 20 - 30 us of a 1.5 ms tick)
    std::vector<int32_t> hIn((size_t)n * 7);
    int32_t *hIdx = hIn.data(), *hNb = hIdx + n, *hPos = hNb + (size_t)n * 4;
    for (int i = 0; i < n; ++i) {
        hIdx[i] = meta[7 * i];
        for (int k = 0; k < 4; ++k) hNb[4 * i + k] = meta[7 * i + 1 + k];
        hPos[2 * i] = meta[7 * i + 5]; hPos[2 * i + 1] = meta[7 * i + 6];
    }
    static_assert(sizeof(int32_t) == 4, "layout of the work area");
    // a device-resident strip: its mesh is enqueued here and booked by the next tick (completeMesh).  (On a stream of its own, beside the
    // next tick's region call, a tick took 2.1 - 2.7 ms instead of 0.8: every hand-over between the null stream and another queue costs
    // tens of microseconds on a chip this empty - profiles/LOG.md round 6.)
    const size_t inNeed = ((size_t)n * 28 + 63) / 64 * 64;
    const bool deferred = !copyToHost && n <= 128 && L.h_back && oTot + 8 - oCnt <= 8192 && L.h_pin && L.pinUsed + inNeed <= L.pinLimit;
    // The strip's kernels read their few inputs (a slot, four neighbour slots and a position per chunk) straight from the pinned arena and write
    // the chunks' counts, offsets and the total straight into pinned memory (hipHostMalloc memory has one address on both sides): a 1 KB copy in
    // each direction is 5 us of transfer and twice that of gaps in a chain of thirty dependent launches, a few PCIe reads inside a kernel are not
    int32_t *kIdx = (int32_t*)(w + oIdx), *kNb = (int32_t*)(w + oNb), *kPos = (int32_t*)(w + oPos);
    uint32_t* kCnt = (uint32_t*)(w + oCnt);
    if (deferred) {
        char* in = L.h_pin + L.pinUsed;                    // (this tick's half of the arena: intact until the next tick has booked this mesh)
        L.pinUsed += inNeed;
        std::memcpy(in, hIn.data(), (size_t)n * 28);
        kIdx = (int32_t*)in; kNb = kIdx + n; kPos = kNb + (size_t)n * 4;
        kCnt = (uint32_t*)L.h_back;
    } else {
        uploadAsync(L, w + oIdx, hIn.data(), (size_t)n * 28, "H2D failed");
    }
    RT_CALL(mmgen_mesh_count(L.d_pool, kIdx, kNb, n, (uint32_t*)(w + oCol), kCnt, st), "mmgen_mesh_count failed");
    std::vector<uint32_t> cnt(n);
    std::vector<uint64_t> off(n);
    uint64_t totalVerts = 0;
    size_t vb = 0, ib = 0;
    char* o = nullptr;
    if (deferred) {
        // device resident, a strip: no host round trip between the count and the fill, and none behind the fill either.  The offsets are
        // summed inside the fill (mmgen_mesh_fill_strip), the output buffers are sized from what earlier ticks needed per chunk, the fill leaves out any chunk
        // that would end beyond them; counts, offsets and the total are written into pinned memory by the kernels themselves, and the tick
        // ends here: completeMesh books the chunks when the next tick (or finish) gets to it
        const uint64_t capacity = (uint64_t)n * L.meshVertsPerChunkCap;
        vb = (size_t)capacity * sizeof(Vertex); ib = (size_t)capacity / 4 * 6 * sizeof(unsigned int);
        o = (char*)ensure(L.d_meshOut, L.meshOutCap, vb + ib + 64);
        RT_CALL(mmgen_mesh_fill_strip(L.d_pool, kIdx, kNb, kPos, n, (uint32_t*)(w + oCol), kCnt, (uint64_t*)(L.h_back + (oOff - oCnt)), (uint64_t*)(L.h_back + (oTot - oCnt)),
                                      capacity, (Vertex*)o, (uint32_t*)(o + vb), st), "mmgen_mesh_fill_strip failed");
        RT_CALL(hipEventRecord(L.evMesh, st), "hipEventRecord failed");
        for (Cell* c : work) c->meshed = true;          // (not picked again; drawable once booked)
        L.pm.active = true; L.pm.work = work; L.pm.capacity = capacity; L.pm.totalAt = oTot - oCnt;
        return;
    } else if (!copyToHost && n <= 128) {
        for (;;) {
            const uint64_t capacity = (uint64_t)n * L.meshVertsPerChunkCap;
            vb = (size_t)capacity * sizeof(Vertex); ib = (size_t)capacity / 4 * 6 * sizeof(unsigned int);
            o = (char*)ensure(L.d_meshOut, L.meshOutCap, vb + ib + 64);
            RT_CALL(mmgen_mesh_offsets((uint32_t*)(w + oCnt), n, (uint64_t*)(w + oOff), (uint64_t*)(w + oTot), st), "mmgen_mesh_offsets failed");
            RT_CALL(mmgen_mesh_fill_capped(L.d_pool, (int32_t*)(w + oIdx), (int32_t*)(w + oNb), (int32_t*)(w + oPos), n, (uint32_t*)(w + oCol), (uint64_t*)(w + oOff),
                                           capacity, (Vertex*)o, (uint32_t*)(o + vb), st), "mmgen_mesh_fill_capped failed");
            // counts, offsets and the total are adjacent in the work area: ONE copy into the pinned arena (two copies to pageable memory were
            // 5 + 5 us of transfers 22 us apart at the very end of the tick)
            const size_t span = oTot + 8 - oCnt, need = (span + 63) / 64 * 64;
            if (L.h_pin && L.pinUsed + need <= L.pinLimit) {
                char* back = L.h_pin + L.pinUsed;                                          // (not kept: the next round of this loop, if any, re-uses it)
                copySync(L, back, w + oCnt, span, hipMemcpyDeviceToHost, "mesh build failed");      // (waits for the whole tick's work on this lane)
                std::memcpy(cnt.data(), back, (size_t)n * 4);
                std::memcpy(&totalVerts, back + (oTot - oCnt), 8);
            } else {
                RT_CALL(hipMemcpyAsync(cnt.data(), w + oCnt, (size_t)n * 4, hipMemcpyDeviceToHost, st), "D2H failed");
                copySync(L, &totalVerts, w + oTot, 8, hipMemcpyDeviceToHost, "mesh build failed");
            }
            uint32_t most = 0;
            for (int i = 0; i < n; ++i) most = std::max(most, cnt[i]);
            if (totalVerts <= capacity) { L.meshVertsPerChunkCap = std::max<uint64_t>(L.meshVertsPerChunkCap, (uint64_t)most + most / 4); break; }
            L.meshVertsPerChunkCap = std::max<uint64_t>(2 * L.meshVertsPerChunkCap, (totalVerts + n - 1) / n * 2);
        }
        uint64_t run = 0;
        for (int i = 0; i < n; ++i) { off[i] = run; run += cnt[i]; }
    } else {
        copySync(L, cnt.data(), w + oCnt, (size_t)n * 4, hipMemcpyDeviceToHost, "D2H failed");
        for (int i = 0; i < n; ++i) { off[i] = totalVerts; totalVerts += cnt[i]; }
        uploadAsync(L, w + oOff, off.data(), (size_t)n * 8, "H2D failed");
        vb = (size_t)totalVerts * sizeof(Vertex); ib = (size_t)totalVerts / 4 * 6 * sizeof(unsigned int);
        o = (char*)ensure(L.d_meshOut, L.meshOutCap, vb + ib + 64);
        if (totalVerts)
            RT_CALL(mmgen_mesh_fill(L.d_pool, (int32_t*)(w + oIdx), (int32_t*)(w + oNb), (int32_t*)(w + oPos), n, (uint32_t*)(w + oCol), (uint64_t*)(w + oOff),
                                    (Vertex*)o, (uint32_t*)(o + vb), st),
                    "mmgen_mesh_fill failed");
        if (copyToHost && totalVerts) {
            L.hostStage.resize(vb + ib);
            copySync(L, L.hostStage.data(), o, vb + ib, hipMemcpyDeviceToHost, "D2H mesh failed");
        } else if (L.stream) {
            RT_CALL(hipStreamSynchronize(L.stream), "mesh build failed");
        } else {
            RT_CALL(hipDeviceSynchronize(), "mesh build failed");
        }
    }
    L.generationOutstanding = false;    // (either branch waited for everything the lane's stream held)
    for (int i = 0; i < n; ++i) {
        Chunk* c = work[i]->chunk.get();
        if (copyToHost) {
            const Vertex* v = (const Vertex*)L.hostStage.data() + off[i];
            const unsigned int* ix = (const unsigned int*)(L.hostStage.data() + vb) + off[i] / 4 * 6;
            c->verts.assign(v, v + cnt[i]);
            c->idx.assign(ix, ix + (size_t)cnt[i] / 4 * 6);
        }
        c->idxCount = (int)(cnt[i] / 4 * 6);
        c->setState(ChunkState::DRAWABLE);
        work[i]->meshed = true;
        drawable.insert(c);
    }
    lastMeshed += n;
}

void RegionTerrain::tick(float)
{
    const double tp0 = g_tp.on ? TickProfile::now() : 0;
    struct Total { double t0; const int* gen; ~Total() { if (g_tp.on && *gen > 0 && *gen <= 128) { g_tp.total += TickProfile::now() - t0; ++g_tp.ticks; g_tp.plan += g_tp.cPlan; g_tp.gen += g_tp.cGen; g_tp.mesh += g_tp.cMesh; g_tp.wait += g_tp.cWait; } g_tp.cPlan = g_tp.cGen = g_tp.cMesh = g_tp.cWait = 0; } } totalGuard{tp0, &lastGenerated};
    lastGenerated = lastMeshed = lastRegions = lastDropped = lastRingReused = lastRingComputed = 0;
    lastBlockBytesD2H = 0;
    ++tickCount;
    for (Lane& L : lanes) {
        L.lastGenerated = 0;
        // (what the tick before the last one staged in this half has been read: that tick either ended with a synchronisation or left a
        // strip's mesh pending, which the last tick completed before it enqueued its own)
        L.pinUsed = (tickCount & 1) ? L.pinCap / 2 : 0;
        L.pinLimit = L.pinUsed + L.pinCap / 2;
    }
    const int R = chunkVbosGenRadius + 1, S = 2 * R + 1;
    if (!planned || !(plannedFor == currentChunkPos)) {
        plannedFor = currentChunkPos; planned = true; pending = true;
        // the far chunks are looked for (a walk over every cell and every cached placement list) when the player has moved four chunks
        // since the last look, not at every step: what is dropped a few steps later is the same set, and a pool that is running low
        // forces the look at once.  (This planning runs before anything of the tick is enqueued: the GPU idles through it.)
        size_t lowest = poolChunks;
        for (const Lane& L : lanes) lowest = std::min(lowest, L.freeSlots.size());
        if (!droppedOnce || std::max(std::abs(plannedFor.x - droppedAt.x), std::abs(plannedFor.y - droppedAt.y)) >= 4 || lowest < (size_t)S * S) {
            dropFarChunks();
            droppedAt = plannedFor; droppedOnce = true;
            // (the last completed square is only known to be whole while no look from further than dropRadius - R away has happened)
            if (std::max(std::abs(plannedFor.x - completeAt.x), std::abs(plannedFor.y - completeAt.y)) + R > dropRadius) completeValid = false;
        }
    }
    if (g_tp.on) g_tp.cPlan += TickProfile::now() - tp0;
    if (!pending) { finish(); return; }                // (nothing to generate: what the last tick left enqueued is booked now)

    // missing cells of the generation square (drawable radius + the ring the border meshes look at).  When the previous plan was
    // completed around a position close by, the cells of ITS square exist (nothing within dropRadius is ever dropped): only the newly
    // exposed strips are looked up
    std::vector<uint8_t> missing((size_t)S * S, 0);
    int numMissing = 0;
    const bool near = completeValid && std::max(std::abs(plannedFor.x - completeAt.x), std::abs(plannedFor.y - completeAt.y)) <= dropRadius - R - 4;
    for (int z = 0; z < S; ++z)
        for (int x = 0; x < S; ++x) {
            const int cx = plannedFor.x - R + x, cz = plannedFor.y - R + z;
            if (near && std::abs(cx - completeAt.x) <= R && std::abs(cz - completeAt.y) <= R) continue;
            if (!cells.count({cx, cz})) { missing[(size_t)x + (size_t)S * z] = 1; ++numMissing; }
        }

    // greedy rectangle cover: maximal x-run of the first missing cell, extended down while the whole run is missing
    int budget = maxChunksPerTick;
    for (int z = 0; z < S && numMissing > 0 && budget > 0; ++z)
        for (int x = 0; x < S && budget > 0; ++x) {
            if (!missing[(size_t)x + (size_t)S * z]) continue;
            int nx = 1;
            while (x + nx < S && missing[(size_t)(x + nx) + (size_t)S * z]) ++nx;
            int nz = 1;
            for (; z + nz < S; ++nz) {
                bool full = true;
                for (int i = 0; i < nx && full; ++i) full = missing[(size_t)(x + i) + (size_t)S * (z + nz)] != 0;
                if (!full) break;
            }
            if (nx * nz > budget) { nz = std::max(1, budget / nx); if (nx * nz > budget) nx = budget; }
            // deal the rectangle to the lanes: a large one is cut into one z-strip per lane (every strip a region call of its own: it builds
            // the placement ring and the erosion padding it needs itself), a small one goes to the lane with the least work this tick
            const int nl = (int)lanes.size();
            const int strips = (nl > 1 && nz >= nl && nx * nz >= 64 * nl) ? nl : 1;
            for (int k = 0; k < strips; ++k) {
                const int z0 = (int)((long long)nz * k / strips), z1 = (int)((long long)nz * (k + 1) / strips);
                int lane = 0;
                for (int l = 1; l < nl; ++l)
                    if (lanes[l].lastGenerated < lanes[lane].lastGenerated || (lanes[l].lastGenerated == lanes[lane].lastGenerated && lanes[l].totalGenerated < lanes[lane].totalGenerated)) lane = l;
                const double tg = g_tp.on ? TickProfile::now() : 0;
                generateRect(lane, plannedFor.x - R + x, plannedFor.y - R + z + z0, nx, z1 - z0);
                if (g_tp.on) g_tp.cGen += TickProfile::now() - tg;
            }
            for (int j = 0; j < nz; ++j) for (int i = 0; i < nx; ++i) missing[(size_t)(x + i) + (size_t)S * (z + j)] = 0;
            numMissing -= nx * nz;
            budget -= nx * nz;
        }
    { const double tm = g_tp.on ? TickProfile::now() : 0; meshReady(); if (g_tp.on) g_tp.cMesh += TickProfile::now() - tm; }
    for (Lane& L : lanes)
        if (L.generationOutstanding && !L.pm.active) {
            use(L);
            if (L.stream) RT_CALL(hipStreamSynchronize(L.stream), "region generation failed"); else RT_CALL(hipDeviceSynchronize(), "region generation failed");
            L.generationOutstanding = false;
        }
    use(lanes[0]);
    pending = numMissing > 0;
    if (!pending) { completeAt = plannedFor; completeValid = true; }
}

}  // namespace mmhost
