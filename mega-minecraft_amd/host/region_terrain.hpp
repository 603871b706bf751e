// mmgen host side — the region-batched streaming scheduler (second half).
//
// Same public interface as the reference's `Terrain` (src/terrain/terrain.hpp:53-120: init / tick / setCurrentChunkPos /
// getCurrentChunkPos / getDrawableChunks / getMaxNumDrawableChunks) and the same result - the same chunks become drawable, with the
// same blocks and the same meshes - but the work is scheduled in CHUNKS, not in "action time": instead of draining nine per-stage
// queues through host `Chunk` objects (≈ 800 chunks/s by construction of the budget, terrain.cpp:65-82), a tick
//   1. finds the chunks around the player that do not exist yet (drawable square of radius chunkVbosGenRadius plus the one-chunk
//      ring whose blocks the meshes of the border chunks look at),
//   2. covers them with a few rectangles and generates each rectangle with ONE device-resident region call (all stages, no host
//      round trip; results are independent of the rectangle decomposition - tests config4 / config5),
//   3. keeps the blocks in a device pool of chunk slots and meshes every chunk whose four neighbours exist in one
//      mmgen_mesh_count / mmgen_mesh_fill pair over the pool,
//   4. copies blocks and meshes into the host `Chunk` objects for consumers that want them there (optional).
// Chunk lifetime (the reference never frees, terrain.cpp:63 DESTROY_ZONES is off): chunks farther than
// `dropRadius` from the player are destroyed and their pool slots recycled, so a bounded pool serves an unbounded walk; a chunk that
// is needed again is simply regenerated (generation is a pure function of position, so it comes back identical).
// SEVERAL GPUs ("multi-GPU streaming"): `RegionTerrain(pool, devices)` keeps one LANE per listed device - its own region
// handle (zone cache, placement ring), chunk pool, placement cache and stream.  A tick deals its rectangles to the lanes (large ones are
// cut into one z-strip per lane first); a region call builds the 3-chunk placement ring and the erosion padding it needs itself, so lanes
// exchange nothing while they generate.  Meshing looks one chunk across: a chunk is meshed by the lane that owns it, and neighbours owned
// by another lane are copied device-to-device (hipMemcpyPeerAsync) into ghost slots behind that lane's pool first.  Which lane made a
// chunk can never show in it (a chunk is a function of its position): mmgen_region_terrain_demo holds a two-lane run - two handles on one
// GPU where there is only one - to the single-lane run and to the oracle's digests.
#pragma once
#include <map>
#include <memory>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "chunk.hpp"

namespace mmhost {

class RegionTerrain {
public:
    static constexpr int chunkVbosGenRadius = 16;                 // terrain.cpp:65
    // devices: one lane per entry (HIP device ordinals; an ordinal may repeat: several handles on one GPU).  Empty = the current device.
    explicit RegionTerrain(size_t poolChunks = 8192, std::vector<int> devices = {});
    ~RegionTerrain();
    void init();
    void tick(float deltaTime);
    std::unordered_set<Chunk*> getDrawableChunks();
    ivec2 getCurrentChunkPos() const { return currentChunkPos; }
    void setCurrentChunkPos(ivec2 p) { currentChunkPos = p; }
    static int getMaxNumDrawableChunks() { return (2 * chunkVbosGenRadius + 1) * (2 * chunkVbosGenRadius + 1); }

    bool allQueuesEmpty() const { return !pending; }
    // Device-resident strips (copyToHost = false, at most 128 chunks to mesh) end a tick with their mesh ENQUEUED: the counts come back with
    // one small copy whose completion the next tick - or getDrawableChunks / finish - waits for, after its own planning, so the host plans
    // tick i + 1 while the device finishes tick i (the reference's ticks are asynchronous in the same way: a chunk becomes drawable
    // when its stage events have fired, terrain.cpp:587-640).  finish(): everything enqueued is complete and booked.
    void finish();
    Chunk* findChunk(ivec2 worldChunkPos);
    size_t numChunks() const { return cells.size(); }

    int maxChunksPerTick = 4096;          // generation budget of one tick, in chunks
    int dropRadius = 40;                  // chunks farther than this (Chebyshev) from the player are destroyed; = chunkMaxGenRadius of the reference
    bool copyToHost = true;               // false: blocks and meshes stay on the device (renderer interop), Chunk::blocks / verts stay empty (set before the first tick)
    int zoneCacheZones = 128;             // eroded zones kept on the device (151 MB; set before init(); 0 = relax every covering zone of every region anew)
    void zoneCacheStats(long long& hits, long long& misses) const;
    bool cachePlacements = true;          // keep the placement lists of generated chunks and feed them back as ring cells of later regions
    bool packedTransfer = true;           // blocks cross PCIe in the run-length wire format (mmgen_pack_*, ~10 KB instead of 96 KB per chunk)
    size_t lastBlockBytesD2H = 0;         // bytes of block data copied to the host by the last tick
    // last tick's accounting
    int lastGenerated = 0, lastMeshed = 0, lastRegions = 0, lastDropped = 0, lastRingReused = 0, lastRingComputed = 0;
    size_t poolInUse() const;
    int numLanes() const { return (int)lanes.size(); }
    int lastGeneratedOnLane(int lane) const { return lanes[lane].lastGenerated; }
    // device-side results of lane 0's last mesh pass (complete behind finish() / getDrawableChunks(); valid until the next tick)
    const Vertex* deviceVerts() const { return (const Vertex*)lanes[0].d_meshOut; }
    // test hook: digest (host/chunk_digest.hpp) of a chunk's blocks as they lie in its lane's pool - what a device-resident consumer would read
    uint64_t deviceBlocksDigest(ivec2 worldChunkPos);

private:
    // Host bookkeeping that sits on a tick's critical path (the planning happens before anything is enqueued: the GPU idles through it).
    // Chunk positions are hashed, not ordered (a tick asks "does this chunk exist" ~3 500 times: 0.25 ms of red-black-tree walks with
    // std::map, a fifth of a streaming tick), and the 240 KB host `Chunk` objects are recycled instead of freed (glibc serves an
    // allocation of that size with mmap / munmap: 35 of each per tick).
    struct PosHash { size_t operator()(const std::pair<int, int>& p) const { return (size_t)(((uint64_t)(uint32_t)p.first << 32 | (uint32_t)p.second) * 0x9E3779B97F4A7C15ull >> 17); } };
    struct ChunkStore { std::vector<void*> free; ~ChunkStore() { for (void* p : free) ::operator delete(p); } };
    struct ChunkRecycler { ChunkStore* store; void operator()(Chunk* c) const { c->~Chunk(); store->free.push_back((void*)c); } };
    typedef std::unique_ptr<Chunk, ChunkRecycler> ChunkPtr;
    ChunkStore chunkStore;                // (declared before `cells`: destroyed after them)
    ChunkPtr newChunk(ivec2 worldChunkPos);
    struct Cell { ChunkPtr chunk; int slot; bool meshed; int lane; };
    std::unordered_map<std::pair<int, int>, Cell, PosHash> cells;
    std::unordered_set<Chunk*> drawable;
    ivec2 currentChunkPos{0, 0}, plannedFor{0, 0};
    bool planned = false, pending = true;
    unsigned long long tickCount = 0;
    ivec2 droppedAt{0, 0}, completeAt{0, 0};      // where the far chunks were last looked for; the centre of the last completed plan
    bool droppedOnce = false, completeValid = false;
    size_t poolChunks;                    // per lane
    static constexpr size_t kGhostSlots = 512;      // per lane, behind the pool: other lanes' chunks that this lane's meshes look at in one tick

    struct Lane {
        int device = 0;
        hipStream_t stream = nullptr;
        mmgen_region* region = nullptr;
        uint8_t* d_pool = nullptr;            // [poolChunks + kGhostSlots][98304]
        bool generationOutstanding = false;   // a device-resident region call whose completion nothing has waited for yet
        std::vector<int> freeSlots;           // sorted descending: slots are handed out in ascending order, so fresh pools fill contiguously
        uint8_t* d_stage = nullptr; size_t stageChunks = 0;      // region output when the free slots are not one contiguous run
        void* d_meshOut = nullptr; size_t meshOutCap = 0;
        uint64_t meshVertsPerChunkCap = 49152;      // output capacity per chunk of a strip's capped mesh fill: grows to 1.25 x the largest chunk seen
        void* d_meshWork = nullptr; size_t meshWorkCap = 0;
        std::vector<uint8_t> hostStage;
        // placement-list cache (device): one slot per chunk whose lists this lane knows, same per-cell layout as the region's placement grid
        std::unordered_map<std::pair<int, int>, int, PosHash> placementSlot;
        std::vector<int> freePlacementSlots;
        FeaturePlacement* d_cacheFp = nullptr;
        CaveFeaturePlacement* d_cacheCfp = nullptr;
        int32_t* d_cacheCnt = nullptr;
        size_t cacheCells = 0;
        void* d_idxWork = nullptr; size_t idxWorkCap = 0;
        int lastGenerated = 0;
        long long totalGenerated = 0;
        // pinned arena for the small host -> device uploads of a tick (index lists, mesh inputs, vertex offsets): an asynchronous copy from
        // pinned memory needs no synchronisation before the host goes on (a blocking copy from pageable memory is a 15 - 25 us round trip,
        // four of them per tick).  Bump-allocated, reset at the start of a tick (the previous tick ended with a synchronisation).
        char* h_pin = nullptr; size_t pinCap = 0, pinUsed = 0, pinLimit = 0;      // (a tick uses one half of the arena, the next tick the other: an
                                                                                  // enqueued strip's uploads are still to be read when the next tick stages its own)
        // a strip's mesh that has been enqueued but not booked yet (completeMesh): its chunks, what the fill could hold, where the counts land
        struct PendingMesh { bool active = false; std::vector<Cell*> work; uint64_t capacity = 0; size_t totalAt = 0; } pm;
        hipEvent_t evMesh = nullptr;          // behind the read-back of the pending mesh's counts
        char* h_back = nullptr;               // pinned: counts [n] .. total of the pending mesh (8 KB)
    };
    std::vector<Lane> lanes;
    void use(const Lane& L);
    void copySync(const Lane& L, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, const char* what);
    void uploadAsync(Lane& L, void* dst, const void* src, size_t bytes, const char* what);
    void uploadAsyncOn(Lane& L, hipStream_t st, void* dst, const void* src, size_t bytes, const char* what);
    void generateRect(int lane, int cx0, int cz0, int nx, int nz);
    void dropFarChunks();
    void meshReady();
    void meshLane(int lane);
    bool completeMesh(Lane& L);          // false: the fill's buffers were too small - its chunks are unmeshed again, the capacity has grown
    void* ensure(void*& p, size_t& cap, size_t bytes);
};

}  // namespace mmhost
