// mmgen region pipeline (host orchestration, C++): generates a rectangle of chunks with every stage resident in HBM.
//
// This is the device-resident replacement of the reference's per-stage host round trips (Terrain::tick dispatch,
// src/terrain/terrain.cpp:643-937 → Chunk::generateHeightfields / gatherHeightfield / generateLayers / erodeZone /
// generateCaves / generateFeaturePlacements / gatherFeaturePlacements / fill, src/terrain/chunk.cu).  Same stages, same
// order, same results; what changes is where the data lives between stages.
//
// Region semantics (canonical "world" definition): for the requested rectangle R of chunks
//   P = R grown by the 3-chunk feature ring (chunk.cu:1158-1167) when features are on;
//   Z = the 12-aligned zones intersecting P (zonePosFromChunkPos, terrain.cpp:259-262);
//   A = union of the zones' 24x24 gathered areas (terrain.cpp:471-522) — heightfield + RAW layers are generated on A;
//   each zone is eroded from RAW planes (never from a neighbour zone's eroded result), the centre 12x12 is kept;
//   caves + placements are generated on P (or on the caller-selected subset when ring cells arrive from another GPU),
//   gather / fill / features / decorators on R.
#include <hip/hip_runtime.h>
#include <map>
#include <utility>
#include <vector>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "../../include/mmgen.h"
#include "mmgen_kernels.h"
#include "mmgen_erosion.h"
#include "mmgen_features.h"
#include "mmgen_prof.h"

namespace {
constexpr size_t mmgen_region_zone_floats = (size_t)144 * 8 * 256;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return (int)e; p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 8;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return (int)e;
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

// A layout table: a view into the region's table arena (HostStage::place), not an allocation of its own.
struct Table {
    void* p = nullptr;
    template <class T> T* as() const { return (T*)p; }
};

// Host side of the layout's index tables: two pinned slots used alternately.  A layout's tables are written into one slot and copied
// from there by ONE asynchronous copy on the caller's stream (the device arena has the slot's layout); the slot is not written again before the event behind those copies has
// fired (two layouts later: long over).  From pageable vectors every copy staged through the runtime and the call ended with a stream
// synchronisation - per tick of a streaming host (a new strip every tick) that was a tenth of the step.
struct HostStage {
    char* slot[2] = {nullptr, nullptr};
    size_t cap = 0, used = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int cur = 0;
    int begin(size_t bytes)                       // picks the other slot, waits until its last copies are over, makes room
    {
        cur ^= 1;
        hipError_t e;
        if (!ev[0]) for (int i = 0; i < 2; ++i) if ((e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)) != hipSuccess) return (int)e;
        if ((e = hipEventSynchronize(ev[cur])) != hipSuccess) return (int)e;
        if (bytes > cap) {
            if ((e = hipEventSynchronize(ev[cur ^ 1])) != hipSuccess) return (int)e;
            for (int i = 0; i < 2; ++i) { if (slot[i]) (void)hipHostFree(slot[i]); slot[i] = nullptr; }
            cap = bytes + bytes / 4;
            for (int i = 0; i < 2; ++i) if ((e = hipHostMalloc((void**)&slot[i], cap, hipHostMallocDefault)) != hipSuccess) { cap = 0; return (int)e; }
        }
        used = 0;
        return 0;
    }
    static size_t padded(size_t bytes) { return (bytes + 255) / 256 * 256; }
    // the table into the slot; its device address is the same offset of `devBase` (the whole slot travels with ONE copy: flush)
    int place(Table& dst, const void* src, size_t bytes, char* devBase)
    {
        dst.p = nullptr;
        if (!bytes) return 0;
        if (used + padded(bytes) > cap) return (int)hipErrorInvalidValue;
        std::memcpy(slot[cur] + used, src, bytes);
        dst.p = devBase + used;
        used += padded(bytes);
        return 0;
    }
    int flush(char* devBase, hipStream_t s) { return used ? (int)hipMemcpyAsync(devBase, slot[cur], used, hipMemcpyHostToDevice, s) : 0; }
    int end(hipStream_t s) { return (int)hipEventRecord(ev[cur], s); }
    void release()
    {
        for (int i = 0; i < 2; ++i) { if (ev[i]) { (void)hipEventSynchronize(ev[i]); (void)hipEventDestroy(ev[i]); ev[i] = nullptr; } if (slot[i]) (void)hipHostFree(slot[i]); slot[i] = nullptr; }
        cap = 0;
    }
};

inline int floordiv(int a, int b) { int q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }

// placement lists of whole cells from one placement grid (region buffers or a caller-owned cache) to another
__global__ void __launch_bounds__(256)
k_copy_placements(const mmgen_feature_placement* __restrict__ sfp, const mmgen_cave_feature_placement* __restrict__ scfp, const int32_t* __restrict__ scnt,
                  const int32_t* __restrict__ srcIdx, mmgen_feature_placement* __restrict__ dfp, mmgen_cave_feature_placement* __restrict__ dcfp,
                  int32_t* __restrict__ dcnt, const int32_t* __restrict__ dstIdx)
{
    const int s = srcIdx[blockIdx.x], d = dstIdx[blockIdx.x], t = threadIdx.x;
    static_assert(sizeof(mmgen_feature_placement) * MMGEN_FP_CAP % 16 == 0 && sizeof(mmgen_cave_feature_placement) * MMGEN_CFP_CAP % 16 == 0, "16-byte words");
    constexpr int nF = (int)(sizeof(mmgen_feature_placement) * MMGEN_FP_CAP / 16), nC = (int)(sizeof(mmgen_cave_feature_placement) * MMGEN_CFP_CAP / 16);
    const uint4* a = (const uint4*)(sfp + (size_t)MMGEN_FP_CAP * s);
    uint4* b = (uint4*)(dfp + (size_t)MMGEN_FP_CAP * d);
    for (int i = t; i < nF; i += 256) b[i] = a[i];
    const uint4* c = (const uint4*)(scfp + (size_t)MMGEN_CFP_CAP * s);
    uint4* e = (uint4*)(dcfp + (size_t)MMGEN_CFP_CAP * d);
    for (int i = t; i < nC; i += 256) e[i] = c[i];
    if (t < 2) dcnt[2 * d + t] = scnt[2 * s + t];
}

// zone cache, hit: the kept chunks' eroded planes from the cache slot into the chunk-major layers of P, then
// Chunk::fixBackwardStratifiedLayers for the chunk (chunk.cu:725-749), like k_erode_finish does for a zone that was relaxed
__global__ void __launch_bounds__(256) k_zone_cache_read(const float* __restrict__ cache, const int* __restrict__ slots, const int* __restrict__ idxOut /*[zones][144], -1 = skip*/,
                                                          float* __restrict__ layersOut)
{
    const int zone = blockIdx.y, cc = blockIdx.x, t = threadIdx.x;
    const int chunk = idxOut[zone * 144 + cc];
    if (chunk < 0) return;
    const float* src = cache + (size_t)slots[zone] * mmgen_region_zone_floats + (size_t)cc * 8 * 256 + t;
    float* col = layersOut + (size_t)MMGEN_LAYERS_SIZE * chunk + t;
    float start12 = 0.f;
#pragma unroll
    for (int plane = 0; plane < 8; ++plane) {
        const float v = src[256 * plane];
        col[256 * (MMGEN_NUM_STRATIFIED_MATERIALS + plane)] = v;
        if (plane == 0) start12 = v;
    }
    col[256 * 10] = start12 - col[256 * 10];
    col[256 * 11] = start12 - col[256 * 11];
}

__global__ void __launch_bounds__(256) k_select(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, int floatsPerChunk)
{
    const int i = blockIdx.x;
    const int o = blockIdx.y * 256 + threadIdx.x;
    if (o < floatsPerChunk) dst[(size_t)floatsPerChunk * i + o] = src[(size_t)floatsPerChunk * idx[i] + o];
}


// ---- compact ring exchange ("counts then payload") ------------------------------------------------------------------
// A cell's message is its two list lengths (header) and then only the entries that exist: 5 words per surface placement, 6 per cave
// placement, instead of the dense [256][5] + [1024][6] words of the placement grid (29.7 KB per cell).
__global__ void __launch_bounds__(256)
k_ring_header(const int32_t* __restrict__ counts, const int32_t* __restrict__ cells, int n, int32_t* __restrict__ header)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = cells[i];
    // raw counts travel (a cave count may exceed its cap: the surplus was dropped when the list was written and is reported by the
    // count); the payload carries min(count, cap) entries
    header[2 * i] = counts[2 * c];
    header[2 * i + 1] = counts[2 * c + 1];
}

// exclusive scan of the cells' payload words (5 c0 + 6 c1) -> offsets[n + 1]; one workgroup (n is a few thousand cells at most)
__global__ void __launch_bounds__(1024)
k_ring_offsets(const int32_t* __restrict__ header, int n, int32_t* __restrict__ offsets)
{
    __shared__ int s_part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = min(lo + per, n);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += 5 * min(header[2 * i], MMGEN_FP_CAP) + 6 * min(header[2 * i + 1], MMGEN_CFP_CAP);
    s_part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = (t >= d) ? s_part[t - d] : 0;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    int run = s_part[t] - sum;
    for (int i = lo; i < hi; ++i) {
        offsets[i] = run;
        run += 5 * min(header[2 * i], MMGEN_FP_CAP) + 6 * min(header[2 * i + 1], MMGEN_CFP_CAP);
    }
    if (t == 1023) offsets[n] = s_part[1023];
}

template <bool PACK>
__global__ void __launch_bounds__(256)
k_ring_move(mmgen_feature_placement* __restrict__ fp, mmgen_cave_feature_placement* __restrict__ cfp, int32_t* __restrict__ counts,
            const int32_t* __restrict__ cells, const int32_t* __restrict__ header, const int32_t* __restrict__ offsets, int32_t* __restrict__ payload)
{
    const int i = blockIdx.x, t = threadIdx.x;
    const int c = cells[i];
    const int n0 = 5 * min(header[2 * i], MMGEN_FP_CAP), n1 = 6 * min(header[2 * i + 1], MMGEN_CFP_CAP);
    int32_t* a = (int32_t*)(fp + (size_t)MMGEN_FP_CAP * c);
    int32_t* b = (int32_t*)(cfp + (size_t)MMGEN_CFP_CAP * c);
    int32_t* w = payload + offsets[i];
    if (PACK) {
        for (int k = t; k < n0; k += 256) w[k] = a[k];
        for (int k = t; k < n1; k += 256) w[n0 + k] = b[k];
    } else {
        for (int k = t; k < n0; k += 256) a[k] = w[k];
        for (int k = t; k < n1; k += 256) b[k] = w[n0 + k];
        if (t < 2) counts[2 * c + t] = header[2 * i + t];
    }
}

// ---- one-phase ring exchange: fixed-size messages, list lengths in-band (no host read between counting and sending) ----------------
// The cells of every peer form one message of a size both sides know from the layout alone: the cells' raw list lengths (2 words per
// cell), then their entries packed back to back, then slack up to `cap` payload words.  slot[i] = {word of the cell's two lengths, first
// payload word of its peer's message, index of that peer's first cell in the cell list, the message's payload capacity}.  A message
// whose entries do not fit is incomplete: both sides see it from the lengths (same arithmetic) and raise *overflow.
__global__ void __launch_bounds__(256)
k_ring_msg_header(const int32_t* __restrict__ counts, const int32_t* __restrict__ cells, const int4* __restrict__ slot, int n, int32_t* __restrict__ header,
                  int32_t* __restrict__ messages)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = cells[i];
    const int c0 = counts[2 * c], c1 = counts[2 * c + 1];
    header[2 * i] = c0; header[2 * i + 1] = c1;
    messages[slot[i].x] = c0; messages[slot[i].x + 1] = c1;
}

__global__ void __launch_bounds__(256)
k_ring_msg_read_header(const int32_t* __restrict__ messages, const int4* __restrict__ slot, int n, int32_t* __restrict__ header)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    header[2 * i] = messages[slot[i].x]; header[2 * i + 1] = messages[slot[i].x + 1];
}

template <bool PACK>
__global__ void __launch_bounds__(256)
k_ring_msg_move(mmgen_feature_placement* __restrict__ fp, mmgen_cave_feature_placement* __restrict__ cfp, int32_t* __restrict__ counts,
                const int32_t* __restrict__ cells, const int4* __restrict__ slot, const int32_t* __restrict__ header, const int32_t* __restrict__ offsets,
                int32_t* __restrict__ messages, int* __restrict__ overflow)
{
    const int i = blockIdx.x, t = threadIdx.x;
    const int c = cells[i];
    const int4 sl = slot[i];
    const int n0 = 5 * min(header[2 * i], MMGEN_FP_CAP), n1 = 6 * min(header[2 * i + 1], MMGEN_CFP_CAP);
    const int rel = offsets[i] - offsets[sl.z];                        // payload words of the peer's earlier cells
    if (rel + n0 + n1 > sl.w) {                                       // this cell's entries do not fit the message
        if (t == 0) atomicMax(overflow, rel + n0 + n1);
        if (!PACK && t < 2) counts[2 * c + t] = 0;
        return;
    }
    int32_t* a = (int32_t*)(fp + (size_t)MMGEN_FP_CAP * c);
    int32_t* b = (int32_t*)(cfp + (size_t)MMGEN_CFP_CAP * c);
    int32_t* w = messages + sl.y + rel;
    if (PACK) {
        for (int k = t; k < n0; k += 256) w[k] = a[k];
        for (int k = t; k < n1; k += 256) w[n0 + k] = b[k];
    } else {
        for (int k = t; k < n0; k += 256) a[k] = w[k];
        for (int k = t; k < n1; k += 256) b[k] = w[n0 + k];
        if (t < 2) counts[2 * c + t] = header[2 * i + t];
    }
}

}  // namespace

struct mmgen_region {
    int cx0 = 0, cz0 = 0, nx = 0, nz = 0;
    unsigned flags = 0;
    int ring = 0, px0 = 0, pz0 = 0, pnx = 0, pnz = 0, np = 0;
    int ax0 = 0, az0 = 0, anx = 0, anz = 0, na = 0;
    int nCompute = 0, nZones = 0, nLazy = 0;
    bool began = false;
    DevBuf hfA, bwA, gathA, layersA;
    DevBuf layersP, caveP, colInfo, fp, cfp, counts;
    // the list lengths of the NEXT step: cleared by this step's finish on the caller's stream, where it runs beside the rasterisers, so that a
    // begin whose caller provides ring cells (they must read as empty until it has written them) swaps buffers instead of putting a 5 us
    // memset in front of its first kernel.  countsSpareClean = bytes of it known to be zero (0: dirty or absent)
    DevBuf countsSpare;
    size_t countsSpareClean = 0;
    DevBuf erodeWork, erodeState, gfp, gcfp, bounds, fillQueue, colNeed, applyWork;
    // the layout's tables: views into ONE device arena that mirrors the pinned slot they are built in, so that a new layout costs one
    // host-to-device copy instead of one per table (six to nine 5 us copies per tick of a streaming host, back to back in front of its first kernel)
    DevBuf tableArena;
    Table posA, computeList, targets, cellLazy, zoneIdx, zoneIdxOut, hitSlots, hitIdxOut, missSlots;
    bool passesPending = false;
    hipEvent_t evPasses = nullptr;      // behind the erosion branch of the last begin: hostMax[2] (its largest pass count) is valid once it has fired
    // layout cache: the host-built index tables (positions, A->P selection, compute list, zone gather / scatter lists, fill targets)
    // depend only on (rectangle, flags, mask); a caller that regenerates the same layout (bench loop, fixed tiles) re-uses the
    // device copies and region_begin / region_finish issue no host rebuild, no H2D copy and no stream synchronisation.
    bool layoutValid = false;
    int kcx0 = 0, kcz0 = 0, knx = 0, knz = 0;
    unsigned kflags = 0;
    bool kHasMask = false;
    std::vector<uint8_t> kMask;
    // ---- zone cache (mmgen_region_set_zone_cache): the eroded planes of whole zones (12 x 12 kept chunks x 8 planes), kept across region calls.
    // A streaming caller generates thin strips; every strip touches up to ten zones, each of which costs 576 chunks of K1 / K2 and a full
    // relaxation for a few dozen new chunks.  Erosion is a pure function of the zone's position (canonical raw padding),
    // so a zone relaxed once serves every later region that touches it.
    static constexpr size_t kZoneCacheFloats = (size_t)144 * 8 * 256;
    int zoneCacheCap = 0;
    DevBuf zoneCache;                                   // [cap][144 chunks][8 planes][256]
    std::map<std::pair<int, int>, int> zoneSlotOf;      // zone (chunk coordinates of its first kept chunk) -> slot
    std::vector<std::pair<int, int>> slotZone;          // slot -> zone (valid if slotUsed)
    std::vector<char> slotUsed;
    std::vector<long long> slotStamp;                   // last use
    long long zoneStamp = 0;
    int nHitZones = 0, nMissZones = 0;
    long long zoneHits = 0, zoneMisses = 0;             // since the cache was (re)sized
    HostStage stage;                                    // pinned host side of the layout's tables
    // (tables hitSlots [nHit] slot, hitIdxOut [nHit][144] P index or -1, missSlots [nMiss] slot or -1: above)
    const unsigned* fillStarted = nullptr;      // started-workgroups word of the early fill's k_fill_cave (one slice) and the value it reaches
    unsigned fillStartedTarget = 0u;
    bool filled = false;          // mmgen_region_fill already ran for the current begin
    uint8_t* filledInto = nullptr;
    uint8_t* earlyBlocks = nullptr;      // mmgen_region_set_output: where the next begin may already put the base blocks
    // ---- stage DAG.  The caller's stream carries K1 / K2, the caves and the placement stages; the erosion branch
    // runs beside the caves on sErode; the base fill runs on sFill in z slices, the rasterisers / decorators of slice i follow on sApply
    // beside the fill of slice i + 1.  serial = everything on the caller's stream in the reference's stage order (per-kernel
    // attribution for the roofline; MMGEN_REGION_SERIAL=1 or mmgen_region_set_serial).
    int* hostMax = nullptr;       // pinned + mapped: [2] largest erosion pass count of the last begin (written by the relaxation itself, no copy);
                                  // [0] largest cave list length beyond MMGEN_CFP_CAP seen by a finish (0 = none); [1] the error word
                                  // of a relaxation that gave up (k_erode_zones; 0 = none).  Its zones are relaxed by the rescue pass of
                                  // the same enqueue, so this is bookkeeping: the next call counts it (mmgen_erosion_stalls) and clears it
    int* hostMaxDev = nullptr;    // its device address
    DevBuf devMax;                // [0] largest cave list length of the finishes since the last query (whatever its size); [1], [2] the longest
                                  // gathered (un-truncated) surface / cave list since the last mmgen_region_max_gathered
    bool serial = false;
    int wantSlices = 0;           // 0 = automatic
    hipStream_t sErode = nullptr, sFill = nullptr, sApply = nullptr;
    static constexpr int kMaxSlices = 16;
    hipEvent_t evK2 = nullptr, evResident = nullptr, evCaveVoxels = nullptr, evBegin = nullptr, evErosion = nullptr, evGather = nullptr, evF1 = nullptr, evTail = nullptr, evFill[kMaxSlices] = {}, evEntry = nullptr, evFillCleared = nullptr;
    int nSlices = 1;
    int sliceRow[kMaxSlices + 1] = {};        // rows of R per slice: [sliceRow[i], sliceRow[i + 1])
    int init_streams()
    {
        if (sErode) return 0;
        hipError_t e;
        // the erosion branch at the highest stream priority: its one persistent launch is ordered ahead of the caves' by an event, the
        // priority only helps its short tail kernels (finish, fix-up) to a free slot beside the caves
        int prLeast = 0, prGreatest = 0;
        if ((e = hipDeviceGetStreamPriorityRange(&prLeast, &prGreatest)) != hipSuccess) return (int)e;
#ifndef MM_ERODE_STREAM_HIGH
#define MM_ERODE_STREAM_HIGH 1
#endif
        if ((e = hipStreamCreateWithPriority(&sErode, hipStreamNonBlocking, MM_ERODE_STREAM_HIGH ? prGreatest : 0)) != hipSuccess) return (int)e;
#ifndef MM_FILL_STREAM_HIGH
#define MM_FILL_STREAM_HIGH 2
#endif
        // the fill and the rasterisers at NORMAL priority.  At the highest (tried in round 5: they are the step's critical path behind the
        // caves) a Python host sees no difference, but a host that enqueues many steps ahead (mmgen_tiled_demo) loses 0.8 ms per step: every
        // hand-over between the caller's queue and a high-priority one then costs 0.05 - 0.16 ms instead of 0.01 - 0.02
        // (profiles/LOG.md, r05z C++ host)
        const int prFill = MM_FILL_STREAM_HIGH == 1 ? prGreatest : MM_FILL_STREAM_HIGH == 2 ? 0 /*normal*/ : prLeast;
        if ((e = hipStreamCreateWithPriority(&sFill, hipStreamNonBlocking, prFill)) != hipSuccess) return (int)e;
        if ((e = hipStreamCreateWithPriority(&sApply, hipStreamNonBlocking, prFill)) != hipSuccess) return (int)e;
        hipEvent_t* ev[] = {&evK2, &evResident, &evCaveVoxels, &evBegin, &evErosion, &evGather, &evF1, &evTail, &evEntry, &evFillCleared};
        for (hipEvent_t* x : ev) if ((e = hipEventCreateWithFlags(x, hipEventDisableTiming)) != hipSuccess) return (int)e;
        for (int i = 0; i < kMaxSlices; ++i)
            if ((e = hipEventCreateWithFlags(&evFill[i], hipEventDisableTiming)) != hipSuccess) return (int)e;
        return 0;
    }
    ~mmgen_region()
    {
        stage.release();
        if (hostMax) (void)hipHostFree(hostMax);
        if (evPasses) (void)hipEventDestroy(evPasses);
        devMax.release();
        if (sErode) {
            (void)hipStreamSynchronize(sErode); (void)hipStreamSynchronize(sFill); (void)hipStreamSynchronize(sApply);
            (void)hipStreamDestroy(sErode); (void)hipStreamDestroy(sFill); (void)hipStreamDestroy(sApply);
            hipEvent_t ev[] = {evK2, evResident, evCaveVoxels, evBegin, evErosion, evGather, evF1, evTail, evEntry, evFillCleared};
            for (hipEvent_t x : ev) if (x) (void)hipEventDestroy(x);
            for (int i = 0; i < kMaxSlices; ++i) if (evFill[i]) (void)hipEventDestroy(evFill[i]);
        }
        DevBuf* all[] = {&tableArena, &hfA, &bwA, &gathA, &layersA, &layersP, &caveP, &colInfo, &fp, &cfp, &counts, &countsSpare,
                         &erodeWork, &erodeState, &gfp, &gcfp, &bounds, &fillQueue, &colNeed, &applyWork, &zoneCache};
        for (DevBuf* b : all) b->release();
    }
};

#define CK(expr) do { int e_ = (int)(expr); if (e_) return e_; } while (0)

extern "C" {

int mmgen_region_create(mmgen_region** out)
{
    if (!out) return (int)hipErrorInvalidValue;
    *out = new mmgen_region();
    {
        mmgen_region* r = *out;
        hipError_t he = hipHostMalloc((void**)&r->hostMax, 4 * sizeof(int), hipHostMallocMapped);
        if (he == hipSuccess) { r->hostMax[0] = r->hostMax[1] = r->hostMax[2] = r->hostMax[3] = 0; he = hipHostGetDevicePointer((void**)&r->hostMaxDev, r->hostMax, 0); }
        if (he == hipSuccess && r->devMax.ensure(3 * sizeof(int)) == 0) he = hipMemset(r->devMax.p, 0, 3 * sizeof(int));
        else if (he == hipSuccess) he = hipErrorOutOfMemory;
        if (he == hipSuccess) he = hipEventCreateWithFlags(&r->evPasses, hipEventDisableTiming);
        if (he != hipSuccess) { delete r; *out = nullptr; return (int)he; }
    }
    const char* e = getenv("MMGEN_REGION_SERIAL");
    (*out)->serial = e && *e && *e != '0';
    const char* sl = getenv("MMGEN_REGION_SLICES");
    (*out)->wantSlices = sl ? atoi(sl) : 0;
    return 0;
}

int mmgen_region_set_zone_cache(mmgen_region* r, int max_zones)
{
    if (!r || max_zones < 0) return (int)hipErrorInvalidValue;
    if (r->sErode) { CK(hipStreamSynchronize(r->sErode)); CK(hipStreamSynchronize(r->sFill)); CK(hipStreamSynchronize(r->sApply)); }
    CK(hipDeviceSynchronize());                        // (a resize frees planes a queued kernel may still read)
    r->zoneSlotOf.clear();
    r->slotZone.assign(max_zones, {0, 0}); r->slotUsed.assign(max_zones, 0); r->slotStamp.assign(max_zones, 0);
    r->zoneHits = r->zoneMisses = 0;
    r->zoneCacheCap = 0;
    r->layoutValid = false;
    if (max_zones == 0) { r->zoneCache.release(); return 0; }
    CK(r->zoneCache.ensure(sizeof(float) * mmgen_region::kZoneCacheFloats * (size_t)max_zones));
    r->zoneCacheCap = max_zones;
    return 0;
}

int mmgen_region_zone_cache_stats(const mmgen_region* r, long long* hits, long long* misses)
{
    if (!r) return (int)hipErrorInvalidValue;
    if (hits) *hits = r->zoneHits;
    if (misses) *misses = r->zoneMisses;
    return 0;
}

int mmgen_region_set_serial(mmgen_region* r, int serial, int slices)
{
    if (!r || slices < 0 || slices > mmgen_region::kMaxSlices) return (int)hipErrorInvalidValue;
    // a change of schedule in the middle of a step (after a begin or a fill): whatever runs on the internal streams is finished first,
    // so that the other schedule's ordering assumptions hold from here on
    if (r->sErode) { CK(hipStreamSynchronize(r->sErode)); CK(hipStreamSynchronize(r->sFill)); CK(hipStreamSynchronize(r->sApply)); }
    r->serial = serial != 0;
    r->wantSlices = slices;
    r->layoutValid = false;
    return 0;
}

void mmgen_region_destroy(mmgen_region* r) { delete r; }

// Host-built index tables of one layout -> device (only when the layout differs from the cached one).
static int region_layout(mmgen_region* r, int cx0, int cz0, int nx, int nz, unsigned flags, const uint8_t* h_local_mask, hipStream_t s)
{
    const bool erosion = flags & MMGEN_REGION_EROSION, features = flags & MMGEN_REGION_FEATURES;
    const int ring = features ? 3 : 0;
    const size_t maskBytes = (size_t)(nx + 2 * ring) * (nz + 2 * ring);
    const bool same = r->layoutValid && r->kcx0 == cx0 && r->kcz0 == cz0 && r->knx == nx && r->knz == nz && r->kflags == flags &&
                      r->kHasMask == (h_local_mask != nullptr) &&
                      (!h_local_mask || (r->kMask.size() == maskBytes && std::memcmp(r->kMask.data(), h_local_mask, maskBytes) == 0));
    if (same && r->zoneCacheCap == 0) return 0;      // (with the zone cache the layout depends on what is cached: rebuilt every call)
    r->layoutValid = false;

    r->cx0 = cx0; r->cz0 = cz0; r->nx = nx; r->nz = nz; r->flags = flags;
    r->ring = ring;
    r->px0 = cx0 - ring; r->pz0 = cz0 - ring; r->pnx = nx + 2 * ring; r->pnz = nz + 2 * ring; r->np = r->pnx * r->pnz;
    const int np = r->np;

    std::vector<int> zonesX, zonesZ, hitX, hitZ, hitSlot, missSlot;
    const bool zoneCaching = erosion && r->zoneCacheCap > 0;
    if (erosion) {
        const int zx0 = floordiv(r->px0, MMGEN_ZONE_SIZE) * MMGEN_ZONE_SIZE, zz0 = floordiv(r->pz0, MMGEN_ZONE_SIZE) * MMGEN_ZONE_SIZE;
        const int zx1 = floordiv(r->px0 + r->pnx - 1, MMGEN_ZONE_SIZE) * MMGEN_ZONE_SIZE, zz1 = floordiv(r->pz0 + r->pnz - 1, MMGEN_ZONE_SIZE) * MMGEN_ZONE_SIZE;
        r->ax0 = zx0 - 6; r->az0 = zz0 - 6; r->anx = (zx1 - zx0) + 24; r->anz = (zz1 - zz0) + 24;
        ++r->zoneStamp;
        for (int zz = zz0; zz <= zz1; zz += MMGEN_ZONE_SIZE) for (int zx = zx0; zx <= zx1; zx += MMGEN_ZONE_SIZE) {
            if (zoneCaching) {
                auto it = r->zoneSlotOf.find({zx, zz});
                if (it != r->zoneSlotOf.end()) { hitX.push_back(zx); hitZ.push_back(zz); hitSlot.push_back(it->second); r->slotStamp[it->second] = r->zoneStamp; ++r->zoneHits; continue; }
                ++r->zoneMisses;
            }
            zonesX.push_back(zx); zonesZ.push_back(zz);
        }
        if (zoneCaching) {
            // a slot for every zone that is relaxed now: a free one, else the least recently used one that this call does not read
            for (size_t z = 0; z < zonesX.size(); ++z) {
                int best = -1;
                for (int k = 0; k < r->zoneCacheCap; ++k) {
                    if (!r->slotUsed[k]) { best = k; break; }
                    if (r->slotStamp[k] < r->zoneStamp && (best < 0 || r->slotStamp[k] < r->slotStamp[best])) best = k;
                }
                if (best >= 0) {
                    if (r->slotUsed[best]) r->zoneSlotOf.erase(r->slotZone[best]);
                    r->slotUsed[best] = 1; r->slotZone[best] = {zonesX[z], zonesZ[z]}; r->slotStamp[best] = r->zoneStamp;
                    r->zoneSlotOf[{zonesX[z], zonesZ[z]}] = best;
                }
                missSlot.push_back(best);                   // -1: relaxed, used, not kept (more zones in one call than the cache holds)
            }
        }
    } else {
        r->ax0 = r->px0; r->az0 = r->pz0; r->anx = r->pnx; r->anz = r->pnz;
    }
    r->nHitZones = (int)hitX.size(); r->nMissZones = (int)zonesX.size();
    // The raw area A: every cell K1 / K2 run on.  Without the zone cache it is the rectangle of the covering zones' gathered areas; with it,
    // the cells of P plus the gathered areas of the zones that are relaxed in this call only (none when every zone is cached).
    std::vector<std::pair<int, int>> areaCells;             // (x, z) relative to (ax0, az0), in A's order BEHIND the cells of P
    std::vector<int> areaIndex;                             // [anx * anz] A index or -1
    if (zoneCaching) {
        areaIndex.assign((size_t)r->anx * r->anz, -1);
        int next = r->np;
        for (int z = 0; z < r->pnz; ++z) for (int x = 0; x < r->pnx; ++x) areaIndex[(r->px0 + x - r->ax0) + (size_t)r->anx * (r->pz0 + z - r->az0)] = x + r->pnx * z;
        for (size_t zi = 0; zi < zonesX.size(); ++zi)
            for (int cz = 0; cz < 24; ++cz) for (int cx = 0; cx < 24; ++cx) {
                const int x = zonesX[zi] - 6 + cx - r->ax0, z = zonesZ[zi] - 6 + cz - r->az0;
                int& a = areaIndex[x + (size_t)r->anx * z];
                if (a < 0) { a = next++; areaCells.push_back({x, z}); }
            }
        r->na = next;
    } else {
        r->na = r->anx * r->anz;
    }
    r->nZones = (int)zonesX.size();
    const int na = r->na, nr = nx * nz, Z = r->nZones;

    // Order of the raw area A: the cells of the placement grid P first, in P's order, then the padding cells.  Every per-chunk array of
    // A (positions, heights, biome weights) then starts with the array of P - no copies - and only the layers exist twice (raw in A for
    // the zones' padding, eroded in P).
    std::vector<int32_t> posA(2 * (size_t)na), aIndex, computeList, targets(nr);
    std::vector<uint8_t> lazy(np, 0);
    if (zoneCaching) {
        aIndex.assign(areaIndex.begin(), areaIndex.end());
        for (int i = 0; i < np; ++i) { posA[2 * (size_t)i] = (r->px0 + i % r->pnx) * 16; posA[2 * (size_t)i + 1] = (r->pz0 + i / r->pnx) * 16; }
        for (size_t k = 0; k < areaCells.size(); ++k) {
            posA[2 * ((size_t)np + k)] = (r->ax0 + areaCells[k].first) * 16; posA[2 * ((size_t)np + k) + 1] = (r->az0 + areaCells[k].second) * 16;
        }
    } else {
        aIndex.resize((size_t)r->anx * r->anz);
        int next = np;
        for (int z = 0; z < r->anz; ++z) for (int x = 0; x < r->anx; ++x) {
            const int gx = r->ax0 + x - r->px0, gz = r->az0 + z - r->pz0;
            const int a = (gx >= 0 && gx < r->pnx && gz >= 0 && gz < r->pnz) ? gx + r->pnx * gz : next++;
            aIndex[x + (size_t)r->anx * z] = a;
            posA[2 * (size_t)a] = (r->ax0 + x) * 16; posA[2 * (size_t)a + 1] = (r->az0 + z) * 16;
        }
        if (next != na) return (int)hipErrorUnknown;        // P lies inside A by construction
    }
    for (int i = 0; i < np; ++i) {
        const int x = i % r->pnx - ring, z = i / r->pnx - ring;
        const bool inR = x >= 0 && x < nx && z >= 0 && z < nz;
        if (inR || !h_local_mask || h_local_mask[i]) computeList.push_back(i);
        if (!inR && h_local_mask && h_local_mask[i] == 2) lazy[i] = 1;
    }
    r->nCompute = (int)computeList.size();
    r->nLazy = 0;
    for (uint8_t l : lazy) r->nLazy += l;
    for (int z = 0; z < nz; ++z) for (int x = 0; x < nx; ++x) targets[x + nx * z] = (x + ring) + r->pnx * (z + ring);
    // z slices of the rectangle for the fill / rasteriser pipeline
    {
        // measured (profiles/README.md r03): the fill and the rasterisers each fill the register file of every CU, so slice i's rasterisers
        // gain nothing from running beside slice i + 1's fill - one slice unless asked otherwise
        int S = r->serial ? 1 : (r->wantSlices > 0 ? r->wantSlices : 1);
        if (S > nz) S = nz;
        if (S > mmgen_region::kMaxSlices) S = mmgen_region::kMaxSlices;
        r->nSlices = S;
        for (int i = 0; i <= S; ++i) r->sliceRow[i] = (int)((long long)nz * i / S);
    }
    std::vector<int> zi((size_t)Z * 576), zo((size_t)Z * 144);
    for (int z = 0; z < Z; ++z) {
        for (int cz = 0; cz < 24; ++cz) for (int cx = 0; cx < 24; ++cx)
            zi[(size_t)z * 576 + cx + 24 * cz] = aIndex[(zonesX[z] - 6 + cx - r->ax0) + (size_t)r->anx * (zonesZ[z] - 6 + cz - r->az0)];
        for (int cz = 0; cz < 12; ++cz) for (int cx = 0; cx < 12; ++cx) {
            const int gx = zonesX[z] + cx - r->px0, gz = zonesZ[z] + cz - r->pz0;
            zo[(size_t)z * 144 + cx + 12 * cz] = (gx >= 0 && gx < r->pnx && gz >= 0 && gz < r->pnz) ? gx + r->pnx * gz : -1;
        }
    }

    // cached zones: where their kept chunks lie in P
    std::vector<int> ho((size_t)r->nHitZones * 144);
    for (int z = 0; z < r->nHitZones; ++z)
        for (int cz = 0; cz < 12; ++cz) for (int cx = 0; cx < 12; ++cx) {
            const int gx = hitX[z] + cx - r->px0, gz = hitZ[z] + cz - r->pz0;
            ho[(size_t)z * 144 + cx + 12 * cz] = (gx >= 0 && gx < r->pnx && gz >= 0 && gz < r->pnz) ? gx + r->pnx * gz : -1;
        }

    {
        using HS = HostStage;
        const size_t total = HS::padded(sizeof(int32_t) * 2 * na) + HS::padded(sizeof(int) * r->nCompute) + HS::padded(sizeof(int) * nr) + HS::padded(np) +
                             HS::padded(sizeof(int) * zi.size()) + HS::padded(sizeof(int) * zo.size()) + HS::padded(sizeof(int) * hitSlot.size()) +
                             HS::padded(sizeof(int) * ho.size()) + HS::padded(sizeof(int) * missSlot.size());
        CK(r->stage.begin(total));
        CK(r->tableArena.ensure(total));
    }
    char* const dev = r->tableArena.as<char>();
    CK(r->stage.place(r->posA, posA.data(), sizeof(int32_t) * 2 * na, dev));
    CK(r->stage.place(r->computeList, computeList.data(), sizeof(int) * r->nCompute, dev));
    CK(r->stage.place(r->targets, targets.data(), sizeof(int) * nr, dev));
    r->cellLazy.p = nullptr;
    if (r->nLazy) {
        CK(r->colNeed.ensure((size_t)256 * np));
        CK(r->stage.place(r->cellLazy, lazy.data(), np, dev));
    }
    CK(r->stage.place(r->zoneIdx, zi.data(), sizeof(int) * zi.size(), dev));
    CK(r->stage.place(r->zoneIdxOut, zo.data(), sizeof(int) * zo.size(), dev));
    CK(r->stage.place(r->hitSlots, hitSlot.data(), sizeof(int) * hitSlot.size(), dev));
    CK(r->stage.place(r->hitIdxOut, ho.data(), sizeof(int) * ho.size(), dev));
    r->missSlots.p = nullptr;
    if (zoneCaching && Z) CK(r->stage.place(r->missSlots, missSlot.data(), sizeof(int) * missSlot.size(), dev));
    CK(r->stage.flush(dev, s));
    CK(r->stage.end(s));             // (no synchronisation: the host side of the tables is the pinned slot, not these vectors)

    r->kcx0 = cx0; r->kcz0 = cz0; r->knx = nx; r->knz = nz; r->kflags = flags; r->kHasMask = h_local_mask != nullptr;
    if (h_local_mask) r->kMask.assign(h_local_mask, h_local_mask + maskBytes); else r->kMask.clear();
    r->layoutValid = true;
    return 0;
}

#ifndef MMGEN_EROSION_ZONE_BATCH
#define MMGEN_EROSION_ZONE_BATCH 96      // zones relaxed per launch sequence (16 MB of planes each; 288 GB of HBM)
#endif

int mmgen_region_max_cave_placements(mmgen_region* r, int* out_max, void* stream)
{
    if (!r || !out_max) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    int m = 0;
    CK(hipMemcpyAsync(&m, r->devMax.p, sizeof(int), hipMemcpyDeviceToHost, s));
    CK(hipMemsetAsync(r->devMax.p, 0, sizeof(int), s));
    CK(hipStreamSynchronize(s));
    const int h = __atomic_exchange_n(r->hostMax, 0, __ATOMIC_RELAXED);
    *out_max = m > h ? m : h;
    return 0;
}

static int region_fill_on(mmgen_region* r, uint8_t* d_blocks, hipEvent_t after0, hipEvent_t after1, hipEvent_t after2);
#ifndef MM_FILL_CAVE_AFTER_F1
#define MM_FILL_CAVE_AFTER_F1 1      // the scan + cave part of the fill waits for the placement pass to leave the chip (region_fill_on)
#endif
// Workgroups of the relaxation per four CUs when it runs beside the caves (46.7 KB of LDS, 8 waves each): ONE per CU.  It waits more than it
// issues (3.2 ms at one per CU, 1.7 alone on the chip), the caves take the rest of every CU (four of their workgroups fit beside one of
// these, two beside two, none beside three), and what the relaxation costs the caves is its occupancy-time, which is smallest here: with
// every workgroup of it on the chip before the caves start (launch_caves' counter watch) 2 / 3 / 4 / 6 / 8 / 12 per four CUs give
// 21.56 / 21.60 / 21.65 and (another box) 4 / 6 / 8 / 12 -> 21.28 / 21.46 / 21.64 / 21.96 ms per step (profiles/LOG.md, round 5)
#ifndef MMGEN_REGION_EROSION_WG_PER_4CU
#define MMGEN_REGION_EROSION_WG_PER_4CU 4
#endif
// with the base fill starting the moment the caves' extents exist, the cave biomes (which only the placement stages wait for) leave it
// most of every CU: persistent workgroups per CU of k_cave_biomes then (1 / 2 / 3 / 6: 24.60 / 24.73 / 24.77 / 24.81 ms per step, LOG round 4)
#ifndef MM_CAVE_BIOME_WG_BESIDE_FILL
#define MM_CAVE_BIOME_WG_BESIDE_FILL 1
#endif
static constexpr int kCaveBiomeWorkgroupsBesideFill = MM_CAVE_BIOME_WG_BESIDE_FILL;

int mmgen_region_max_gathered(mmgen_region* r, int* out_surface, int* out_cave, void* stream)
{
    if (!r) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    int m[2] = {0, 0};
    CK(hipMemcpyAsync(m, r->devMax.as<int>() + 1, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    CK(hipMemsetAsync(r->devMax.as<int>() + 1, 0, 2 * sizeof(int), s));
    CK(hipStreamSynchronize(s));
    if (out_surface) *out_surface = m[0];
    if (out_cave) *out_cave = m[1];
    return 0;
}

int mmgen_region_begin(mmgen_region* r, int cx0, int cz0, int nx, int nz, unsigned flags, const uint8_t* h_local_mask, void* stream)
{
    if (!r) return (int)hipErrorInvalidValue;
    uint8_t* const early = r->earlyBlocks;      // one-shot: whatever this call returns, the next begin starts without it
    r->earlyBlocks = nullptr;
    if (nx <= 0 || nz <= 0) return (int)hipErrorInvalidValue;
    if (__atomic_load_n(r->hostMax, __ATOMIC_RELAXED) > MMGEN_CFP_CAP) return MMGEN_ERROR_PLACEMENT_OVERFLOW;
    // a relaxation of an earlier step that gave up: its zones were relaxed by the rescue pass (same planes); counted, not an error
    if (__atomic_load_n(r->hostMax + 1, __ATOMIC_RELAXED) != 0) { __atomic_store_n(r->hostMax + 1, 0, __ATOMIC_RELAXED); mmk::erosion_note_stall(); }
    hipStream_t s = (hipStream_t)stream;
    const bool erosion = flags & MMGEN_REGION_EROSION, features = flags & MMGEN_REGION_FEATURES;
    const bool par = !r->serial;
    // a begin -> fill without a finish leaves the fill stream unjoined: order it before this begin's writes (whatever the mode is now:
    // the streams exist if the earlier fill ran on them)
    if (r->filled && r->sFill) CK(hipStreamWaitEvent(s, r->evFill[r->nSlices - 1], 0));
    if (par) {
        CK(r->init_streams());
        CK(hipEventRecord(r->evBegin, s));          // what the caller's stream held when this begin was called
    }
    r->began = false; r->filled = false; r->filledInto = nullptr; r->fillStarted = nullptr; r->fillStartedTarget = 0u;
    // zone cache: region_layout registers the zones this begin is going to relax in the cache map BEFORE anything is enqueued.  If the begin
    // fails after that (an allocation, an upload, the relaxation's enqueue) their slots were never written, and the next begin would take
    // cache hits on them: an error return drops every cached zone (slots and map; the layout is rebuilt).
    struct CacheGuard {
        mmgen_region* r; bool ok = false;
        ~CacheGuard() { if (!ok && r->zoneCacheCap) { r->zoneSlotOf.clear(); std::fill(r->slotUsed.begin(), r->slotUsed.end(), 0); r->layoutValid = false; } }
    } cacheGuard{r};
    CK(region_layout(r, cx0, cz0, nx, nz, flags, h_local_mask, s));
    const int np = r->np, na = r->na;
    hipStream_t sE = par ? r->sErode : s;

    CK(r->hfA.ensure(sizeof(float) * 256 * (size_t)na));
    CK(r->bwA.ensure(sizeof(float) * MMGEN_BIOME_WEIGHTS_SIZE * (size_t)na));
    CK(r->gathA.ensure(sizeof(float) * MMGEN_GATHERED_HEIGHTFIELD_SIZE * (size_t)na));
    CK(r->layersA.ensure(sizeof(float) * MMGEN_LAYERS_SIZE * (size_t)na));
    CK(r->caveP.ensure(sizeof(mmgen_cave_layer) * MMGEN_CAVE_LAYERS_SIZE * (size_t)np));
    CK(r->colInfo.ensure(mmk::cave_scratch_bytes(np)));
    if (features) {
        CK(r->fp.ensure(sizeof(mmgen_feature_placement) * MMGEN_FP_CAP * (size_t)np));
        CK(r->cfp.ensure(sizeof(mmgen_cave_feature_placement) * MMGEN_CFP_CAP * (size_t)np));
        // cells the caller provides (mask 0) read as empty until it has written them; every computed cell's lengths are stored by F1
        const size_t countBytes = sizeof(int) * 2 * (size_t)np;
        if (r->nCompute < np && r->countsSpareClean >= countBytes) {
            std::swap(r->counts, r->countsSpare);       // (the last finish cleared it behind its gather: stream order)
            r->countsSpareClean = 0;
        } else {
            CK(r->counts.ensure(countBytes));
            if (r->nCompute < np) CK(hipMemsetAsync(r->counts.p, 0, countBytes, s));
        }
    }

    // ---- K1 + K2 on the raw area A.  With erosion the P grid's layers exist twice (raw in A for the zones' padding, eroded in P): K2 stores
    // the twelve stratified layers of the P cells - the head of A's order - into both, the eight eroded ones arrive from the relaxation
    if (erosion) CK(r->layersP.ensure(sizeof(float) * MMGEN_LAYERS_SIZE * (size_t)np));
    // In the stage DAG with erosion the layers (K2) go with the erosion branch: nothing on the caller's stream needs them before that branch
    // is joined (the caves and their per-column pass read heights and weights only), so the per-column cave pass runs beside K2 instead of
    // behind it and the caves start 0.1 ms earlier (the relaxation, which K2 feeds, still has to be resident first).
#ifndef MM_LAYERS_ON_BRANCH
#define MM_LAYERS_ON_BRANCH 1
#endif
    const bool layersOnBranch = MM_LAYERS_ON_BRANCH && erosion && par;
    {
        mmk::StageRange sr("mmgen:heightfield+layers");
        CK(mmk::launch_heightfield(r->posA.as<int32_t>(), na, r->hfA.as<float>(), r->bwA.as<float>(), r->gathA.as<float>(), s));
        if (layersOnBranch) { CK(hipEventRecord(r->evK2, s)); CK(hipStreamWaitEvent(sE, r->evK2, 0)); }      // (evK2: "K1 is done" in this schedule)
        CK(mmk::launch_layers(r->gathA.as<float>(), r->bwA.as<float>(), r->posA.as<int32_t>(), na, r->layersA.as<float>(), layersOnBranch ? sE : s,
                              erosion ? r->layersP.as<float>() : nullptr, erosion ? np : 0));
        if (erosion && par && !layersOnBranch) { CK(hipEventRecord(r->evK2, s)); CK(hipStreamWaitEvent(sE, r->evK2, 0)); }
    }

    float *hfP, *bwP, *layersP;
    int32_t* posP;
    if (erosion) {
        // The P grid's arrays are the first np chunks of A's (region_layout orders A that way); only the layers exist twice: eroded planes
        // are scattered into layersP, layersA stays raw for the other zones' padding.
        hfP = r->hfA.as<float>(); bwP = r->bwA.as<float>(); layersP = r->layersP.as<float>(); posP = r->posA.as<int32_t>();
    } else {
        hfP = r->hfA.as<float>(); bwP = r->bwA.as<float>(); layersP = r->layersA.as<float>(); posP = r->posA.as<int32_t>();
        CK(mmk::launch_fix_backward(layersP, np, s));
    }

    const int* list = r->computeList.as<int>();
    const uint8_t* colNeed = nullptr;
    if (r->nLazy && features) {
        // lazy ring cells: only the columns that can produce a placement reaching the rectangle get caves and placements
        CK(mmk::launch_ring_need(bwP, posP, list, r->nCompute, r->cellLazy.as<uint8_t>(), 16 * cx0, 16 * cz0, 16 * (cx0 + nx) - 1, 16 * (cz0 + nz) - 1,
                                 r->colNeed.as<uint8_t>(), s));
        colNeed = r->colNeed.as<uint8_t>();
    }

    const unsigned* startedCounter = nullptr;      // the relaxation's started-workgroups counter (one zone batch only: the next batch re-uses it)
    unsigned startedTarget = 0u;
    // ---- E1 / K3 / E3: the erosion branch, enqueued FIRST.  The relaxation is one persistent launch per zone batch (no host reads): its
    // workgroups are on the chip before the caves' launch starts to fill every free slot, and then run beside it
    if (erosion) {
        mmk::StageRange sr("mmgen:erosion");
        const int Z = r->nZones;
        const int batch = Z < MMGEN_EROSION_ZONE_BATCH ? (Z > 0 ? Z : 1) : MMGEN_EROSION_ZONE_BATCH;
        CK(r->erodeWork.ensure(mmk::erosion_work_bytes(batch)));
        CK(r->erodeState.ensure(mmk::erosion_state_bytes(batch)));
        startedCounter = nullptr; startedTarget = 0u;
        for (int z0 = 0; z0 < Z; z0 += batch) {
            const int nb = (Z - z0) < batch ? (Z - z0) : batch;
            // (no E1 copy: the relaxation reads the zones' raw planes through their chunk lists)
            CK(mmk::erode_zones(nullptr, 0, nb, r->erodeWork.as<float>(), r->erodeState.as<mm::ErosionState>(), nullptr, 0, sE, nullptr,
                                r->zoneIdxOut.as<int>() + (size_t)z0 * 144, layersP, r->hostMaxDev + 2, (par && z0 == 0) ? r->evResident : nullptr,
                                r->layersA.as<float>(), r->hfA.as<float>(), r->zoneIdx.as<int>() + (size_t)z0 * 576,
                                par ? MMGEN_REGION_EROSION_WG_PER_4CU : 0, (par && Z <= batch) ? &startedCounter : nullptr, &startedTarget, (unsigned*)(r->hostMaxDev + 1),
                                /*clearPassesDev*/ z0 == 0, /*fixBackward (E3 fix-up of the kept chunks)*/ true,
                                r->zoneCacheCap > 0 ? r->zoneCache.as<float>() : nullptr, r->zoneCacheCap > 0 ? r->missSlots.as<int>() + z0 : nullptr));
        }
        if (Z == 0) CK(hipMemsetAsync(r->hostMaxDev + 2, 0, sizeof(int), sE));      // (every zone came out of the cache: no relaxation, no passes)
        // zones that were relaxed by an earlier call: their kept chunks' planes out of the cache
        if (r->nHitZones)
            MMK_LAUNCH(mmk::KID_EROSION_SCATTER, k_zone_cache_read, dim3(144, r->nHitZones), dim3(256), sE, (const float*)r->zoneCache.as<float>(),
                       (const int*)r->hitSlots.as<int>(), (const int*)r->hitIdxOut.as<int>(), layersP);
        CK(hipEventRecord(r->evPasses, sE));
        r->passesPending = true;
    }

    // ---- K4 caves on the caller's stream, behind the point at which the relaxation's persistent launch is next in its queue: that launch
    // is a few hundred workgroups and must be on the chip before the caves' 150 000 start taking every slot that frees up
    {
        mmk::StageRange sr("mmgen:caves");
        CK(mmk::launch_caves(hfP, bwP, posP, r->nCompute, r->caveP.as<mmgen_cave_layer>(), r->colInfo.as<float>(), np, list, colNeed, s,
                             par ? r->evCaveVoxels : nullptr, (par && early) ? kCaveBiomeWorkgroupsBesideFill : 0,
                             (erosion && par) ? r->evResident : nullptr, startedCounter, startedTarget));
    }
    if (erosion && par) { CK(hipEventRecord(r->evErosion, sE)); CK(hipStreamWaitEvent(s, r->evErosion, 0)); }
    r->began = true;
    // ---- F1 placements (eroded layers + cave layers of every computed cell)
    if (features) {
        mmk::StageRange sr("mmgen:feature_placements");
        CK(mmk::launch_feature_placements(hfP, bwP, layersP, r->caveP.as<mmgen_cave_layer>(), posP, r->nCompute, r->fp.as<mmgen_feature_placement>(),
                                          r->cfp.as<mmgen_cave_feature_placement>(), r->counts.as<int>(), list, colNeed, s));
        if (par) CK(hipEventRecord(r->evF1, s));
    }
    // ---- the base fill as soon as its inputs exist (the caves' extents, the eroded layers), beside the cave biomes and the placement
    // stages that only the rasterisers wait for (mmgen_region_set_output).  Enqueued after F1 so that its cave part can wait for F1's end
    // (region_fill_on)
    if (par && early) CK(region_fill_on(r, early, /*after*/ r->evBegin, r->evCaveVoxels, erosion ? r->evErosion : nullptr));
    cacheGuard.ok = true;
    return 0;
}

int mmgen_region_placement_buffers(mmgen_region* r, mmgen_feature_placement** d_fp, mmgen_cave_feature_placement** d_cfp, int32_t** d_counts,
                                   int* grid_x0, int* grid_z0, int* grid_w, int* grid_h)
{
    if (!r || !r->began || !(r->flags & MMGEN_REGION_FEATURES)) return (int)hipErrorInvalidValue;
    if (d_fp) *d_fp = r->fp.as<mmgen_feature_placement>();
    if (d_cfp) *d_cfp = r->cfp.as<mmgen_cave_feature_placement>();
    if (d_counts) *d_counts = r->counts.as<int32_t>();
    if (grid_x0) *grid_x0 = r->px0;
    if (grid_z0) *grid_z0 = r->pz0;
    if (grid_w) *grid_w = r->pnx;
    if (grid_h) *grid_h = r->pnz;
    return 0;
}

// Base blocks of the rectangle (kernFill without the feature lists).  Needs nothing from the placement ring, so a tiling caller runs
// it while the ring exchange with the neighbouring GPUs is in flight; mmgen_region_finish then only gathers / rasterises / decorates.
// the rectangle's blocks all lie within the pruning domain of the exact prunings (mm_noise.cuh): the plain kernels are not launched
static bool region_in_prune_domain(const mmgen_region* r)
{
    const long long lim = 32768;      // MM_PRUNE_DOMAIN
    return 16LL * r->cx0 > -lim && 16LL * (r->cx0 + r->nx) - 1 < lim && 16LL * r->cz0 > -lim && 16LL * (r->cz0 + r->nz) - 1 < lim;
}

static size_t slice_queue_bytes(const mmgen_region* r)
{
    int rows = 0;
    for (int i = 0; i < r->nSlices; ++i) rows = (r->sliceRow[i + 1] - r->sliceRow[i]) > rows ? (r->sliceRow[i + 1] - r->sliceRow[i]) : rows;
    return (mmk::fill_queue_bytes(rows * r->nx) + 255) / 256 * 256;
}

// the base fill of the rectangle on the fill stream, behind the given events (parallel schedule only)
static int region_fill_on(mmgen_region* r, uint8_t* d_blocks, hipEvent_t after0, hipEvent_t after1, hipEvent_t after2)
{
    const bool erosion = r->flags & MMGEN_REGION_EROSION;
    float* hfP = r->hfA.as<float>();        // the P grid's arrays are the first np chunks of A's
    float* bwP = r->bwA.as<float>();
    float* layersP = erosion ? r->layersP.as<float>() : r->layersA.as<float>();
    int32_t* posP = r->posA.as<int32_t>();
    mmk::StageRange sr("mmgen:fill");
    const size_t qb = slice_queue_bytes(r);
    CK(r->fillQueue.ensure(qb * r->nSlices));
    hipStream_t sF = r->sFill;
    // the scratch's counters before the waits (the stream's previous fill is over by stream order): when the inputs are ready the first
    // thing in the queue is k_fill_base, not a memset that has to find a free slot beside whatever is running then
    for (int i = 0; i < r->nSlices; ++i)
        CK(mmk::launch_fill_clear((r->sliceRow[i + 1] - r->sliceRow[i]) * r->nx, (unsigned*)((char*)r->fillQueue.p + qb * i), qb, sF));
    CK(hipEventRecord(r->evFillCleared, sF));       // (the gather's watch of the cave fill's started-workgroups word must not see the previous step's count)
    hipEvent_t after[3] = {after0, after1, after2};
    for (hipEvent_t e : after) if (e) CK(hipStreamWaitEvent(sF, e, 0));
    // The cave fill's persistent workgroups (six per CU, 26.3 KB of LDS each) must find the chip EMPTY: started while workgroups of
    // k_feature_placements or k_cave_biomes are resident, the sixth one of many CUs has no contiguous LDS left and starts only when a
    // neighbour exits - k_fill_cave then takes 6.1 ms instead of 5.4 and F1 beside it 2.1 ms instead of 1.0 (profiles/LOG.md, r05z C++
    // host trace).  So the scan + cave part waits for F1's end; k_fill_base still runs beside F1.
    const bool features = r->flags & MMGEN_REGION_FEATURES;
    hipEvent_t beforeCave = (features && MM_FILL_CAVE_AFTER_F1) ? r->evF1 : nullptr;
    for (int i = 0; i < r->nSlices; ++i) {
        const int c0 = r->sliceRow[i] * r->nx, n = (r->sliceRow[i + 1] - r->sliceRow[i]) * r->nx;
        CK(mmk::launch_fill(hfP, bwP, layersP, r->caveP.as<mmgen_cave_layer>(), posP, n, d_blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * c0, r->targets.as<int>() + c0,
                            (unsigned*)((char*)r->fillQueue.p + qb * i), qb, region_in_prune_domain(r), sF, true,
                            r->nSlices == 1 ? &r->fillStarted : nullptr, r->nSlices == 1 ? &r->fillStartedTarget : nullptr, beforeCave));
        CK(hipEventRecord(r->evFill[i], sF));
    }
    r->filled = true; r->filledInto = d_blocks;
    return 0;
}

int mmgen_region_set_output(mmgen_region* r, uint8_t* d_blocks)
{
    if (!r) return (int)hipErrorInvalidValue;
    r->earlyBlocks = d_blocks;
    return 0;
}

int mmgen_region_fill(mmgen_region* r, uint8_t* d_blocks, void* stream)
{
    if (!r || !r->began || !d_blocks) return (int)hipErrorInvalidValue;
    if (r->filled && r->filledInto == d_blocks) return 0;          // begin already issued it (mmgen_region_set_output)
    hipStream_t s = (hipStream_t)stream;
    if (!r->serial) {
        // d_blocks is the caller's: whatever its stream holds at this point (a consumer of the buffer's previous contents, e.g. a copy of
        // the last tile out of memory that the caller's allocator has just handed back) comes before the first write.  That also orders
        // the fill behind this region's caves and erosion (both joined into the caller's stream by begin).
        CK(hipEventRecord(r->evEntry, s));
        return region_fill_on(r, d_blocks, r->evEntry, nullptr, nullptr);
    }
    const bool erosion = r->flags & MMGEN_REGION_EROSION;
    float* layersP = erosion ? r->layersP.as<float>() : r->layersA.as<float>();
    mmk::StageRange sr("mmgen:fill");
    const size_t qb = slice_queue_bytes(r);
    CK(r->fillQueue.ensure(qb * r->nSlices));
    for (int i = 0; i < r->nSlices; ++i) {
        const int c0 = r->sliceRow[i] * r->nx, n = (r->sliceRow[i + 1] - r->sliceRow[i]) * r->nx;
        CK(mmk::launch_fill(r->hfA.as<float>(), r->bwA.as<float>(), layersP, r->caveP.as<mmgen_cave_layer>(), r->posA.as<int32_t>(), n,
                            d_blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * c0, r->targets.as<int>() + c0, (unsigned*)((char*)r->fillQueue.p + qb * i), qb,
                            region_in_prune_domain(r), s));
    }
    r->filled = true; r->filledInto = d_blocks;
    return 0;
}

int mmgen_region_finish(mmgen_region* r, uint8_t* d_blocks, float* d_heightfields, float* d_layers, mmgen_cave_layer* d_cave_layers, void* stream)
{
    if (!r || !r->began || !d_blocks) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const bool erosion = r->flags & MMGEN_REGION_EROSION, features = r->flags & MMGEN_REGION_FEATURES, decor = r->flags & MMGEN_REGION_DECORATORS;
    const bool par = !r->serial;
    const int nr = r->nx * r->nz;
    float* hfP = r->hfA.as<float>();        // the P grid's arrays are the first np chunks of A's
    float* bwP = r->bwA.as<float>();
    float* layersP = erosion ? r->layersP.as<float>() : r->layersA.as<float>();
    int32_t* posP = r->posA.as<int32_t>();
    const int* tgt = r->targets.as<int>();

    if (__atomic_load_n(r->hostMax, __ATOMIC_RELAXED) > MMGEN_CFP_CAP) return MMGEN_ERROR_PLACEMENT_OVERFLOW;
    // a relaxation of an earlier step that gave up: its zones were relaxed by the rescue pass (same planes); counted, not an error
    if (__atomic_load_n(r->hostMax + 1, __ATOMIC_RELAXED) != 0) { __atomic_store_n(r->hostMax + 1, 0, __ATOMIC_RELAXED); mmk::erosion_note_stall(); }
    if (!r->filled || r->filledInto != d_blocks) CK(mmgen_region_fill(r, d_blocks, stream));
    hipStream_t sA = par ? r->sApply : s;
    if (features) {
        mmk::StageRange sr("mmgen:features");
        // (the gather checks every list length it uses, local or received, against MMGEN_CFP_CAP on its way - include/mmgen.h:
        // MMGEN_ERROR_PLACEMENT_OVERFLOW - and clears the rasterisers' work counters: two launches and a memset less per step)
        CK(r->gfp.ensure(sizeof(mmgen_feature_placement) * MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK * (size_t)nr));
        CK(r->gcfp.ensure(sizeof(mmgen_cave_feature_placement) * MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK * (size_t)nr));
        CK(r->bounds.ensure(sizeof(int) * 4 * nr));
        CK(r->applyWork.ensure(mmk::apply_work_bytes() * r->kMaxSlices));      // k_apply_features' work counters, one set per slice
#ifndef MM_GATHER_WAITS_FOR_FILL
#define MM_GATHER_WAITS_FOR_FILL 1
#endif
        // the gather runs beside the cave fill and must not start before it: the fill's persistent workgroups take the chip first, the gather
        // the slots they leave (measured: a gather that starts a few microseconds early holds slots the cave fill then never gets)
        if (MM_GATHER_WAITS_FOR_FILL && par && r->filled && r->fillStarted && r->fillStartedTarget) {
            CK(hipStreamWaitEvent(s, r->evFillCleared, 0));      // the word was cleared on the fill stream (long ago: region_fill_on's first enqueue)
            CK(mmk::launch_wait_counter(r->fillStarted, r->fillStartedTarget, s));
        }
        CK(mmk::launch_gather_placements(r->fp.as<mmgen_feature_placement>(), r->cfp.as<mmgen_cave_feature_placement>(), r->counts.as<int>(), tgt, nr,
                                         r->pnx, r->pnz, r->gfp.as<mmgen_feature_placement>(), r->gcfp.as<mmgen_cave_feature_placement>(),
                                         r->bounds.as<int>(), posP, s, r->devMax.as<int>() + 1, r->hostMaxDev, r->devMax.as<int>(),
                                         (unsigned*)r->applyWork.p, (int)(mmk::apply_work_bytes() * r->kMaxSlices / 4), hfP, d_heightfields));
        d_heightfields = nullptr;                       // (copied by the gather)
    }
    if (par) { CK(hipEventRecord(r->evGather, s)); CK(hipStreamWaitEvent(sA, r->evGather, 0)); }
    if (par && features && r->nCompute < r->np) {
        // the next step's list lengths (see countsSpare): same size as this step's, cleared here - behind the event the rasterisers wait
        // for, so beside them, not in front of them
        const size_t countBytes = sizeof(int) * 2 * (size_t)r->np;
        r->countsSpareClean = 0;
        CK(r->countsSpare.ensure(countBytes));
        CK(hipMemsetAsync(r->countsSpare.p, 0, countBytes, s));
        r->countsSpareClean = countBytes;
    }
    // rasterisers + decorators slice by slice behind that slice's base fill
    for (int i = 0; i < r->nSlices; ++i) {
        const int c0 = r->sliceRow[i] * r->nx, n = (r->sliceRow[i + 1] - r->sliceRow[i]) * r->nx;
        uint8_t* blk = d_blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * c0;
        if (par) CK(hipStreamWaitEvent(sA, r->evFill[i], 0));
        if (features) {
            mmk::StageRange sr("mmgen:features");
            CK(mmk::launch_apply_features(blk, posP, n, r->gfp.as<mmgen_feature_placement>() + (size_t)MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK * c0,
                                          r->gcfp.as<mmgen_cave_feature_placement>() + (size_t)MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK * c0,
                                          r->bounds.as<int>() + 4 * c0, tgt + c0, (unsigned*)((char*)r->applyWork.p + mmk::apply_work_bytes() * i), sA, true));
        }
        if (decor) {
            mmk::StageRange sr("mmgen:decorators");
            CK(mmk::launch_decorators(blk, hfP, bwP, r->caveP.as<mmgen_cave_layer>(), posP, n, tgt + c0, sA));
        }
    }
    if (par) { CK(hipEventRecord(r->evTail, sA)); CK(hipStreamWaitEvent(s, r->evTail, 0)); }      // the caller's stream sees the finished blocks

    if (d_heightfields) MMK_LAUNCH(mmk::KID_SELECT, k_select, dim3(nr, 1), dim3(256), s, hfP, tgt, d_heightfields, 256);
    if (d_layers) MMK_LAUNCH(mmk::KID_SELECT, k_select, dim3(nr, MMGEN_LAYERS_SIZE / 256), dim3(256), s, layersP, tgt, d_layers, MMGEN_LAYERS_SIZE);
    if (d_cave_layers)
        MMK_LAUNCH(mmk::KID_SELECT, k_select, dim3(nr, (3 * MMGEN_CAVE_LAYERS_SIZE) / 256), dim3(256), s, (const float*)r->caveP.p, tgt, (float*)d_cave_layers,
                   3 * MMGEN_CAVE_LAYERS_SIZE);
    r->filled = false; r->filledInto = nullptr;
    return 0;
}

int mmgen_region_generate(mmgen_region* r, int cx0, int cz0, int nx, int nz, unsigned flags, uint8_t* d_blocks, float* d_heightfields, void* stream)
{
    // the ring's placement lists never leave this call: compute them lazily (mask value 2)
    const uint8_t* mask = nullptr;
    std::vector<uint8_t> lazyMask;
    if (r && (flags & MMGEN_REGION_FEATURES) && !(flags & MMGEN_REGION_EXACT_RING) && nx > 0 && nz > 0) {
        lazyMask.assign((size_t)(nx + 6) * (nz + 6), 2);
        mask = lazyMask.data();
    }
    flags &= ~MMGEN_REGION_EXACT_RING;
    if (r) r->earlyBlocks = d_blocks;          // nothing of the caller's can touch d_blocks between this call's begin and its fill
    CK(mmgen_region_begin(r, cx0, cz0, nx, nz, flags, mask, stream));
    return mmgen_region_finish(r, d_blocks, d_heightfields, nullptr, nullptr, stream);
}

int mmgen_region_last_erosion_passes(const mmgen_region* r)
{
    if (!r) return -1;
    if (r->passesPending && hipEventSynchronize(r->evPasses) != hipSuccess) return -1;
    // a relaxation of an earlier step that gave up: its zones were relaxed by the rescue pass (same planes); counted, not an error
    if (__atomic_load_n(r->hostMax + 1, __ATOMIC_RELAXED) != 0) { __atomic_store_n(r->hostMax + 1, 0, __ATOMIC_RELAXED); mmk::erosion_note_stall(); }
    return __atomic_load_n(r->hostMax + 2, __ATOMIC_RELAXED);
}

int mmgen_copy_placements(const mmgen_feature_placement* d_src_fp, const mmgen_cave_feature_placement* d_src_cfp, const int32_t* d_src_counts,
                          const int32_t* d_src_idx, mmgen_feature_placement* d_dst_fp, mmgen_cave_feature_placement* d_dst_cfp, int32_t* d_dst_counts,
                          const int32_t* d_dst_idx, int n, void* stream)
{
    if (n < 0 || (n > 0 && (!d_src_fp || !d_src_cfp || !d_src_counts || !d_src_idx || !d_dst_fp || !d_dst_cfp || !d_dst_counts || !d_dst_idx)))
        return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    MMK_LAUNCH(mmk::KID_COPY_PLACEMENTS, k_copy_placements, dim3(n), dim3(256), (hipStream_t)stream, d_src_fp, d_src_cfp, d_src_counts, d_src_idx,
               d_dst_fp, d_dst_cfp, d_dst_counts, d_dst_idx);
    return 0;
}

int mmgen_ring_header(const int32_t* d_counts, const int32_t* d_cells, int n, int32_t* d_header, void* stream)
{
    if (n < 0 || (n > 0 && (!d_counts || !d_cells || !d_header))) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    MMK_LAUNCH(mmk::KID_RING_PACK, k_ring_header, dim3((n + 255) / 256), dim3(256), (hipStream_t)stream, d_counts, d_cells, n, d_header);
    return 0;
}

int mmgen_ring_offsets(const int32_t* d_header, int n, int32_t* d_offsets, void* stream)
{
    if (n < 0 || !d_offsets || (n > 0 && !d_header)) return (int)hipErrorInvalidValue;
    MMK_LAUNCH(mmk::KID_RING_PACK, k_ring_offsets, dim3(1), dim3(1024), (hipStream_t)stream, d_header, n, d_offsets);
    return 0;
}

int mmgen_ring_pack(const mmgen_feature_placement* d_fp, const mmgen_cave_feature_placement* d_cfp, const int32_t* d_cells, const int32_t* d_header,
                    const int32_t* d_offsets, int n, int32_t* d_payload, void* stream)
{
    if (n < 0 || (n > 0 && (!d_fp || !d_cfp || !d_cells || !d_header || !d_offsets || !d_payload))) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    MMK_LAUNCH(mmk::KID_RING_PACK, k_ring_move<true>, dim3(n), dim3(256), (hipStream_t)stream, (mmgen_feature_placement*)d_fp,
               (mmgen_cave_feature_placement*)d_cfp, (int32_t*)nullptr, d_cells, d_header, d_offsets, d_payload);
    return 0;
}

int mmgen_ring_unpack(const int32_t* d_payload, const int32_t* d_header, const int32_t* d_offsets, const int32_t* d_cells, int n,
                      mmgen_feature_placement* d_fp, mmgen_cave_feature_placement* d_cfp, int32_t* d_counts, void* stream)
{
    if (n < 0 || (n > 0 && (!d_payload || !d_header || !d_offsets || !d_cells || !d_fp || !d_cfp || !d_counts))) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    MMK_LAUNCH(mmk::KID_RING_UNPACK, k_ring_move<false>, dim3(n), dim3(256), (hipStream_t)stream, d_fp, d_cfp, d_counts, d_cells, d_header, d_offsets,
               (int32_t*)d_payload);
    return 0;
}

int mmgen_ring_pack_messages(const mmgen_feature_placement* d_fp, const mmgen_cave_feature_placement* d_cfp, const int32_t* d_counts, const int32_t* d_cells,
                             const int32_t* d_slots, int n, int32_t* d_scratch, int32_t* d_messages, int32_t* d_overflow, void* stream)
{
    if (n < 0 || (n > 0 && (!d_fp || !d_cfp || !d_counts || !d_cells || !d_slots || !d_scratch || !d_messages || !d_overflow))) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int32_t* header = d_scratch;                 // [n][2]
    int32_t* offsets = d_scratch + 2 * (size_t)n; // [n + 1]
    MMK_LAUNCH(mmk::KID_RING_PACK, k_ring_msg_header, dim3((n + 255) / 256), dim3(256), s, d_counts, d_cells, (const int4*)d_slots, n, header, d_messages);
    MMK_LAUNCH(mmk::KID_RING_PACK, k_ring_offsets, dim3(1), dim3(1024), s, (const int32_t*)header, n, offsets);
    MMK_LAUNCH(mmk::KID_RING_PACK, k_ring_msg_move<true>, dim3(n), dim3(256), s, (mmgen_feature_placement*)d_fp, (mmgen_cave_feature_placement*)d_cfp,
               (int32_t*)nullptr, d_cells, (const int4*)d_slots, (const int32_t*)header, (const int32_t*)offsets, d_messages, (int*)d_overflow);
    return 0;
}

int mmgen_ring_unpack_messages(const int32_t* d_messages, const int32_t* d_cells, const int32_t* d_slots, int n, int32_t* d_scratch,
                               mmgen_feature_placement* d_fp, mmgen_cave_feature_placement* d_cfp, int32_t* d_counts, int32_t* d_overflow, void* stream)
{
    if (n < 0 || (n > 0 && (!d_messages || !d_cells || !d_slots || !d_scratch || !d_fp || !d_cfp || !d_counts || !d_overflow))) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int32_t* header = d_scratch;
    int32_t* offsets = d_scratch + 2 * (size_t)n;
    MMK_LAUNCH(mmk::KID_RING_UNPACK, k_ring_msg_read_header, dim3((n + 255) / 256), dim3(256), s, d_messages, (const int4*)d_slots, n, header);
    MMK_LAUNCH(mmk::KID_RING_UNPACK, k_ring_offsets, dim3(1), dim3(1024), s, (const int32_t*)header, n, offsets);
    MMK_LAUNCH(mmk::KID_RING_UNPACK, k_ring_msg_move<false>, dim3(n), dim3(256), s, d_fp, d_cfp, d_counts, d_cells, (const int4*)d_slots, (const int32_t*)header,
               (const int32_t*)offsets, (int32_t*)d_messages, (int*)d_overflow);
    return 0;
}

}  // extern "C"
