// mmgen erosion for gfx950: the relaxation ("slope method") of the 8 eroded layers over a 384x384-column zone grid.
// Behavioural spec: kernDoErosion chunk.cu:477-601 + the host loop of Chunk::erodeZone chunk.cu:682-705, copyLayers :603-656.
//
// Design (MI355X-first):
//  * every relaxation pass is a synchronous Jacobi step on a snapshot, so no workgroup ever reads a halo cell another workgroup is
//    rewriting (the reference updates in place across thread blocks; its result depends on block scheduling — DESIGN.md "Canonical
//    semantics");
//  * TEMPORAL BLOCKING: a pass is 19 us of launch latency for 3 us of L2-resident work, and a zone needs 27 - 39 of them.  One
//    launch therefore runs EROSION_K passes of the current layer on (32 + 2 K)^2 LDS tiles: after pass j the cells at distance > j
//    from the tile border are still exact, the 32 x 32 centre is exact after all K (the ring is recomputed by the neighbouring
//    tiles: 1.5 x the arithmetic, 1/K of the launches and of the global traffic).  Grid edges clamp like the reference (chunk.cu:545),
//    so towards an edge of the grid nothing is lost;
//  * convergence is a property of the whole zone, known only after the launch.  Passes after the first unchanged one are the
//    identity EXCEPT after an unchanged FIRST pass of a layer (it lifts by the accumulated height, later passes do not): the launch
//    that starts a layer also stores the state after its first pass, and the next launch picks that plane if bit 0 of the zone's
//    "changed" mask is clear.  The number of passes the reference's host loop would have run is recovered from the mask;
//  * the loop lives on the device: every launch derives its phase {layer, isFirst, done, planes} from the previous launch's phase and
//    "changed" mask with plain loads (every workgroup redundantly, a few scalar ops); one no-return atomicOr per workgroup and launch
//    publishes the mask.  The host enqueues launches back to back and reads the states every few launches;
//  * many zones per launch (blockIdx.z = zone); the zone working set stays L2 / Infinity-Cache resident.
#include <hip/hip_runtime.h>
#include <atomic>
#include "mm_biome.cuh"
#include "mmgen_erosion.h"
#include "mmgen_prof.h"

namespace mm {

#define ZS MMGEN_EROSION_GRID_SIDE
#define ZN MMGEN_EROSION_GRID_NUM_COLS

// Per-zone workspace layout (floats): work[8 layers][3][ZN] start planes (0 / 1: ping-pong across launches, 2: state after the
// layer's first pass), acc[2][ZN].
#define ZONE_WORK_FLOATS ((size_t)(8 * 3 + 2) * ZN)

#define EROSION_EXT (32 + 2 * EROSION_K)
#define EROSION_CELLS_EXT (EROSION_EXT * EROSION_EXT)
#ifndef EROSION_STRIPS
#define EROSION_STRIPS 11                                  // row groups: one lane = one column of the extended tile x EROSION_ROWS rows (mmgen_erosion.h: K x row groups)
#endif
#define EROSION_ROWS (EROSION_EXT / EROSION_STRIPS)
#define EROSION_THREADS (EROSION_EXT * EROSION_STRIPS)
static_assert(EROSION_EXT % EROSION_STRIPS == 0, "strips must tile the extended tile");

// phase of launch t from the phase and the changed mask of launch t - 1 (bit j = pass j of that launch altered some column)
MM_DEV ErosionPhase next_phase(const ErosionPhase& prev, unsigned maskPrev)
{
    ErosionPhase cur = prev;
    if (prev.fresh) { cur.fresh = 0; return cur; }
    if (prev.done) return cur;
    const int L = prev.layer;
    const int outSel = prev.isFirst ? 0 : 1 - prev.plane(L);           // plane the previous launch wrote its final state to
    const unsigned full = (1u << EROSION_K) - 1u;
    if ((maskPrev & full) == full) {                                 // every pass changed something: not converged yet
        cur.passes = prev.passes + EROSION_K;
        cur.setPlane(L, outSel); cur.accSel = 1 - prev.accSel; cur.isFirst = 0;
        return cur;
    }
    const int firstUnchanged = __builtin_ctz(~maskPrev);             // the pass at which the reference's loop stops
    cur.passes = prev.passes + firstUnchanged + 1;
    if (prev.isFirst && firstUnchanged == 0) cur.setPlane(L, 2);         // unchanged first pass: its own output is final, acc untouched
    else { cur.setPlane(L, outSel); cur.accSel = 1 - prev.accSel; }
    if (L == 0) cur.done = 1;
    else { cur.layer = L - 1; cur.isFirst = 1; }
    return cur;
}

// Planes, masks and phases travel between the workgroups of a zone INSIDE one launch.  They are read and written with device-scope
// relaxed atomics (global_load / global_store with sc1: served at the level every XCD sees) and ordered by the zone's barrier - no
// cache write-back / invalidate fences, which every workgroup of the chip would pay for at every barrier of every other zone.
MM_DEV float ld_dev(const float* p) { return __int_as_float(__hip_atomic_load((const int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
MM_DEV void st_dev(float* p, float v) { __hip_atomic_store((int*)p, __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One Jacobi pass of one lane: column ex of the extended tile, rows [r0, r1].  sIn / tIn = start plane and thickness (end - start)
// of the previous state, read through a sliding 3 x 3 register window (6 LDS reads per cell); sOut / tOut receive the new state.
// LIFT = first pass of a layer: sIn / tIn hold the values RAISED by the accumulated height of the layers above (built by the caller),
// the cell's own un-raised start is in sOut (it stays if the reference would not write, chunk.cu:578) and its end is raised too.
// Returns: bit 0 = some cell changed, bit 1 = some cell of the tile's own 32 x 32 centre changed.
template <bool LIFT>
MM_DEV int relax_strip(const float* __restrict__ sIn, const float* __restrict__ tIn, float* __restrict__ sOut, float* __restrict__ tOut,
                       const float* __restrict__ s_end, float* __restrict__ s_acc, float k1, float k2, int ex, int exL, int exR, int r0, int r1, int ezMin,
                       int ezMax, bool ownCol, float* __restrict__ startFirst /*grid pointer of this column, row 0 of the tile*/)
{
    int flags = 0;
    if (r0 > r1) return 0;
    // window rows: a = row above, b = this row, c = row below (clamped at the grid edges: chunk.cu:545)
    int rowA = EROSION_EXT * imax(r0 - 1, ezMin), rowB = EROSION_EXT * r0;
    float aS0 = sIn[rowA + exL], aS1 = sIn[rowA + ex], aS2 = sIn[rowA + exR];
    float aT0 = tIn[rowA + exL], aT1 = tIn[rowA + ex], aT2 = tIn[rowA + exR];
    float bS0 = sIn[rowB + exL], bS1 = sIn[rowB + ex], bS2 = sIn[rowB + exR];
    float bT0 = tIn[rowB + exL], bT1 = tIn[rowB + ex], bT2 = tIn[rowB + exR];
#pragma unroll 2
    for (int ez = r0; ez <= r1; ++ez) {
        const int rowC = EROSION_EXT * imin(ez + 1, ezMax);
        const float cS0 = sIn[rowC + exL], cS1 = sIn[rowC + ex], cS2 = sIn[rowC + exR];
        const float cT0 = tIn[rowC + exL], cT1 = tIn[rowC + ex], cT2 = tIn[rowC + exR];
        const int c = EROSION_EXT * ez + ex;
        const float thisStart = bS1;
        float raw = thisStart, thisEnd = s_end[c];
        if (LIFT) { raw = sOut[c]; thisEnd = thisEnd + s_acc[c]; }
        // neighbour order of dev_dirVecs2d (N, NE, E, SE, S, SW, W, NW; +z = "north" = row below in this layout): max is order-free
        float newStart = thisStart;
        newStart = gmax(newStart, cS1 - k1); newStart = gmax(newStart, cS2 - k2); newStart = gmax(newStart, bS2 - k1); newStart = gmax(newStart, aS2 - k2);
        newStart = gmax(newStart, aS1 - k1); newStart = gmax(newStart, aS0 - k2); newStart = gmax(newStart, bS0 - k1); newStart = gmax(newStart, cS0 - k2);
        float maxThickness = thisEnd - thisStart;
        maxThickness = gmax(maxThickness, cT1); maxThickness = gmax(maxThickness, cT2); maxThickness = gmax(maxThickness, bT2); maxThickness = gmax(maxThickness, aT2);
        maxThickness = gmax(maxThickness, aT1); maxThickness = gmax(maxThickness, aT0); maxThickness = gmax(maxThickness, bT0); maxThickness = gmax(maxThickness, cT0);
        newStart = gmin(newStart, thisEnd);
        float outStart = raw;
        if (maxThickness > 0.f) {
            outStart = newStart;
            if (newStart != thisStart) {
                s_acc[c] = s_acc[c] + (newStart - thisStart);
                flags |= (ownCol && ez >= EROSION_K && ez < EROSION_K + 32) ? 3 : 1;
            }
        }
        sOut[c] = outStart;
        tOut[c] = s_end[c] - outStart;
        if (LIFT && ownCol && ez >= EROSION_K && ez < EROSION_K + 32) st_dev(startFirst + (size_t)ZS * ez, outStart);
        aS0 = bS0; aS1 = bS1; aS2 = bS2; aT0 = bT0; aT1 = bT1; aT2 = bT2;
        bS0 = cS0; bS1 = cS1; bS2 = cS2; bT0 = cT0; bT1 = cT1; bT2 = cT2;
    }
    return flags;
}

// a phase another workgroup stored before the zone's barrier (word-wise device-scope loads: never from a stale scalar / vector cache line)
MM_DEV ErosionPhase load_phase(const ErosionPhase* p)
{
    static_assert(sizeof(ErosionPhase) == 7 * sizeof(int), "seven words");
    ErosionPhase r;
    const int* src = (const int*)p;
    int* dst = (int*)&r;
#pragma unroll
    for (int i = 0; i < 7; ++i) dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return r;
}
MM_DEV void store_phase(ErosionPhase* p, const ErosionPhase& v)
{
    const int* src = (const int*)&v;
    int* dst = (int*)p;
#pragma unroll
    for (int i = 0; i < 7; ++i) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One launch-equivalent ("round") of one 32 x 32 tile: EROSION_K Jacobi passes of the phase's layer on the (32 + 2 K)^2 extended tile.
// Called by every thread of the workgroup; the LDS planes are the caller's.  Returns nothing; the tile's "changed" bits are ORed into
// *zoneMask (device scope) by one thread.
// The raw planes of a zone (its 8 eroded layers' starts + the heightfield), read-only for the whole relaxation: either packed by
// k_erosion_gather (the per-stage ABI: Chunk::erodeZone's gathered buffer) or, in the region path, straight from the chunk-major
// layers / heightfields through the zone's 24 x 24 chunk list (copyLayers' index math, chunk.cu:603-656, without the copy).
struct RawPlanes {
    const float* gathered;            // [9][ZN] or null
    const float* layers;              // chunk-major raw layers [chunk][20][256]
    const float* hf;                  // chunk-major heightfields [chunk][256]
    const int* chunkIdx;              // the zone's [24 * 24] chunks
    MM_DEV float at(int plane, int gx, int gz) const
    {
        if (gathered) return gathered[(size_t)plane * ZN + gx + ZS * gz];
        const int chunk = chunkIdx[(gz >> 4) * 24 + (gx >> 4)], idx2d = (gz & 15) * 16 + (gx & 15);
        return plane == 8 ? hf[(size_t)256 * chunk + idx2d] : layers[(size_t)MMGEN_LAYERS_SIZE * chunk + 256 * (MMGEN_NUM_STRATIFIED_MATERIALS + plane) + idx2d];
    }
};

MM_DEV void erode_tile(const RawPlanes& raw, float* work, const ErosionPhase& ph, int tileX, int tileZ, unsigned* zoneMask,
                       float (*s_s)[EROSION_CELLS_EXT], float (*s_t)[EROSION_CELLS_EXT], float* s_end, float* s_acc, unsigned* s_mask)
{
    const int tid = threadIdx.x;
    const int layer = ph.layer;
    const bool isFirst = ph.isFirst != 0;
    const float* accIn = work + (size_t)24 * ZN + (size_t)ph.accSel * ZN;
    float* accOut = work + (size_t)24 * ZN + (size_t)(1 - ph.accSel) * ZN;
    const float* startIn = work + ((size_t)layer * 3 + ph.plane(layer)) * ZN;                     // (a layer's first round reads the raw plane instead)
    float* startOut = work + ((size_t)layer * 3 + (isFirst ? 0 : 1 - ph.plane(layer))) * ZN;
    float* startFirst = work + ((size_t)layer * 3 + 2) * ZN;
    // end plane = final start plane of the layer above (already eroded), or the heightfield plane for the top layer
    const bool rawEnd = layer == MMGEN_NUM_ERODED_MATERIALS - 1;
    const float* endIn = work + ((size_t)(rawEnd ? layer : layer + 1) * 3 + ph.plane(rawEnd ? layer : layer + 1)) * ZN;

    // extended tile: ex, ez in [0, EXT) <-> grid (gx0 + ex, gz0 + ez); cells beyond the grid do not exist (neighbours clamp to the edge)
    const int gx0 = tileX * 32 - EROSION_K, gz0 = tileZ * 32 - EROSION_K;
    const int exMin = imax(0, -gx0), exMax = imin(EROSION_EXT - 1, ZS - 1 - gx0);
    const int ezMin = imax(0, -gz0), ezMax = imin(EROSION_EXT - 1, ZS - 1 - gz0);
    const int ex = tid % EROSION_EXT, strip = tid / EROSION_EXT;
    const int rowLo = strip * EROSION_ROWS, rowHi = rowLo + EROSION_ROWS - 1;
    const bool colExists = ex >= exMin && ex <= exMax;
    const int exL = imax(ex - 1, exMin), exR = imin(ex + 1, exMax);
    const bool ownCol = ex >= EROSION_K && ex < EROSION_K + 32;
    if (tid == 0) *s_mask = 0u;

    // load: a first launch stages the un-raised start in plane 0 and the RAISED start / thickness in plane 1 (the input of pass 0)
    if (colExists) {
        for (int ez = imax(rowLo, ezMin); ez <= imin(rowHi, ezMax); ++ez) {
            const int c = EROSION_EXT * ez + ex, g = (gx0 + ex) + ZS * (gz0 + ez);
            const float sv = isFirst ? raw.at(layer, gx0 + ex, gz0 + ez) : ld_dev(startIn + g);
            const float ev = rawEnd ? raw.at(8, gx0 + ex, gz0 + ez) : ld_dev(endIn + g);
            const float av = ld_dev(accIn + g);
            s_s[0][c] = sv; s_end[c] = ev; s_acc[c] = av;
            if (isFirst) { const float ls = sv + av; s_s[1][c] = ls; s_t[1][c] = (ev + av) - ls; }
            else s_t[0][c] = ev - sv;
        }
    }
    __syncthreads();

    const float k1 = kMaterialAmpOrTan[MMGEN_NUM_STRATIFIED_MATERIALS + layer];
    const float k2 = k1 * MM_SQRT_2;
    float* colFirst = startFirst + (gx0 + ex) + (size_t)ZS * gz0;
    int cur = 0;                                             // plane holding the current state (after a first pass: 0 again)
    unsigned myMask = 0u;
#pragma unroll 1
    for (int j = 0; j < EROSION_K; ++j) {
        // cells still exact after this pass: at distance > j from every tile border that is not an edge of the grid
        const int xl = gx0 < 0 ? exMin : j + 1, xh = gx0 + EROSION_EXT > ZS ? exMax : EROSION_EXT - 2 - j;
        const int zl = gz0 < 0 ? ezMin : j + 1, zh = gz0 + EROSION_EXT > ZS ? ezMax : EROSION_EXT - 2 - j;
        const bool colLive = ex >= xl && ex <= xh;
        const int r0 = colLive ? imax(rowLo, zl) : 1, r1 = colLive ? imin(rowHi, zh) : 0;
        int flags;
        if (isFirst && j == 0) {
            flags = relax_strip<true>(s_s[1], s_t[1], s_s[0], s_t[0], s_end, s_acc, k1, k2, ex, exL, exR, r0, r1, ezMin, ezMax, ownCol, colFirst);
            // the result is in plane 0 again
        } else {
            flags = relax_strip<false>(s_s[cur], s_t[cur], s_s[1 - cur], s_t[1 - cur], s_end, s_acc, k1, k2, ex, exL, exR, r0, r1, ezMin, ezMax, ownCol,
                                       colFirst);
            cur = 1 - cur;
        }
        if (flags & 2) myMask |= 1u << j;
        // a pass (other than a first pass) that changes no live cell of the tile is the identity from here on: stop
        const int any = __syncthreads_or(flags & 1);
        if (!any && !(isFirst && j == 0)) break;
    }
    if (myMask) atomicOr(s_mask, myMask);
    // results of the centre
    if (ownCol) {
        for (int ez = imax(rowLo, EROSION_K); ez <= imin(rowHi, EROSION_K + 31); ++ez) {
            const int c = EROSION_EXT * ez + ex, g = (gx0 + ex) + ZS * (gz0 + ez);
            st_dev(startOut + g, s_s[cur][c]);
            st_dev(accOut + g, s_acc[c]);
        }
    }
    __syncthreads();
    if (tid == 0 && *s_mask) atomicOr(zoneMask, *s_mask);
}

// The whole relaxation of a batch of zones in ONE launch (the host loop of Chunk::erodeZone chunk.cu:682-705 on the device).  A zone is
// worked on by `perZone` persistent workgroups; a round = what one launch of the round-3 kernel did (EROSION_K passes of the zone's current
// layer on each of its 144 tiles, the workgroup's share of them one after the other), then a barrier among the zone's workgroups
// (release: fence + counter; acquire: spin + fence), then every workgroup derives the next phase from the zone's "changed" mask exactly
// like the launches did.  Zones do not wait for each other and the host is not involved: no launch gaps, no state read-backs.
// Workgroups take their (zone, member) from a ticket, zone-major: whatever order the dispatcher places workgroups in, the zones with the
// lowest tickets are complete and make progress, so the barrier cannot deadlock even when the launch does not fit the chip at once.
__global__ void __launch_bounds__(EROSION_THREADS)
k_erode_zones(const float* __restrict__ gatheredBase, size_t gatheredStride, const float* __restrict__ rawLayers, const float* __restrict__ rawHf,
              const int* __restrict__ zoneChunkIdx /*[zones][576]*/, float* workBase, ErosionState* states, unsigned* ticket, int perZone, int* maxPasses,
              int* maxPassesAlso)
{
    __shared__ float s_s[2][EROSION_CELLS_EXT];            // start planes, ping-pong over the passes
    __shared__ float s_t[2][EROSION_CELLS_EXT];            // thickness = end - start of the same states (what the neighbours compare)
    __shared__ float s_end[EROSION_CELLS_EXT];
    __shared__ float s_acc[EROSION_CELLS_EXT];             // accumulated heights; after the load every cell is touched by its own lane only
    __shared__ unsigned s_mask;
    __shared__ unsigned s_ticket, s_tile;
    __shared__ ErosionPhase s_ph;

    const int tid = threadIdx.x;
    if (tid == 0) s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    const int zone = (int)(s_ticket / (unsigned)perZone), member = (int)(s_ticket % (unsigned)perZone);
    ErosionState* st = states + zone;
    const RawPlanes raw = {gatheredBase ? gatheredBase + gatheredStride * zone : nullptr, rawLayers, rawHf, zoneChunkIdx ? zoneChunkIdx + 576 * zone : nullptr};
    float* work = workBase + ZONE_WORK_FLOATS * zone;

#pragma unroll 1
    for (int t = 0;; ++t) {
        if (tid == 0) {
            const ErosionPhase prev = load_phase(&st->slot[(t + 1) & 1]);
            const unsigned maskPrev = __hip_atomic_load(&st->changed[(t + 3) & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const ErosionPhase ph = next_phase(prev, maskPrev);
            s_ph = ph;
            if (member == 0) {
                store_phase(&st->slot[t & 1], ph);
                if (ph.done) {                                       // both slots final: the finish kernels read slot[0]
                    store_phase(&st->slot[(t + 1) & 1], ph);
                    atomicMax(maxPasses, ph.passes);
                    if (maxPassesAlso) atomicMax(maxPassesAlso, ph.passes);
                }
                __hip_atomic_store(&st->changed[(t + 1) & 3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&st->tileTicket[(t + 1) & 1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_tile = ph.done ? 144u : __hip_atomic_fetch_add(&st->tileTicket[t & 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const ErosionPhase ph = s_ph;
        if (ph.done) break;
        // the zone's 144 tiles are dealt out on demand: a tile costs anything between one pass (nothing moves any more) and EROSION_K
        unsigned tile = s_tile;
        while (tile < 144u) {
            __syncthreads();                                          // everyone has read s_tile
            if (tid == 0) s_tile = __hip_atomic_fetch_add(&st->tileTicket[t & 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // the next draw is in flight while this tile is worked on
            erode_tile(raw, work, ph, (int)(tile % 12u), (int)(tile / 12u), &st->changed[t & 3], s_s, s_t, s_end, s_acc, &s_mask);
            tile = s_tile;                                            // (erode_tile ends with a workgroup barrier)
        }
        // ---- barrier among the zone's workgroups: every store above is a device-scope store this wave has waited for (__syncthreads)
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(&st->barrier, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)perZone * (unsigned)(t + 1);
            while (__hip_atomic_load(&st->barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(4);
        }
        __syncthreads();
    }
}

// final planes back into the caller's gathered-layers buffer (in-place contract of Chunk::erodeZone) and the accumulated heights
__global__ void __launch_bounds__(256)
k_erode_writeback(float* __restrict__ gatheredBase, size_t gatheredStride, const float* __restrict__ workBase, const ErosionState* __restrict__ states,
                  float* __restrict__ accOutBase, size_t accStride, int lastT)
{
    const int zone = blockIdx.z;
    const ErosionPhase* st = &states[zone].slot[lastT & 1];      // the phase the last launch ran with: done, all planes final
    const int c = blockIdx.x * 256 + threadIdx.x;
    const float* work = workBase + ZONE_WORK_FLOATS * zone;
    float* gathered = gatheredBase + gatheredStride * zone;
#pragma unroll
    for (int l = 0; l < 8; ++l) gathered[(size_t)l * ZN + c] = work[((size_t)l * 3 + st->plane(l)) * ZN + c];
    if (accOutBase) accOutBase[accStride * zone + c] = work[(size_t)24 * ZN + (size_t)st->accSel * ZN + c];
}

__global__ void k_erode_init(ErosionState* states, float* workBase, int zones, unsigned* ticket)
{
    const int zone = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (zone >= zones) return;
    // zero both accumulator buffers (thrust::fill_n of chunk.cu:679-680)
    float* acc = workBase + ZONE_WORK_FLOATS * zone + (size_t)24 * ZN;
    if (i < 2 * ZN) acc[i] = 0.f;
    if (i == 0) {
        ErosionPhase s;
        s.layer = MMGEN_NUM_ERODED_MATERIALS - 1; s.isFirst = 1; s.done = 0; s.passes = 0; s.accSel = 0; s.fresh = 1;
        s.sel = 0u;
        states[zone].slot[1] = s;            // launch 0 reads slot[(0 - 1) & 1]
        states[zone].slot[0] = s;
        for (int k = 0; k < 4; ++k) states[zone].changed[k] = 0u;
        states[zone].barrier = 0u; states[zone].tileTicket[0] = 0u; states[zone].tileTicket[1] = 0u;
        if (zone == 0) { ticket[0] = 0u; ticket[1] = 0u; }
    }
}

// E1: chunk-major raw layers of a chunk grid -> packed zone planes (copyLayers(to) chunk.cu:603-656).
// grid: (24*24 chunks, 9 planes, zones); block 256 = the chunk's columns.
__global__ void __launch_bounds__(256)
k_erosion_gather(const float* __restrict__ layers, const float* __restrict__ hf, const int* __restrict__ zoneChunkIdx /*[zones][576]*/,
                 float* __restrict__ gatheredBase, size_t gatheredStride)
{
    const int zone = blockIdx.z, plane = blockIdx.y, cc = blockIdx.x;
    const int chunk = zoneChunkIdx[zone * 576 + cc];
    const int t = threadIdx.x;
    const int cx = cc % 24, cz = cc / 24;
    const float v = (plane == 8) ? hf[(size_t)256 * chunk + t] : layers[(size_t)MMGEN_LAYERS_SIZE * chunk + 256 * (MMGEN_NUM_STRATIFIED_MATERIALS + plane) + t];
    const int gx = cx * 16 + (t & 15), gz = cz * 16 + (t >> 4);
    gatheredBase[gatheredStride * zone + (size_t)plane * ZN + gx + ZS * gz] = v;
}

// E3: centre 12x12 chunks, 8 eroded planes -> chunk-major layers of the destination buffer (copyLayers(from)).
__global__ void __launch_bounds__(256)
k_erosion_scatter(const float* __restrict__ gatheredBase, size_t gatheredStride, const int* __restrict__ zoneChunkIdxOut /*[zones][144], -1 = skip*/,
                  float* __restrict__ layersOut)
{
    const int zone = blockIdx.z, plane = blockIdx.y, cc = blockIdx.x;
    const int chunk = zoneChunkIdxOut[zone * 144 + cc];
    if (chunk < 0) return;
    const int t = threadIdx.x;
    const int cx = cc % 12 + 6, cz = cc / 12 + 6;
    const int gx = cx * 16 + (t & 15), gz = cz * 16 + (t >> 4);
    layersOut[(size_t)MMGEN_LAYERS_SIZE * chunk + 256 * (MMGEN_NUM_STRATIFIED_MATERIALS + plane) + t] =
        gatheredBase[gatheredStride * zone + (size_t)plane * ZN + gx + ZS * gz];
}

// Region path: the centre 12 x 12 chunks' eroded planes straight from the zones' work planes into the chunk-major layers (what
// k_erode_writeback + k_erosion_scatter do through the gathered buffer, for the quarter of each zone that is kept).
__global__ void __launch_bounds__(256)
k_erode_finish(const float* __restrict__ workBase, const ErosionState* __restrict__ states, int lastT, const int* __restrict__ zoneChunkIdxOut /*[zones][144], -1 = skip*/,
               float* __restrict__ layersOut)
{
    const int zone = blockIdx.z, plane = blockIdx.y, cc = blockIdx.x;
    const int chunk = zoneChunkIdxOut[zone * 144 + cc];
    if (chunk < 0) return;
    const ErosionPhase* st = &states[zone].slot[lastT & 1];      // the phase the last launch ran with: done, all planes final
    const int t = threadIdx.x;
    const int cx = cc % 12 + 6, cz = cc / 12 + 6;
    const int gx = cx * 16 + (t & 15), gz = cz * 16 + (t >> 4);
    layersOut[(size_t)MMGEN_LAYERS_SIZE * chunk + 256 * (MMGEN_NUM_STRATIFIED_MATERIALS + plane) + t] =
        workBase[ZONE_WORK_FLOATS * zone + ((size_t)plane * 3 + st->plane(plane)) * ZN + gx + ZS * gz];
}

}  // namespace mm

namespace mmk {

size_t erosion_work_bytes(int zones) { return (size_t)zones * ZONE_WORK_FLOATS * sizeof(float); }
// the zones' states, then one more record's worth of words: [0] = the launch's ticket counter, [1] = largest pass count of the zones
size_t erosion_state_bytes(int zones) { return (size_t)(zones + 1) * sizeof(mm::ErosionState); }

// workgroups of k_erode_zones the chip holds at once (LDS-bound: three per CU on gfx950), or `quarterCuCap` / 4 per CU if that is fewer
static int erosion_resident_workgroups(int quarterCuCap)
{
    static std::atomic<int> perCuCached{0};                 // (the kernel's occupancy is the same on every gfx950)
    int perCu = perCuCached.load(std::memory_order_relaxed);
    if (!perCu) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, mm::k_erode_zones, EROSION_THREADS, 0) != hipSuccess || perCu < 1) perCu = 1;
        perCuCached.store(perCu, std::memory_order_relaxed);
    }
    int cus = device_cus();
    if (cus < 1) cus = 1;
    const int fit = perCu * cus;
    if (quarterCuCap <= 0) return fit;
    const int cap = (int)((long long)cus * quarterCuCap / 4);
    return cap < fit ? (cap > 0 ? cap : 1) : fit;
}

// Enqueues the relaxation of `zones` packed zone buffers (stride in floats) to convergence: ONE persistent launch, then the kernel that
// moves the final planes out.  Nothing is read back unless the caller asks for the pass count (maxPasses != null: the stream is
// synchronised, like the reference's erodeZone); maxPassesDev (device, may be null) is raised to the largest pass count with the stream.
int erode_zones(float* gathered, size_t strideFloats, int zones, float* work, mm::ErosionState* states, float* accOut, size_t accStride,
                hipStream_t s, int* maxPasses, const int* zoneChunkIdxOut, float* layersOut, int* maxPassesDev, hipEvent_t beforeRelaxation,
                const float* rawLayers, const float* rawHf, const int* zoneChunkIdx, int workgroupsPer4Cu, const unsigned** startedCounter,
                unsigned* startedTarget)
{
    if (!gathered && !(rawLayers && rawHf && zoneChunkIdx && layersOut)) return (int)hipErrorInvalidValue;
    if (zones <= 0) return 0;
    unsigned* ticket = (unsigned*)(states + zones);
    int* passesWord = (int*)(ticket + 1);
    MMK_LAUNCH(KID_ERODE_INIT, mm::k_erode_init, dim3((2 * ZN + 255) / 256, zones), dim3(256), s, states, work, zones, ticket);
    // as many workgroups per zone as keep the whole launch resident (a zone's 144 tiles are dealt out to them round by round)
    int perZone = erosion_resident_workgroups(workgroupsPer4Cu) / zones;
    perZone = perZone < 1 ? 1 : (perZone > 144 ? 144 : perZone);
    perZone = (144 + (144 + perZone - 1) / perZone - 1) / ((144 + perZone - 1) / perZone);      // fewest workgroups with the same tiles per round
    if (beforeRelaxation) { hipError_t e = hipEventRecord(beforeRelaxation, s); if (e != hipSuccess) return (int)e; }
    // the launch's ticket counts the workgroups that have STARTED (k_erode_init has just cleared it): a caller that wants them on the chip
    // before it launches something that takes every slot waits for the counter to reach the grid size (launch_caves)
    if (startedCounter) *startedCounter = ticket;
    if (startedTarget) *startedTarget = (unsigned)(zones * perZone);
    MMK_LAUNCH(KID_ERODE_PASS, mm::k_erode_zones, dim3(zones * perZone), dim3(EROSION_THREADS), s, (const float*)gathered, strideFloats, rawLayers, rawHf,
               zoneChunkIdx, work, states, ticket, perZone, passesWord, maxPassesDev);
    if (layersOut) {
        // region path: no in-place contract to honour, the kept chunks' planes go straight to the layers
        MMK_LAUNCH(KID_EROSION_SCATTER, mm::k_erode_finish, dim3(144, 8, zones), dim3(256), s, (const float*)work, (const mm::ErosionState*)states, 0,
                   zoneChunkIdxOut, layersOut);
    } else {
        MMK_LAUNCH(KID_ERODE_WRITEBACK, mm::k_erode_writeback, dim3(ZN / 256, 1, zones), dim3(256), s, gathered, strideFloats, work, states, accOut,
                   accStride, 0);
    }
    if (maxPasses) {
        hipError_t e = hipMemcpyAsync(maxPasses, passesWord, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return (int)e;
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

int erosion_gather(const float* layers, const float* hf, const int* zoneChunkIdx, int zones, float* gathered, size_t strideFloats, hipStream_t s)
{
    if (zones <= 0) return 0;
    MMK_LAUNCH(KID_EROSION_GATHER, mm::k_erosion_gather, dim3(576, 9, zones), dim3(256), s, layers, hf, zoneChunkIdx, gathered, strideFloats);
    return 0;
}

int erosion_scatter(const float* gathered, size_t strideFloats, const int* zoneChunkIdxOut, int zones, float* layersOut, hipStream_t s)
{
    if (zones <= 0) return 0;
    MMK_LAUNCH(KID_EROSION_SCATTER, mm::k_erosion_scatter, dim3(144, 8, zones), dim3(256), s, gathered, strideFloats, zoneChunkIdxOut, layersOut);
    return 0;
}

}  // namespace mmk
