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
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>
#include "../../include/mmgen.h"
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
#define EROSION_THREADS 512                                 // 8 waves: 484 lanes relax (lane = column x row group), all 512 move 16-byte pieces
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
// accesses (sc1: loads served below the per-CU L1, stores written through) and ordered by the zone's barrier - no cache write-back /
// invalidate fences, which every workgroup of the chip would pay for at every barrier of every other zone.  The hand-off form is the
// one of the MI355X guide (inter-workgroup visibility, "every store sc1, every load sc1"): EVERY storing wave drains its stores
// (s_waitcnt vmcnt(0)) before the workgroup barrier behind which one lane arrives at the zone's counter; the consumer polls that
// counter and every load of a handed-off byte is an sc1 load to registers.  Planes move as 16-byte pieces (one fabric transaction per
// lane instead of four).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct F4 { float x, y, z, w; };
#define MM_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define MM_AUX_SC1 16
#define MM_BUF_OOB 0xFFFFFFF0u            /* beyond every descriptor's range: the load returns zeros and moves nothing */
#ifndef MM_ERODE_LOAD_AUX
#define MM_ERODE_LOAD_AUX MM_AUX_SC1
#endif
#ifndef MM_ERODE_ROUND_INV
#define MM_ERODE_ROUND_INV 0
#endif
MM_DEV F4 ld4_dev(__amdgpu_buffer_rsrc_t rs, unsigned byteOff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byteOff, 0, MM_ERODE_LOAD_AUX);
    return F4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
// A zone is relaxed by workgroups of ONE XCD (k_erode_zones reads HW_REG_XCC_ID and forms its groups per XCD): its planes travel through
// that XCD's L2, which every CU of the XCD shares.  The stores are therefore plain (the line stays in the L2: the guide's "plain / sc0 / nt
// KEEP the line in the XCD's L2, sc1 DROPs it"), the loads keep sc1 (past the per-CU L1, served by the L2).  MM_ERODE_STORE_AUX=16 restores
// the write-through stores (then any placement of a zone's workgroups is coherent, and every round's planes cross the fabric).
#ifndef MM_ERODE_STORE_AUX
#define MM_ERODE_STORE_AUX 0
#endif
template <int AUX = MM_ERODE_STORE_AUX>
MM_DEV void st4_dev(__amdgpu_buffer_rsrc_t rs, unsigned byteOff, const F4& f)
{
    const u32x4 v = {__float_as_uint(f.x), __float_as_uint(f.y), __float_as_uint(f.z), __float_as_uint(f.w)};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)byteOff, 0, AUX);
}
#ifndef MM_ERODE_FIRST_AUX
#define MM_ERODE_FIRST_AUX MM_ERODE_STORE_AUX          /* the "state after the first pass" plane, read only by zones whose first pass changed nothing */
#endif

// Geometry of a tile's traffic.  Loads: the rows of the extended tile as aligned 16-byte pieces (columns [gx0 - PAD, gx0 - PAD + 4 SEGS) of
// the grid; the PAD + (4 SEGS - EXT - PAD) columns outside the extended tile are dropped), ROWS x SEGS pieces per plane, three planes
// (start, end, accumulated heights): 1 584 pieces dealt out flat to the 512 threads, at most EROSION_SLOTS per thread, held in registers
// while the previous tile is relaxed.  Stores: the 32 x 32 centre as 8 pieces per row, start plane by threads 0..255, accumulator by
// threads 256..511.
#define EROSION_PAD ((4 - EROSION_K % 4) % 4)
#define EROSION_SEGS ((EROSION_EXT + EROSION_PAD + 3) / 4)
#define EROSION_PLANE_PIECES (EROSION_EXT * EROSION_SEGS)
#define EROSION_SLOTS ((3 * EROSION_PLANE_PIECES + EROSION_THREADS - 1) / EROSION_THREADS)
static_assert(EROSION_K % 2 == 0, "8-byte LDS pieces: the extended tile starts at an even column");
static_assert(EROSION_THREADS == 512 && EROSION_THREADS >= EROSION_EXT * EROSION_STRIPS, "centre stores: 2 x 256 pieces; one lane per column and row group");

// One Jacobi pass of one lane: column ex of the extended tile, rows [r0, r1].  sIn / tIn = start plane and thickness (end - start)
// of the previous state, read through a sliding 3 x 3 register window (6 LDS reads per cell); sOut / tOut receive the new state.
// LIFT = first pass of a layer: sIn / tIn hold the values RAISED by the accumulated height of the layers above (built by the caller),
// the cell's own un-raised start is in sOut (it stays if the reference would not write, chunk.cu:578) and its end is raised too.
// Returns: bit 0 = some cell changed, bit 1 = some cell of the tile's own 32 x 32 centre changed.
template <bool LIFT>
MM_DEV int relax_strip(const float* __restrict__ sIn, const float* __restrict__ tIn, float* __restrict__ sOut, float* __restrict__ tOut,
                       const float* __restrict__ s_end, float* __restrict__ s_acc, float k1, float k2, int ex, int exL, int exR, int r0, int r1, int ezMin,
                       int ezMax, bool ownCol)
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

// The raw planes of a zone (its 8 eroded layers' starts + the heightfield), read-only for the whole relaxation: either packed by
// k_erosion_gather (the per-stage ABI: Chunk::erodeZone's gathered buffer) or, in the region path, straight from the chunk-major
// layers / heightfields through the zone's 24 x 24 chunk list (copyLayers' index math, chunk.cu:603-656, without the copy).
struct RawPlanes {
    const float* gathered;            // [9][ZN] or null
    const float* layers;              // chunk-major raw layers [chunk][20][256]
    const float* hf;                  // chunk-major heightfields [chunk][256]
    const int* chunkIdx;              // the zone's [24 * 24] chunks
    // four columns gx .. gx + 3 (gx a multiple of 4: inside one chunk row) of a plane; chunk = chunkIdx[(gz >> 4) * 24 + (gx >> 4)] (region path)
    MM_DEV F4 at4(int plane, int gx, int gz, int chunk) const
    {
        if (gathered) {          // the caller's buffers: a zone's stride is odd (the trailing flag word), so no 16-byte alignment to rely on
            const float* p = gathered + (size_t)plane * ZN + gx + ZS * gz;
            return F4{p[0], p[1], p[2], p[3]};
        }
        const int idx2d = (gz & 15) * 16 + (gx & 15);
        const float* p = plane == 8 ? hf + (size_t)256 * chunk + idx2d : layers + (size_t)MMGEN_LAYERS_SIZE * chunk + 256 * (MMGEN_NUM_STRATIFIED_MATERIALS + plane) + idx2d;
        const float4 v = *reinterpret_cast<const float4*>(p);
        return F4{v.x, v.y, v.z, v.w};
    }
};

// What a round of a zone reads and writes (workgroup-uniform; float offsets into the zone's work planes).
struct RoundPlanes {
    int layer;
    bool isFirst, rawEnd, topFirst;       // topFirst: the very first round of the zone - the accumulated heights are zero, not loaded
    unsigned startIn, endIn, accIn, startOut, accOut, startFirst;
};
MM_DEV RoundPlanes round_planes(const ErosionPhase& ph)
{
    RoundPlanes rp;
    const int layer = ph.layer;
    rp.layer = layer;
    rp.isFirst = ph.isFirst != 0;
    // end plane = final start plane of the layer above (already eroded), or the heightfield plane for the top layer
    rp.rawEnd = layer == MMGEN_NUM_ERODED_MATERIALS - 1;
    rp.topFirst = rp.rawEnd && rp.isFirst;
    rp.accIn = (unsigned)((24 + ph.accSel) * ZN);
    rp.accOut = (unsigned)((24 + 1 - ph.accSel) * ZN);
    rp.startIn = (unsigned)((layer * 3 + ph.plane(layer)) * ZN);                                // (a layer's first round reads the raw plane instead)
    rp.startOut = (unsigned)((layer * 3 + (rp.isFirst ? 0 : 1 - ph.plane(layer))) * ZN);
    rp.startFirst = (unsigned)((layer * 3 + 2) * ZN);
    const int le = rp.rawEnd ? layer : layer + 1;
    rp.endIn = (unsigned)((le * 3 + ph.plane(le)) * ZN);
    return rp;
}

// A thread's pieces of the next tile, in flight while the current tile is relaxed: slot k is piece threadIdx.x + 512 k of the flat list
// [plane][row][16-byte piece].
struct TilePieces { F4 v[EROSION_SLOTS]; };
// the thread index as a value the compiler cannot see through: the piece / lane geometry derived from it is a handful of integer
// operations, and hoisting all of it out of the round loop (it is loop invariant) costs more registers than the kernel has
MM_DEV int opaque_tid() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; }

MM_DEV void tile_issue_loads(TilePieces& r, const RawPlanes& raw, const int* s_chunk, __amdgpu_buffer_rsrc_t work, const RoundPlanes& rp, int tileX, int tileZ)
{
    const int tid0 = opaque_tid();
#pragma unroll
    for (int k = 0; k < EROSION_SLOTS; ++k) {
        const int p = tid0 + EROSION_THREADS * k;
        const int plane = p / EROSION_PLANE_PIECES, rem = p % EROSION_PLANE_PIECES;
        const int row = rem / EROSION_SEGS, seg = rem % EROSION_SEGS;
        const int gx = tileX * 32 - EROSION_K - EROSION_PAD + 4 * seg, gz = tileZ * 32 - EROSION_K + row;
        // (a piece lies inside the grid or outside it: 384 is a multiple of 4)
        const bool ok = p < 3 * EROSION_PLANE_PIECES && gx >= 0 && gx < ZS && gz >= 0 && gz < ZS;
        // (plane -> round's plane by masks: as a three-way select the compiler builds a table in scratch)
        const unsigned m0 = plane == 0 ? ~0u : 0u, m1 = plane == 1 ? ~0u : 0u, m2 = ~(m0 | m1);
        const bool useRaw = ok && ((m0 & (unsigned)rp.isFirst) | (m1 & (unsigned)rp.rawEnd)) != 0u;
        const bool fromWork = ok && !useRaw && (m2 & (unsigned)rp.topFirst) == 0u;
        const unsigned base = (rp.startIn & m0) | (rp.endIn & m1) | (rp.accIn & m2);
        r.v[k] = ld4_dev(work, fromWork ? 4u * (base + (unsigned)(gx + ZS * gz)) : MM_BUF_OOB);      // (zeros, and no traffic, where the piece is not a work plane's)
        if (useRaw) r.v[k] = raw.at4(plane == 1 ? 8 : rp.layer, gx, gz, raw.gathered ? 0 : s_chunk[(gz >> 4) * 24 + (gx >> 4)]);
    }
}

// the pieces into the LDS planes as loaded: start -> plane 0 of s_s, end -> s_end, accumulated heights -> s_acc.  Cells beyond the grid
// do not exist (neighbours clamp to the edge) and are never read.
MM_DEV void tile_commit(const TilePieces& r, int tileX, int tileZ, float* s_start, float* s_end, float* s_acc)
{
    const int tid0 = opaque_tid();
#pragma unroll
    for (int k = 0; k < EROSION_SLOTS; ++k) {
        const int p = tid0 + EROSION_THREADS * k;
        const int plane = p / EROSION_PLANE_PIECES, rem = p % EROSION_PLANE_PIECES;
        const int row = rem / EROSION_SEGS, seg = rem % EROSION_SEGS;
        const int ex0 = 4 * seg - EROSION_PAD;
        const int gx = tileX * 32 - EROSION_K + ex0, gz = tileZ * 32 - EROSION_K + row;
        if (p >= 3 * EROSION_PLANE_PIECES || !(gx >= 0 && gx < ZS && gz >= 0 && gz < ZS)) continue;
        float* dst = (plane == 0 ? s_start : (plane == 1 ? s_end : s_acc)) + EROSION_EXT * row;
        if (ex0 >= 0) *reinterpret_cast<float2*>(dst + ex0) = make_float2(r.v[k].x, r.v[k].y);
        if (ex0 + 3 < EROSION_EXT) *reinterpret_cast<float2*>(dst + ex0 + 2) = make_float2(r.v[k].z, r.v[k].w);
    }
}

// what pass 0 reads beside the committed planes: a first round's input is the start RAISED by the accumulated height of the layers above
// and the thickness of that raised state (plane 1; the un-raised start stays in plane 0), any other round's is the thickness of plane 0
MM_DEV void tile_derive(bool isFirst, float (*s_s)[EROSION_CELLS_EXT], float (*s_t)[EROSION_CELLS_EXT], const float* s_end, const float* s_acc)
{
    for (int c = opaque_tid(); c < EROSION_CELLS_EXT; c += EROSION_THREADS) {
        const float sv = s_s[0][c], ev = s_end[c];
        if (isFirst) { const float av = s_acc[c], ls = sv + av; s_s[1][c] = ls; s_t[1][c] = (ev + av) - ls; }
        else s_t[0][c] = ev - sv;
    }
}

// a thread's piece of the tile's 32 x 32 centre out of an LDS plane (q = 0 .. 255: row q / 8, columns 4 (q % 8) .. + 3)
MM_DEV F4 centre_piece(const float* plane, int q)
{
    const int c = EROSION_EXT * (EROSION_K + (q >> 3)) + EROSION_K + 4 * (q & 7);
    const float2 lo = *reinterpret_cast<const float2*>(plane + c), hi = *reinterpret_cast<const float2*>(plane + c + 2);
    return F4{lo.x, lo.y, hi.x, hi.y};
}
MM_DEV unsigned centre_offset(int tileX, int tileZ, int q) { return (unsigned)((tileX * 32 + 4 * (q & 7)) + ZS * (tileZ * 32 + (q >> 3))); }

// One round of one 32 x 32 tile whose planes tile_commit has staged: EROSION_K Jacobi passes of the round's layer on the (32 + 2 K)^2
// extended tile, then the centre's results to the round's output planes.  Called by every thread of the workgroup, begins behind a
// workgroup barrier after the commit and ends with its stores issued (not waited for).  Returns the tile's "changed" bits (bit j = pass
// j altered a column of the centre), the same value in every thread.
MM_DEV unsigned relax_tile(__amdgpu_buffer_rsrc_t work, const RoundPlanes& rp, int tileX, int tileZ, float (*s_s)[EROSION_CELLS_EXT],
                           float (*s_t)[EROSION_CELLS_EXT], float* s_end, float* s_acc, unsigned* s_passFlags /*[EROSION_K], zero*/)
{
    const int tid = opaque_tid();
    const bool isFirst = rp.isFirst;
    // extended tile: ex, ez in [0, EXT) <-> grid (gx0 + ex, gz0 + ez); cells beyond the grid do not exist (neighbours clamp to the edge)
    const int gx0 = tileX * 32 - EROSION_K, gz0 = tileZ * 32 - EROSION_K;
    const int exMin = imax(0, -gx0), exMax = imin(EROSION_EXT - 1, ZS - 1 - gx0);
    const int ezMin = imax(0, -gz0), ezMax = imin(EROSION_EXT - 1, ZS - 1 - gz0);
    const bool worker = tid < EROSION_EXT * EROSION_STRIPS;          // (the threads beyond only move pieces)
    const int ex = tid % EROSION_EXT, strip = tid / EROSION_EXT;
    const int rowLo = strip * EROSION_ROWS, rowHi = rowLo + EROSION_ROWS - 1;
    const int exL = imax(ex - 1, exMin), exR = imin(ex + 1, exMax);
    const bool ownCol = ex >= EROSION_K && ex < EROSION_K + 32;

    const float k1 = kMaterialAmpOrTan[MMGEN_NUM_STRATIFIED_MATERIALS + rp.layer];
    const float k2 = k1 * MM_SQRT_2;
    int cur = 0;                                             // plane holding the current state (after a first pass: 0 again)
    unsigned mask = 0u;
#pragma unroll 1
    for (int j = 0; j < EROSION_K; ++j) {
        // cells still exact after this pass: at distance > j from every tile border that is not an edge of the grid
        const int xl = gx0 < 0 ? exMin : j + 1, xh = gx0 + EROSION_EXT > ZS ? exMax : EROSION_EXT - 2 - j;
        const int zl = gz0 < 0 ? ezMin : j + 1, zh = gz0 + EROSION_EXT > ZS ? ezMax : EROSION_EXT - 2 - j;
        const bool colLive = worker && ex >= xl && ex <= xh;
        const int r0 = colLive ? imax(rowLo, zl) : 1, r1 = colLive ? imin(rowHi, zh) : 0;
        int flags;
        if (isFirst && j == 0) {
            flags = relax_strip<true>(s_s[1], s_t[1], s_s[0], s_t[0], s_end, s_acc, k1, k2, ex, exL, exR, r0, r1, ezMin, ezMax, ownCol);
            // the result is in plane 0 again
        } else {
            flags = relax_strip<false>(s_s[cur], s_t[cur], s_s[1 - cur], s_t[1 - cur], s_end, s_acc, k1, k2, ex, exL, exR, r0, r1, ezMin, ezMax, ownCol);
            cur = 1 - cur;
        }
        // both flag bits of the pass over the workgroup with ONE barrier: a wave's bits from two ballots, one LDS atomic per wave that has any
        const unsigned waveBits = (__any(flags & 1) ? 1u : 0u) | (__any(flags & 2) ? 2u : 0u);
        if (waveBits && (tid & 63) == 0) atomicOr(&s_passFlags[j], waveBits);
        __syncthreads();
        const unsigned any = s_passFlags[j];
        if (any & 2) mask |= 1u << j;
        if (isFirst && j == 0) {
            // the state after a layer's first pass is the layer's final state if that pass changed no column of the ZONE (next_phase): kept
            // in the layer's third plane by the tiles whose own centre it left alone (if it moved one, the zone's mask says so and the plane
            // is not looked at).  Plane 0 is read-only until the barrier behind the next pass.
            if (!(any & 2) && tid < 256) st4_dev<MM_ERODE_FIRST_AUX>(work, 4u * (rp.startFirst + centre_offset(tileX, tileZ, tid)), centre_piece(s_s[0], tid));
        } else if (!(any & 1)) break;      // a pass (other than a first pass) that changes no live cell of the tile is the identity from here on
    }
    // results of the centre
    if (tid < 256) {
        const unsigned g = centre_offset(tileX, tileZ, tid);
        st4_dev(work, 4u * (rp.startOut + g), centre_piece(s_s[cur], tid));
        if (rp.topFirst) {                 // an unchanged first pass keeps accIn: it has to exist
            float z = 0.f;
            asm volatile("" : "+v"(z));          // (made here: as a hoisted constant it occupies four registers for the whole launch)
            st4_dev(work, 4u * (rp.accIn + g), F4{z, z, z, z});
        }
    } else if (tid < 512) {
        st4_dev(work, 4u * (rp.accOut + centre_offset(tileX, tileZ, tid - 256)), centre_piece(s_acc, tid - 256));
    }
    return mask;
}

// The whole relaxation of a batch of zones in ONE launch (the host loop of Chunk::erodeZone chunk.cu:682-705 on the device).
// GROUPS PER XCD.  A workgroup reads the XCD it runs on (HW_REG_XCC_ID) and takes a number there; numbers g * groupSize .. (g + 1) *
// groupSize - 1 of an XCD form its group g.  A group that is complete draws zones from a launch-wide counter, one after the other, until
// none is left.  A zone's planes therefore never leave one XCD's L2 between its rounds (plain stores, sc1 loads: st4_dev), and which XCD
// a workgroup landed on decides only WHO works on a zone, never whether the result is right.  Nothing assumes that the launch is resident
// as a whole: the groups that exist do the work, workgroups that start late form later groups (and usually find no zone left), members
// of a group that can never become complete (the launch's last workgroups on an XCD) leave once every workgroup has started.  The host
// still sizes the launch so that it fits the chip: that is speed, not correctness.
// A round = EROSION_K passes of the zone's current layer on each of its 144 tiles, a group member's share of them (tiles member,
// member + groupSize, ...: fixed, so the next tile's planes are in flight while the current one is relaxed) one after the other, then a
// barrier among the group, then every member derives the next phase from the zone's "changed" mask exactly like the host loop does from
// its flag.  Zones do not wait for each other and the host is not involved.
// No wait is unbounded: after `timeoutTicks` of the 100 MHz clock the waiting workgroup raises *err = 0x80000000 | zone << 8 | round
// (zone 0x7fffff: waiting for the group to form or for its next zone), every workgroup of the launch leaves at its next look at that
// word, and the host reports MMGEN_ERROR_EROSION_STALL instead of hanging.  (Round 4's kernel took (zone, member) from one launch-wide
// ticket and could starve for good: workgroups that cannot start are bound to an XCD by the dispatcher, and their XCD's slots were held
// by spinning members of the very zone they belonged to - profiles/LOG.md.  Here a group only ever waits for workgroups of its own XCD
// with LOWER numbers than the ones still to come, i.e. for workgroups that have started.)
#define EROSION_WORD_REGISTERED 0      // the launch's words (behind the zones' states): workgroups that have started
#define EROSION_WORD_PASSES 1
#define EROSION_WORD_ERR 2             // ... [2..7] the error word and what the workgroup that gave up saw
#define EROSION_WORD_XCD 8             // ... [8..15] workgroups started per XCD
#define EROSION_WORD_ZONE_NEXT 16      // ... the next zone to hand out
#define EROSION_WORD_RESCUED 17        // ... zones the rescue pass had to relax (0 in a healthy launch)
#define EROSION_WORD_GROUPS 32         // ... then two words per group (XCD * 32 + group of the XCD): draws so far, the zone drawn last (-1: none left)
#define EROSION_GROUPS_PER_XCD 32
#define EROSION_LAUNCH_WORDS (EROSION_WORD_GROUPS + 2 * 8 * EROSION_GROUPS_PER_XCD)
// spins until *word >= want; false = gave up (the error word is set) or another workgroup did
MM_DEV bool erosion_spin(const unsigned* word, unsigned want, unsigned* err, unsigned* errHost, unsigned code, unsigned long long timeoutTicks)
{
    const unsigned long long t0 = wall_clock64();
    unsigned polls = 0u;
    // (one lane polls: the loaded value goes to a scalar register, so that `want` and the loop's other operands stay scalar too)
    while ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < want) {
        __builtin_amdgcn_s_sleep(4);
        if ((++polls & 63u) == 0u) {
            unsigned e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (e == 0u && (unsigned long long)(wall_clock64() - t0) > timeoutTicks) {
                e = code;
                if (atomicCAS(err, 0u, e) == 0u) {
                    if (errHost) __hip_atomic_store(errHost, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    // what the workgroup that gave up saw (read by the host's report)
                    err[2] = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); err[3] = want; err[4] = polls;
                    err[5] = (unsigned)((wall_clock64() - t0) >> 10);
                }
            }
            if (e != 0u) return false;
        }
    }
    return true;
}

template <bool RESCUE>
MM_DEV void erode_zones_body(const float* __restrict__ gatheredBase, size_t gatheredStride, const float* __restrict__ rawLayers, const float* __restrict__ rawHf,
              const int* __restrict__ zoneChunkIdx /*[zones][576]*/, float* workBase, ErosionState* states, unsigned* words, int zones, int groupSize,
              int groupWait /*workgroups a group waits for: groupSize (more only in the test of the give-up path)*/, int* maxPassesAlso /*nullable*/,
              unsigned* errHost /*host-visible copy of the error word, nullable*/, unsigned long long timeoutTicks)
{
    // RESCUE = false: the persistent launch.  RESCUE = true (k_erode_rescue): the pass behind it - workgroup i looks at zones i, i + gridDim.x,
    // ... and relaxes every zone that is NOT done (never drawn, or abandoned by a launch that gave up) from its raw planes, on its own: a
    // group of one waits for nobody, so this pass cannot stall and needs no residency.  A template parameter, not an argument: as an
    // argument it cost the persistent kernel 20 B of scratch and 0.1 ms (80 VGPRs is what six waves per SIMD allow).
    constexpr bool rescue = RESCUE;
    __shared__ float s_s[2][EROSION_CELLS_EXT];            // start planes, ping-pong over the passes
    __shared__ float s_t[2][EROSION_CELLS_EXT];            // thickness = end - start of the same states (what the neighbours compare)
    __shared__ float s_end[EROSION_CELLS_EXT];
    __shared__ float s_acc[EROSION_CELLS_EXT];             // accumulated heights; after the commit every cell is touched by its own lane only
    __shared__ int s_chunk[576];                           // the zone's 24 x 24 chunks (region path): raw pieces are addressed without a dependent global load
    __shared__ unsigned s_passFlags[EROSION_K];            // per pass of the current tile: bit 0 = a live cell changed, bit 1 = a cell of the centre did
    __shared__ int s_group[2];                             // the group's slot (XCD * 32 + group of the XCD) and this workgroup's member index (< 0: leave)
    __shared__ int s_zone;
    __shared__ int s_abort;
    __shared__ ErosionPhase s_ph;

    const int tid = threadIdx.x;
    unsigned* err = words + EROSION_WORD_ERR;
    const unsigned giveUpCode = 0x80000000u | (0x7fffffu << 8);
    if constexpr (rescue) {
        if (tid == 0) { s_abort = 0; s_group[0] = 0; s_group[1] = 0; }
    } else if (tid == 0) {
        s_abort = 0;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        const unsigned mine = __hip_atomic_fetch_add(&words[EROSION_WORD_XCD + xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the XCD's count is in before the launch-wide one says "started")
        __hip_atomic_fetch_add(&words[EROSION_WORD_REGISTERED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int g = (int)mine / groupSize;
        int member = (int)mine % groupSize;
        if (g >= EROSION_GROUPS_PER_XCD) member = -2;
        else {
            // the group is complete once the XCD has handed out the numbers of all its members; it never will be if every workgroup of the
            // launch has started and the XCD's count is still short (the last, partial group of an XCD: nothing to do)
            const unsigned need = (unsigned)(g * groupSize + groupWait);
            const unsigned long long t0 = wall_clock64();
            for (unsigned polls = 1u;; ++polls) {
                if (__hip_atomic_load(&words[EROSION_WORD_XCD + xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) break;
                if (__hip_atomic_load(&words[EROSION_WORD_REGISTERED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x && groupWait == groupSize) {
                    if (__hip_atomic_load(&words[EROSION_WORD_XCD + xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) member = -2;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
                if ((polls & 63u) == 0u) {
                    unsigned e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (e == 0u && (unsigned long long)(wall_clock64() - t0) > timeoutTicks) {
                        e = giveUpCode;
                        if (atomicCAS(err, 0u, e) == 0u && errHost) __hip_atomic_store(errHost, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    if (e != 0u) { member = -1; break; }
                }
            }
        }
        s_group[0] = (int)xcc * EROSION_GROUPS_PER_XCD + g; s_group[1] = member;
    }
    __syncthreads();
    const int member = __builtin_amdgcn_readfirstlane(s_group[1]), perZone = rescue ? 1 : groupSize;      // (workgroup-uniform: scalar registers)
    if (member < 0) return;
    unsigned* groupWords = words + EROSION_WORD_GROUPS + 2 * __builtin_amdgcn_readfirstlane(s_group[0]);
    // beside the caves (16 issue-bound waves per CU) this kernel's 8 waves mostly wait: at the highest wave priority they get the issue slot
    // whenever they can use one
    __builtin_amdgcn_s_setprio(3);

#pragma unroll 1
  for (unsigned draw = 1u;; ++draw) {
    // the group's next zone: its first member draws it and tells the others (who cannot be a zone behind: every zone has rounds, every
    // round a barrier among the group)
    if constexpr (rescue) {
        if (tid == 0) {
            int z = (int)blockIdx.x + (int)(draw - 1u) * (int)gridDim.x;
            while (z < zones && states[z].slot[0].done) z += (int)gridDim.x;      // (done: both slots final, see below)
            if (z < zones) {
                // the zone starts over from its raw planes (read-only inputs of the relaxation): the state k_erode_init gives it
                ErosionPhase f;
                f.layer = MMGEN_NUM_ERODED_MATERIALS - 1; f.isFirst = 1; f.done = 0; f.passes = 0; f.accSel = 0; f.fresh = 1; f.sel = 0u;
                store_phase(&states[z].slot[0], f); store_phase(&states[z].slot[1], f);
                for (int k = 0; k < 4; ++k) __hip_atomic_store(&states[z].changed[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&states[z].barrier, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&words[EROSION_WORD_RESCUED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            s_zone = z < zones ? z : -1;
        }
    } else if (tid == 0) {
        int z = -1;
        if (member == 0) {
            const unsigned d = __hip_atomic_fetch_add(&words[EROSION_WORD_ZONE_NEXT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            z = d < (unsigned)zones ? (int)d : -1;
            __hip_atomic_store(&groupWords[1], (unsigned)z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&groupWords[0], draw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (erosion_spin(&groupWords[0], draw, err, errHost, giveUpCode, timeoutTicks)) {
            z = (int)__hip_atomic_load(&groupWords[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_zone = z;
    }
    __syncthreads();
    const int zone = __builtin_amdgcn_readfirstlane(s_zone);
    if (zone < 0) return;
    ErosionState* st = states + zone;
    const RawPlanes raw = {gatheredBase ? gatheredBase + gatheredStride * zone : nullptr, rawLayers, rawHf, zoneChunkIdx ? zoneChunkIdx + 576 * zone : nullptr};
    if (raw.chunkIdx) for (int i = opaque_tid(); i < 576; i += EROSION_THREADS) s_chunk[i] = raw.chunkIdx[i];      // (visible behind the first round's barrier)
    const __amdgpu_buffer_rsrc_t work = __builtin_amdgcn_make_buffer_rsrc(workBase + ZONE_WORK_FLOATS * zone, 0, (int)(ZONE_WORK_FLOATS * sizeof(float)), 0x00020000);

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
                    atomicMax((int*)&words[EROSION_WORD_PASSES], ph.passes);
                    if (maxPassesAlso) __hip_atomic_fetch_max(maxPassesAlso, ph.passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (may be host memory)
                }
                __hip_atomic_store(&st->changed[(t + 1) & 3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        const ErosionPhase ph = s_ph;
        if (ph.done) break;
        const RoundPlanes rp = round_planes(ph);
        unsigned roundMask = 0u;
        TilePieces piece;
        int tile = member;
        if (tile < 144) tile_issue_loads(piece, raw, s_chunk, work, rp, tile % 12, tile / 12);
#pragma unroll 1
        while (tile < 144) {
            const int tileX = tile % 12, tileZ = tile / 12;
            tile_commit(piece, tileX, tileZ, s_s[0], s_end, s_acc);
            if (tid < EROSION_K) s_passFlags[tid] = 0u;
            __syncthreads();
            tile_derive(rp.isFirst, s_s, s_t, s_end, s_acc);
            __syncthreads();
            const int next = tile + perZone;
            if (next < 144) tile_issue_loads(piece, raw, s_chunk, work, rp, next % 12, next / 12);      // in flight while this tile is relaxed
            roundMask |= relax_tile(work, rp, tileX, tileZ, s_s, s_t, s_end, s_acc, s_passFlags);
            __syncthreads();                                          // the centre's pieces have been read: the planes may be overwritten
            tile = next;
        }
        // ---- barrier among the zone's group.  Release: every wave waits for its own stores (they are in the XCD's L2 then), then the
        // workgroup's barrier, then ONE lane publishes the mask (and waits for that too) and arrives.  The spin is bounded.
        MM_DRAIN_VMEM();
        __syncthreads();
        if (tid == 0) {
            if (roundMask) {
                __hip_atomic_fetch_or(&st->changed[t & 3], roundMask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                MM_DRAIN_VMEM();
            }
            __hip_atomic_fetch_add(&st->barrier, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!erosion_spin(&st->barrier, (unsigned)perZone * (unsigned)(t + 1), err, errHost, 0x80000000u | ((unsigned)zone << 8) | ((unsigned)t & 255u), timeoutTicks))
                s_abort = 1;
        }
#if MM_ERODE_ROUND_INV
        if (tid == 0) { asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory"); }
#endif
        __syncthreads();
        if (s_abort) return;
    }
    __syncthreads();                                                  // (s_chunk, s_ph and s_zone are rewritten for the next zone)
  }
}

__global__ void __launch_bounds__(EROSION_THREADS) __attribute__((amdgpu_waves_per_eu(6, 6)))
k_erode_zones(const float* __restrict__ gatheredBase, size_t gatheredStride, const float* __restrict__ rawLayers, const float* __restrict__ rawHf,
              const int* __restrict__ zoneChunkIdx, float* workBase, ErosionState* states, unsigned* words, int zones, int groupSize, int groupWait,
              int* maxPassesAlso, unsigned* errHost, unsigned long long timeoutTicks)
{
    erode_zones_body<false>(gatheredBase, gatheredStride, rawLayers, rawHf, zoneChunkIdx, workBase, states, words, zones, groupSize, groupWait, maxPassesAlso, errHost, timeoutTicks);
}

__global__ void __launch_bounds__(EROSION_THREADS) __attribute__((amdgpu_waves_per_eu(6, 6)))
k_erode_rescue(const float* __restrict__ gatheredBase, size_t gatheredStride, const float* __restrict__ rawLayers, const float* __restrict__ rawHf,
               const int* __restrict__ zoneChunkIdx, float* workBase, ErosionState* states, unsigned* words, int zones, int* maxPassesAlso, unsigned long long timeoutTicks)
{
    erode_zones_body<true>(gatheredBase, gatheredStride, rawLayers, rawHf, zoneChunkIdx, workBase, states, words, zones, 1, 1, maxPassesAlso, nullptr, timeoutTicks);
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

// the zones' state machines and the launch's words (ticket, pass count, error).  The accumulated heights need no clearing
// (thrust::fill_n of chunk.cu:679-680): a zone's first round takes them as zero without loading them and writes both buffers.
__global__ void k_erode_init(ErosionState* states, int zones, unsigned* ticket, int* clearWord /*nullable: the caller's pass-count word, cleared before its first batch*/)
{
    const int zone = blockIdx.x * blockDim.x + threadIdx.x;
    if (zone >= zones) return;
    ErosionPhase s;
    s.layer = MMGEN_NUM_ERODED_MATERIALS - 1; s.isFirst = 1; s.done = 0; s.passes = 0; s.accSel = 0; s.fresh = 1;
    s.sel = 0u;
    states[zone].slot[1] = s;            // round 0 reads slot[(0 - 1) & 1]
    states[zone].slot[0] = s;
    for (int k = 0; k < 4; ++k) states[zone].changed[k] = 0u;
    states[zone].barrier = 0u;
    for (int k = 0; k < 3; ++k) states[zone].reserved[k] = 0u;
    if (zone == 0) {
        for (int k = 0; k < EROSION_LAUNCH_WORDS; ++k) ticket[k] = 0u;
        if (clearWord) *clearWord = 0;
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
// k_erode_writeback + k_erosion_scatter do through the gathered buffer, for the quarter of each zone that is kept), and, with the
// chunk's eight planes in hand, Chunk::fixBackwardStratifiedLayers for it (chunk.cu:725-749: layers 10 and 11 become start_12 - layer).
// One workgroup per kept chunk.
__global__ void __launch_bounds__(256)
k_erode_finish(const float* __restrict__ workBase, const ErosionState* __restrict__ states, int lastT, const int* __restrict__ zoneChunkIdxOut /*[zones][144], -1 = skip*/,
               float* __restrict__ layersOut, int fixBackward, float* __restrict__ zoneCache /*nullable*/, const int* __restrict__ cacheSlot /*[zones], -1 = not kept*/)
{
    const int zone = blockIdx.y, cc = blockIdx.x;
    const int chunk = zoneChunkIdxOut[zone * 144 + cc];
    // zone cache (mmgen_region_set_zone_cache): ALL 144 kept chunks of the zone go into its slot, inside the region's grid or not
    float* keep = (zoneCache && cacheSlot[zone] >= 0) ? zoneCache + (size_t)cacheSlot[zone] * 144 * 8 * 256 + (size_t)cc * 8 * 256 + threadIdx.x : nullptr;
    if (chunk < 0 && !keep) return;
    const ErosionPhase* st = &states[zone].slot[lastT & 1];      // the phase the last launch ran with: done, all planes final
    const int t = threadIdx.x;
    const int cx = cc % 12 + 6, cz = cc / 12 + 6;
    const int gx = cx * 16 + (t & 15), gz = cz * 16 + (t >> 4);
    float* col = layersOut + (size_t)MMGEN_LAYERS_SIZE * (chunk < 0 ? 0 : chunk) + t;
    float start12 = 0.f;
#pragma unroll
    for (int plane = 0; plane < 8; ++plane) {
        const float v = workBase[ZONE_WORK_FLOATS * zone + ((size_t)plane * 3 + st->plane(plane)) * ZN + gx + ZS * gz];
        if (keep) keep[256 * plane] = v;
        if (chunk >= 0) col[256 * (MMGEN_NUM_STRATIFIED_MATERIALS + plane)] = v;
        if (plane == 0) start12 = v;
    }
    if (fixBackward && chunk >= 0) {
        col[256 * 10] = start12 - col[256 * 10];
        col[256 * 11] = start12 - col[256 * 11];
    }
}

}  // namespace mm

namespace mmk {

size_t erosion_work_bytes(int zones) { return (size_t)zones * ZONE_WORK_FLOATS * sizeof(float); }
// the zones' states, then one more record's worth of words: [0] = the launch's ticket counter, [1] = largest pass count of the zones,
// [2] = the launch's error word (0, or which zone's barrier gave up in which round)
size_t erosion_state_bytes(int zones) { return (size_t)zones * sizeof(mm::ErosionState) + sizeof(unsigned) * EROSION_LAUNCH_WORDS; }

// XCDs of the current device as the runtime reports them (hipDeviceAttributeNumberOfXccs); only where that fails, the gfx950 layout of 32
// CUs per XCD.  A wrong count only costs speed (groups that never fill leave, their zones go to the rescue pass), never a result.
static int device_xcds(int cus)
{
    static std::atomic<int> cached{0};
    int n = cached.load(std::memory_order_relaxed);
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || v < 1 || v > 8) v = cus >= 64 ? cus / 32 : 1;
        if (v > 8) v = 8;
        if (v < 1) v = 1;
        n = v;
        cached.store(n, std::memory_order_relaxed);
    }
    return n;
}

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

// How long a zone's workgroups wait for each other before the launch gives up (k_erode_zones): MMGEN_EROSION_TIMEOUT_MS in the
// environment, 2 000 ms otherwise - three orders of magnitude above the longest healthy wait (a round of one workgroup's tiles).
static std::atomic<long long> g_timeoutMs{-1};
static std::atomic<int> g_debugMissing{0};              // test hook: launch this many workgroups too few (mmgen_debug_erosion_stall)
static unsigned long long erosion_timeout_ticks()
{
    long long ms = g_timeoutMs.load(std::memory_order_relaxed);
    if (ms < 0) {
        const char* e = getenv("MMGEN_EROSION_TIMEOUT_MS");
        ms = (e && atoll(e) > 0) ? atoll(e) : 2000;
        g_timeoutMs.store(ms, std::memory_order_relaxed);
    }
    return (unsigned long long)ms * 100000ull;              // wall_clock64(): 100 MHz
}
// Persistent relaxations of one device run one after the other, whatever streams they are enqueued on.  Each is sized to be resident as a
// whole; two of them in flight would share the chip's slots, and a workgroup that has not started cannot take a slot in another XCD than
// the one the dispatcher has bound it to (workgroup i of a launch goes to XCD i mod 8) - measured: with zone 0 finished and half the chip
// free, the last workgroups of zone 1 waited for slots of two XCDs that the spinning rest of zone 1 occupied, for as long as they spun.
// An event chain per device orders the launches without blocking the host; the bounded spin stays as the last line (another process).
static std::mutex g_chainMu;
static std::map<int, hipEvent_t> g_chain;
static int chain_before_launch(hipStream_t s, hipEvent_t* ev)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    auto it = g_chain.find(dev);
    if (it == g_chain.end()) {
        hipEvent_t n = nullptr;
        if ((e = hipEventCreateWithFlags(&n, hipEventDisableTiming)) != hipSuccess) return (int)e;
        it = g_chain.emplace(dev, n).first;
    } else if ((e = hipStreamWaitEvent(s, it->second, 0)) != hipSuccess) return (int)e;
    *ev = it->second;
    return 0;
}

static std::atomic<long long> g_stalls{0}, g_rescued{0};
void erosion_rescue_counts(long long* stalls, long long* zonesRescued)
{
    if (stalls) *stalls = g_stalls.load(std::memory_order_relaxed);
    if (zonesRescued) *zonesRescued = g_rescued.load(std::memory_order_relaxed);
}
void erosion_note_stall() { g_stalls.fetch_add(1, std::memory_order_relaxed); }

void erosion_debug_stall(int missingWorkgroups, int timeoutMs)
{
    g_debugMissing.store(missingWorkgroups < 0 ? 0 : missingWorkgroups, std::memory_order_relaxed);
    g_timeoutMs.store(timeoutMs > 0 ? timeoutMs : -1, std::memory_order_relaxed);
}

// Enqueues the relaxation of `zones` packed zone buffers (stride in floats) to convergence: ONE persistent launch, then the kernel that
// moves the final planes out.  Nothing is read back unless the caller asks for the pass count (maxPasses != null: the stream is
// synchronised, like the reference's erodeZone, and a relaxation that gave up is reported as MMGEN_ERROR_EROSION_STALL); maxPassesDev
// (device, may be null) is raised to the largest pass count with the stream; errHost (host-visible, may be null) receives the error word
// of a launch that gave up (written by the device, nothing on success).
int erode_zones(float* gathered, size_t strideFloats, int zones, float* work, mm::ErosionState* states, float* accOut, size_t accStride,
                hipStream_t s, int* maxPasses, const int* zoneChunkIdxOut, float* layersOut, int* maxPassesDev, hipEvent_t beforeRelaxation,
                const float* rawLayers, const float* rawHf, const int* zoneChunkIdx, int workgroupsPer4Cu, const unsigned** startedCounter,
                unsigned* startedTarget, unsigned* errHost, bool clearPassesDev, bool fixBackward, float* zoneCache, const int* zoneCacheSlot)
{
    if (!gathered && !(rawLayers && rawHf && zoneChunkIdx && layersOut)) return (int)hipErrorInvalidValue;
    if (zones <= 0) return 0;
    unsigned* words = (unsigned*)(states + zones);        // the launch's words (EROSION_WORD_*)
    int* passesWord = (int*)(words + EROSION_WORD_PASSES);
    MMK_LAUNCH(KID_ERODE_INIT, mm::k_erode_init, dim3((zones + 63) / 64), dim3(64), s, states, zones, words, clearPassesDev ? maxPassesDev : (int*)nullptr);
    // The launch is resident as a whole: at most what the chip holds (or the caller's cap), a multiple of the XCD count so that the
    // dispatcher's round-robin gives every XCD the same share.  Groups: with few zones every XCD's workgroups form one group (all of them
    // on one zone); with many, groups of 32 - measured on the 72-zone bench tile alone on the chip (96 workgroups per XCD): groups of 32 /
    // 48 / 96 -> 1.65 / 1.87 / 2.77 ms, 2.1 / 2.3 / 2.0 GB of traffic; beside the caves an XCD has 32 workgroups anyway (16: 21.9 ms per
    // step instead of 21.6).
    const int cus = device_cus();
    const int nXcd = device_xcds(cus);
    static const int forcedCap = [] { const char* e = getenv("MMGEN_EROSION_WG_PER_4CU"); return e ? atoi(e) : 0; }();      // (measurements: the DAG's share of the chip in a serial run)
    int grid = erosion_resident_workgroups(forcedCap > 0 ? forcedCap : workgroupsPer4Cu);
    if (grid > nXcd) grid -= grid % nXcd;
    const int perXcd = grid / nXcd > 0 ? grid / nXcd : 1;
    int groupSize = perXcd;
    if (zones > nXcd && perXcd >= 64) groupSize = 32;
    {
        static const int forced = [] { const char* e = getenv("MMGEN_EROSION_GROUP"); return e ? atoi(e) : 0; }();      // (experiments: workgroups per zone)
        if (forced > 0 && forced <= perXcd) groupSize = forced;
    }
    if (groupSize > 144) groupSize = 144;
    // (fewer zones than groups: the surplus groups find nothing to do; launching fewer workgroups than the XCDs' shares would only unbalance them)
    const int groupWait = groupSize + g_debugMissing.load(std::memory_order_relaxed);      // (test hook: groups that can never become complete)
    if (beforeRelaxation) { hipError_t e = hipEventRecord(beforeRelaxation, s); if (e != hipSuccess) return (int)e; }
    // the launch counts the workgroups that have STARTED (k_erode_init has just cleared the word): a caller that wants them on the chip
    // before it launches something that takes every slot waits for the counter to reach the grid size (launch_caves)
    if (startedCounter) *startedCounter = words + EROSION_WORD_REGISTERED;
    if (startedTarget) *startedTarget = (unsigned)grid;
    {
        std::lock_guard<std::mutex> lk(g_chainMu);
        hipEvent_t chain = nullptr;
        int ce = chain_before_launch(s, &chain);
        if (ce) return ce;
        MMK_LAUNCH(KID_ERODE_PASS, mm::k_erode_zones, dim3(grid), dim3(EROSION_THREADS), s, (const float*)gathered, strideFloats, rawLayers, rawHf,
                   zoneChunkIdx, work, states, words, zones, groupSize, groupWait, maxPassesDev, errHost, erosion_timeout_ticks());
        hipError_t re = hipEventRecord(chain, s);
        if (re != hipSuccess) return (int)re;
    }
    // The rescue pass: whatever the persistent launch left undone - a zone nobody drew (fewer complete groups than the host assumed: a
    // device whose XCDs the dispatcher fills unevenly, a CU mask), the zones of a launch that gave up (starved by another process'
    // persistent kernel) - is relaxed here by single workgroups that wait for nobody.  In a healthy step every workgroup finds its zones
    // done and leaves: one launch of a few workgroups.  The reference's host loop cannot stall (chunk.cu:682-705); with this pass neither can
    // ours, it can only be slow.
    {
        const int rescueGrid = zones < 64 ? zones : 64;
        MMK_LAUNCH(KID_ERODE_RESCUE, mm::k_erode_rescue, dim3(rescueGrid), dim3(EROSION_THREADS), s, (const float*)gathered, strideFloats, rawLayers, rawHf,
                   zoneChunkIdx, work, states, words, zones, maxPassesDev, erosion_timeout_ticks());
    }
    const int perZone = groupSize;
    if (layersOut) {
        // region path: no in-place contract to honour, the kept chunks' planes go straight to the layers
        MMK_LAUNCH(KID_EROSION_SCATTER, mm::k_erode_finish, dim3(144, zones), dim3(256), s, (const float*)work, (const mm::ErosionState*)states, 0,
                   zoneChunkIdxOut, layersOut, fixBackward ? 1 : 0, zoneCache, zoneCacheSlot);
    } else {
        MMK_LAUNCH(KID_ERODE_WRITEBACK, mm::k_erode_writeback, dim3(ZN / 256, 1, zones), dim3(256), s, gathered, strideFloats, work, states, accOut,
                   accStride, 0);
    }
    if (maxPasses) {
        static_assert(EROSION_WORD_ERR == EROSION_WORD_PASSES + 1, "the pass count and the error word are read with one copy");
        int hw[3] = {0, 0, 0};                             // largest pass count, error word of the persistent launch, zones the rescue pass relaxed
        hipError_t e = hipMemcpyAsync(hw, passesWord, 2 * sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(hw + 2, words + EROSION_WORD_RESCUED, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return (int)e;
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return (int)e;
        *maxPasses = hw[0];
        if (hw[1] != 0) {
            // the persistent launch gave up; its zones were relaxed by the rescue pass (same planes, same pass counts): report, do not fail
            unsigned tail[EROSION_LAUNCH_WORDS] = {0};
            (void)hipMemcpy(tail, words, sizeof(tail), hipMemcpyDeviceToHost);
            g_stalls.fetch_add(1, std::memory_order_relaxed);
            fprintf(stderr, "mmgen: the erosion relaxation gave up waiting (zone %d of the launch - 8388607 = the registration -, round %d; %d zones, groups of %d workgroups); "
                            "%u zones were relaxed by the rescue pass\n",
                    (int)(((unsigned)hw[1] >> 8) & 0x7FFFFFu), hw[1] & 255, zones, perZone, tail[EROSION_WORD_RESCUED]);
            fprintf(stderr, "mmgen:   workgroups started: %u of %d; gave up at %u arrivals of %u after %u polls, %u x 10.24 us; per XCD: %u %u %u %u %u %u %u %u\n",
                    tail[EROSION_WORD_REGISTERED], grid, tail[EROSION_WORD_ERR + 2], tail[EROSION_WORD_ERR + 3], tail[EROSION_WORD_ERR + 4], tail[EROSION_WORD_ERR + 5],
                    tail[EROSION_WORD_XCD], tail[EROSION_WORD_XCD + 1], tail[EROSION_WORD_XCD + 2], tail[EROSION_WORD_XCD + 3], tail[EROSION_WORD_XCD + 4],
                    tail[EROSION_WORD_XCD + 5], tail[EROSION_WORD_XCD + 6], tail[EROSION_WORD_XCD + 7]);
        }
        if (hw[2] != 0) g_rescued.fetch_add(hw[2], std::memory_order_relaxed);
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
