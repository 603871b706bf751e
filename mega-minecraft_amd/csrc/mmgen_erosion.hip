// mmgen erosion for gfx950: the relaxation ("slope method") of the 8 eroded layers over a 384x384-column zone grid.
// Behavioural spec: kernDoErosion chunk.cu:477-601 + the host loop of Chunk::erodeZone chunk.cu:682-705, copyLayers :603-656.
//
// Design (MI355X-first):
//  * every relaxation pass is a synchronous Jacobi step: it reads (start, accumulated) from one buffer of a ping-pong pair
//    and writes the other, so no workgroup ever reads a halo cell another workgroup is rewriting (the reference updates in
//    place across thread blocks; its result depends on block scheduling — DESIGN.md "Canonical semantics");
//  * the convergence loop lives on the device: every pass derives its phase {layer, isFirst, done, parities} from the previous
//    pass's phase and its "changed" word (mmgen_erosion.h) with plain loads/stores — no ticket, no same-address atomics (144
//    agent-scope atomics on one word per zone and pass cost more than the relaxation itself); the host just enqueues passes back
//    to back on one stream and only reads the state every few dozen passes.  Passes launched after a zone is done exit immediately;
//  * many zones run in one launch (blockIdx.z = zone), 12x12 tiles of 32x32 columns per zone with 34x34 LDS halo tiles;
//  * the zone working set (2 planes + 2 accumulators in flight) is 2.4 MB: L2 / Infinity-Cache resident, HBM sees it once.
#include <hip/hip_runtime.h>
#include <vector>
#include "mm_biome.cuh"
#include "mmgen_erosion.h"
#include "mmgen_prof.h"

namespace mm {

#define ZS MMGEN_EROSION_GRID_SIDE
#define ZN MMGEN_EROSION_GRID_NUM_COLS

// Per-zone workspace layout (floats): work[8 layers][2][ZN] ping-pong start planes, acc[2][ZN].
#define ZONE_WORK_FLOATS ((size_t)(8 * 2 + 2) * ZN)

MM_DEV ErosionPhase next_phase(const ErosionPhase& prev, int changedPrev)
{
    ErosionPhase cur = prev;
    if (prev.fresh) { cur.fresh = 0; return cur; }
    if (prev.done) return cur;
    cur.passes = prev.passes + 1;
    cur.parity[prev.layer] = 1 - prev.parity[prev.layer];
    cur.accParity = 1 - prev.accParity;
    if (changedPrev) cur.isFirst = 0;
    else if (prev.layer == 0) cur.done = 1;
    else { cur.layer = prev.layer - 1; cur.isFirst = 1; }
    return cur;
}

// One workgroup = one 32x32-column tile; EROSION_ROWS rows of 32 lanes, each lane relaxes 32 / EROSION_ROWS columns (independent
// loads in flight per lane; a 1024-lane workgroup spends its life in two barriers and one dependent load)
#ifndef EROSION_ROWS
#define EROSION_ROWS 4
#endif
#define EROSION_CELLS (32 / EROSION_ROWS)
__global__ void __launch_bounds__(32 * EROSION_ROWS)
k_erode_pass(const float* __restrict__ gatheredBase, size_t gatheredStride, float* __restrict__ workBase, ErosionState* __restrict__ states, int t)
{
    __shared__ float s_start[34 * 34];
    __shared__ float s_end[34 * 34];
    __shared__ int s_changed;

    const int zone = blockIdx.z;
    ErosionState* st = states + zone;
    const int lx = threadIdx.x, lz0 = threadIdx.y;
    const int lid = lx + 32 * lz0;
    const ErosionPhase ph = next_phase(st->slot[(t + 1) & 1], st->changed[(t + 3) & 3]);
    if (lid == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
        st->slot[t & 1] = ph;
        st->changed[(t + 1) & 3] = 0;
    }
    if (ph.done) return;
    const int layer = ph.layer;
    const bool isFirst = ph.isFirst != 0;
    const int p = ph.parity[layer];           // buffer holding the current start plane of this layer (ignored on the first pass)
    const int ap = ph.accParity;

    const float* gathered = gatheredBase + gatheredStride * zone;
    float* work = workBase + ZONE_WORK_FLOATS * zone;
    const float* accIn = work + (size_t)16 * ZN + (size_t)ap * ZN;
    float* accOut = work + (size_t)16 * ZN + (size_t)(1 - ap) * ZN;
    const float* startIn = isFirst ? (gathered + (size_t)layer * ZN) : (work + ((size_t)layer * 2 + p) * ZN);
    float* startOut = work + ((size_t)layer * 2 + (1 - p)) * ZN;
    // end plane = final start plane of the layer above (already eroded), or the heightfield plane for the top layer
    const float* endIn = (layer == MMGEN_NUM_ERODED_MATERIALS - 1) ? (gathered + (size_t)8 * ZN)
                                                                   : (work + ((size_t)(layer + 1) * 2 + ph.parity[layer + 1]) * ZN);

    const int bx = blockIdx.x * 32, bz = blockIdx.y * 32;
    if (lid == 0) s_changed = 0;

    float rawStart[EROSION_CELLS], accPrev[EROSION_CELLS], thisStart[EROSION_CELLS], thisEnd[EROSION_CELLS];
#pragma unroll
    for (int k = 0; k < EROSION_CELLS; ++k) {
        const int lz = lz0 + EROSION_ROWS * k;
        const int c = (bx + lx) + ZS * (bz + lz);
        rawStart[k] = startIn[c];
        accPrev[k] = accIn[c];
        const float thisAcc = isFirst ? accPrev[k] : 0.f;
        thisStart[k] = rawStart[k] + thisAcc;
        thisEnd[k] = endIn[c] + thisAcc;
        const int sc = (lx + 1) + 34 * (lz + 1);
        s_start[sc] = thisStart[k];
        s_end[sc] = thisEnd[k];
    }

    // halo: 132 border cells, clamped to the grid (values outside extend the border, chunk.cu:545)
    for (int h = lid; h < 132; h += 32 * EROSION_ROWS) {
        int hx, hz;
        if (h < 32) { hx = h + 1; hz = 0; }
        else if (h < 64) { hx = h - 32 + 1; hz = 33; }
        else if (h < 96) { hx = 0; hz = h - 64 + 1; }
        else if (h < 128) { hx = 33; hz = h - 96 + 1; }
        else { hx = (h & 1) ? 33 : 0; hz = (h & 2) ? 33 : 0; }
        const int px = imin(imax(bx - 1 + hx, 0), ZS - 1), pz = imin(imax(bz - 1 + hz, 0), ZS - 1);
        const int n = px + ZS * pz;
        const float a = isFirst ? accIn[n] : 0.f;
        s_start[hx + 34 * hz] = startIn[n] + a;
        s_end[hx + 34 * hz] = endIn[n] + a;
    }
    __syncthreads();

    const float tanAoR = kMaterialAmpOrTan[MMGEN_NUM_STRATIFIED_MATERIALS + layer];
    bool changed = false;
#pragma unroll
    for (int k = 0; k < EROSION_CELLS; ++k) {
        const int lz = lz0 + EROSION_ROWS * k;
        const int c = (bx + lx) + ZS * (bz + lz);
        const int sc = (lx + 1) + 34 * (lz + 1);
        float newStart = thisStart[k];
        float maxThickness = thisEnd[k] - thisStart[k];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = sc + kDirX[i] + 34 * kDirZ[i];
            const float ns = s_start[n];
            newStart = gmax(newStart, ns - tanAoR * ((i & 1) ? MM_SQRT_2 : 1.f));
            maxThickness = gmax(maxThickness, s_end[n] - ns);
        }
        newStart = gmin(newStart, thisEnd[k]);

        // the reference writes only when maxThickness > 0; otherwise the stored plane keeps its previous value (without the lift)
        float outStart = rawStart[k];
        float outAcc = accPrev[k];
        if (maxThickness > 0.f) {
            outStart = newStart;
            if (newStart != thisStart[k]) {
                changed = true;
                outAcc = accPrev[k] + (newStart - thisStart[k]);
            }
        }
        startOut[c] = outStart;
        accOut[c] = outAcc;
    }
    if (changed) s_changed = 1;
    __syncthreads();

    if (lid == 0 && s_changed) st->changed[t & 3] = 1;      // plain store of the same value by every workgroup that changed something
}

// final planes back into the caller's gathered-layers buffer (in-place contract of Chunk::erodeZone) and the accumulated heights
__global__ void __launch_bounds__(256)
k_erode_writeback(float* __restrict__ gatheredBase, size_t gatheredStride, const float* __restrict__ workBase, const ErosionState* __restrict__ states,
                  float* __restrict__ accOutBase, size_t accStride, int lastT)
{
    const int zone = blockIdx.z;
    const ErosionPhase* st = &states[zone].slot[lastT & 1];      // the phase the last launched pass ran with: done, all parities final
    const int c = blockIdx.x * 256 + threadIdx.x;
    const float* work = workBase + ZONE_WORK_FLOATS * zone;
    float* gathered = gatheredBase + gatheredStride * zone;
#pragma unroll
    for (int l = 0; l < 8; ++l) gathered[(size_t)l * ZN + c] = work[((size_t)l * 2 + st->parity[l]) * ZN + c];
    if (accOutBase) accOutBase[accStride * zone + c] = work[(size_t)16 * ZN + (size_t)st->accParity * ZN + c];
}

__global__ void k_erode_init(ErosionState* states, float* workBase, int zones)
{
    const int zone = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (zone >= zones) return;
    // zero both accumulator buffers (thrust::fill_n of chunk.cu:679-680)
    float* acc = workBase + ZONE_WORK_FLOATS * zone + (size_t)16 * ZN;
    if (i < 2 * ZN) acc[i] = 0.f;
    if (i == 0) {
        ErosionPhase s;
        s.layer = MMGEN_NUM_ERODED_MATERIALS - 1; s.isFirst = 1; s.done = 0; s.passes = 0; s.accParity = 0; s.fresh = 1;
        for (int l = 0; l < 8; ++l) s.parity[l] = 0;
        states[zone].slot[1] = s;            // pass 0 reads slot[(0 - 1) & 1]
        states[zone].slot[0] = s;
        for (int k = 0; k < 4; ++k) states[zone].changed[k] = 0;
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

}  // namespace mm

namespace mmk {

size_t erosion_work_bytes(int zones) { return (size_t)zones * ZONE_WORK_FLOATS * sizeof(float); }
size_t erosion_state_bytes(int zones) { return (size_t)zones * sizeof(mm::ErosionState); }

// Runs the relaxation to convergence for `zones` packed zone buffers (stride in floats).  Synchronises the stream (the
// reference's erodeZone is synchronous too).  Returns 0 or a hipError_t; *maxPasses receives the largest pass count.
int erode_zones(float* gathered, size_t strideFloats, int zones, float* work, mm::ErosionState* states, float* accOut, size_t accStride,
                hipStream_t s, int* maxPasses)
{
    if (zones <= 0) return 0;
    MMK_LAUNCH(KID_ERODE_INIT, mm::k_erode_init, dim3((2 * ZN + 255) / 256, zones), dim3(256), s, states, work, zones);
    hipError_t e = hipSuccess;

    std::vector<mm::ErosionState> h(zones);
    const dim3 grid(12, 12, zones), block(32, EROSION_ROWS);
    int launched = 0;
    for (;;) {
        const int batch = launched == 0 ? 48 : 16;
        // one event pair around the whole batch of passes (an event per 19 us launch would perturb what it measures)
        const bool prof_ = profile_enabled();
        if (prof_) profile_begin(KID_ERODE_PASS, s);
        for (int i = 0; i < batch; ++i) hipLaunchKernelGGL(mm::k_erode_pass, grid, block, 0, s, gathered, strideFloats, work, states, launched + i);
        if (prof_) profile_end(s);
        launched += batch;
        e = hipMemcpyAsync(h.data(), states, sizeof(mm::ErosionState) * zones, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return (int)e;
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return (int)e;
        // slot[(launched - 1) & 1] = the phase the last launched pass ran with; `done` shows up there one pass after convergence
        bool all = true;
        for (auto& z : h) all = all && z.slot[(launched - 1) & 1].done;
        if (all) break;
        if (launched > 100000) return (int)hipErrorLaunchFailure;
    }
    if (maxPasses) { int m = 0; for (auto& z : h) { const int ps = z.slot[(launched - 1) & 1].passes; m = ps > m ? ps : m; } *maxPasses = m; }
    MMK_LAUNCH(KID_ERODE_WRITEBACK, mm::k_erode_writeback, dim3(ZN / 256, 1, zones), dim3(256), s, gathered, strideFloats, work, states, accOut,
               accStride, launched - 1);
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
