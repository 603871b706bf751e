// mmgen erosion for gfx950: the relaxation ("slope method") of the 8 eroded layers over a 384x384-column zone grid.
// Behavioural spec: kernDoErosion chunk.cu:477-601 + the host loop of Chunk::erodeZone chunk.cu:682-705, copyLayers :603-656.
//
// Design (MI355X-first):
//  * every relaxation pass is a synchronous Jacobi step: it reads (start, accumulated) from one buffer of a ping-pong pair
//    and writes the other, so no workgroup ever reads a halo cell another workgroup is rewriting (the reference updates in
//    place across thread blocks; its result depends on block scheduling — DESIGN.md "Canonical semantics");
//  * the convergence loop lives on the device: the last workgroup of a pass (agent-scope ticket) folds the "changed" flags
//    and advances a per-zone state {layer, isFirst, done}; the host just enqueues passes back to back on one stream and
//    only reads one word every few dozen passes.  Passes launched after a zone is done exit immediately;
//  * many zones run in one launch (blockIdx.z = zone), 12x12 tiles of 32x32 columns per zone with 34x34 LDS halo tiles;
//  * the zone working set (2 planes + 2 accumulators in flight) is 2.4 MB: L2 / Infinity-Cache resident, HBM sees it once.
#include <hip/hip_runtime.h>
#include <vector>
#include "mm_biome.cuh"
#include "mmgen_erosion.h"

namespace mm {

#define ZS MMGEN_EROSION_GRID_SIDE
#define ZN MMGEN_EROSION_GRID_NUM_COLS

// Per-zone workspace layout (floats): work[8 layers][2][ZN] ping-pong start planes, acc[2][ZN].
#define ZONE_WORK_FLOATS ((size_t)(8 * 2 + 2) * ZN)

__global__ void __launch_bounds__(1024)
k_erode_pass(const float* __restrict__ gatheredBase, size_t gatheredStride, float* __restrict__ workBase, ErosionState* __restrict__ states)
{
    __shared__ float s_start[34 * 34];
    __shared__ float s_end[34 * 34];
    __shared__ int s_changed;

    const int zone = blockIdx.z;
    ErosionState* st = states + zone;
    if (st->done) return;
    const int layer = st->layer;
    const bool isFirst = st->isFirst != 0;
    const int p = st->parity[layer];          // buffer holding the current start plane of this layer (ignored on the first pass)
    const int ap = st->accParity;

    const float* gathered = gatheredBase + gatheredStride * zone;
    float* work = workBase + ZONE_WORK_FLOATS * zone;
    float* accIn = work + (size_t)16 * ZN + (size_t)ap * ZN;
    float* accOut = work + (size_t)16 * ZN + (size_t)(1 - ap) * ZN;
    const float* startIn = isFirst ? (gathered + (size_t)layer * ZN) : (work + ((size_t)layer * 2 + p) * ZN);
    float* startOut = work + ((size_t)layer * 2 + (1 - p)) * ZN;
    // end plane = final start plane of the layer above (already eroded), or the heightfield plane for the top layer
    const float* endIn = (layer == MMGEN_NUM_ERODED_MATERIALS - 1) ? (gathered + (size_t)8 * ZN)
                                                                   : (work + ((size_t)(layer + 1) * 2 + st->parity[layer + 1]) * ZN);

    const int lx = threadIdx.x, lz = threadIdx.y;
    const int lid = lx + 32 * lz;
    const int bx = blockIdx.x * 32, bz = blockIdx.y * 32;
    const int gx = bx + lx, gz = bz + lz;
    const int c = gx + ZS * gz;
    if (lid == 0) s_changed = 0;

    const float thisAcc = isFirst ? accIn[c] : 0.f;
    const float accPrev = accIn[c];
    const float thisStart = startIn[c] + thisAcc;
    const float thisEnd = endIn[c] + thisAcc;
    const int sc = (lx + 1) + 34 * (lz + 1);
    s_start[sc] = thisStart;
    s_end[sc] = thisEnd;

    // halo: 132 border cells, clamped to the grid (values outside extend the border, chunk.cu:545)
    if (lid < 132) {
        int hx, hz;
        if (lid < 32) { hx = lid + 1; hz = 0; }
        else if (lid < 64) { hx = lid - 32 + 1; hz = 33; }
        else if (lid < 96) { hx = 0; hz = lid - 64 + 1; }
        else if (lid < 128) { hx = 33; hz = lid - 96 + 1; }
        else { hx = (lid & 1) ? 33 : 0; hz = (lid & 2) ? 33 : 0; }
        const int px = imin(imax(bx - 1 + hx, 0), ZS - 1), pz = imin(imax(bz - 1 + hz, 0), ZS - 1);
        const int n = px + ZS * pz;
        const float a = isFirst ? accIn[n] : 0.f;
        s_start[hx + 34 * hz] = startIn[n] + a;
        s_end[hx + 34 * hz] = endIn[n] + a;
    }
    __syncthreads();

    float newStart = thisStart;
    float maxThickness = thisEnd - thisStart;
    const float tanAoR = kMaterialAmpOrTan[MMGEN_NUM_STRATIFIED_MATERIALS + layer];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = sc + kDirX[i] + 34 * kDirZ[i];
        const float ns = s_start[n];
        newStart = gmax(newStart, ns - tanAoR * ((i & 1) ? MM_SQRT_2 : 1.f));
        maxThickness = gmax(maxThickness, s_end[n] - ns);
    }
    newStart = gmin(newStart, thisEnd);

    // the reference writes only when maxThickness > 0; otherwise the stored plane keeps its previous value (without the lift)
    float outStart = startIn[c];
    float outAcc = accPrev;
    if (maxThickness > 0.f) {
        outStart = newStart;
        if (newStart != thisStart) {
            s_changed = 1;
            outAcc = accPrev + (newStart - thisStart);
        }
    }
    startOut[c] = outStart;
    accOut[c] = outAcc;
    __syncthreads();

    // fold flags; the last workgroup of this zone advances the state machine (host loop of chunk.cu:682-705)
    if (lid == 0) {
        if (s_changed) atomicOr(&st->changed, 1);
        __threadfence();
        const int ticket = atomicAdd(&st->ticket, 1);
        if (ticket == (int)(gridDim.x * gridDim.y) - 1) {
            __threadfence();
            const int changed = atomicOr(&st->changed, 0);
            st->passes += 1;
            st->parity[layer] = 1 - p;
            st->accParity = 1 - ap;
            if (changed) {
                st->isFirst = 0;
            } else {
                if (layer == 0) st->done = 1;
                else { st->layer = layer - 1; st->isFirst = 1; }
            }
            st->changed = 0;
            st->ticket = 0;
            __threadfence();
        }
    }
}

// final planes back into the caller's gathered-layers buffer (in-place contract of Chunk::erodeZone) and the accumulated heights
__global__ void __launch_bounds__(256)
k_erode_writeback(float* __restrict__ gatheredBase, size_t gatheredStride, const float* __restrict__ workBase, const ErosionState* __restrict__ states,
                  float* __restrict__ accOutBase, size_t accStride)
{
    const int zone = blockIdx.z;
    const ErosionState* st = states + zone;
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
        ErosionState s;
        s.layer = MMGEN_NUM_ERODED_MATERIALS - 1; s.isFirst = 1; s.changed = 0; s.ticket = 0; s.done = 0; s.passes = 0; s.accParity = 0;
        for (int l = 0; l < 8; ++l) s.parity[l] = 0;
        states[zone] = s;
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
                hipStream_t s, int* maxPasses, void (*prof)(int, hipStream_t, bool))
{
    if (zones <= 0) return 0;
    hipLaunchKernelGGL(mm::k_erode_init, dim3((2 * ZN + 255) / 256, zones), dim3(256), 0, s, states, work, zones);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;

    std::vector<mm::ErosionState> h(zones);
    const dim3 grid(12, 12, zones), block(32, 32);
    int launched = 0;
    for (;;) {
        const int batch = launched == 0 ? 48 : 16;
        if (prof) prof(0, s, true);
        for (int i = 0; i < batch; ++i) hipLaunchKernelGGL(mm::k_erode_pass, grid, block, 0, s, gathered, strideFloats, work, states);
        if (prof) prof(batch, s, false);
        launched += batch;
        e = hipMemcpyAsync(h.data(), states, sizeof(mm::ErosionState) * zones, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return (int)e;
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return (int)e;
        bool all = true;
        for (auto& z : h) all = all && z.done;
        if (all) break;
        if (launched > 100000) return (int)hipErrorLaunchFailure;
    }
    if (maxPasses) { int m = 0; for (auto& z : h) m = z.passes > m ? z.passes : m; *maxPasses = m; }
    hipLaunchKernelGGL(mm::k_erode_writeback, dim3(ZN / 256, 1, zones), dim3(256), 0, s, gathered, strideFloats, work, states, accOut, accStride);
    e = hipGetLastError();
    return (int)e;
}

int erosion_gather(const float* layers, const float* hf, const int* zoneChunkIdx, int zones, float* gathered, size_t strideFloats, hipStream_t s)
{
    if (zones <= 0) return 0;
    hipLaunchKernelGGL(mm::k_erosion_gather, dim3(576, 9, zones), dim3(256), 0, s, layers, hf, zoneChunkIdx, gathered, strideFloats);
    return (int)hipGetLastError();
}

int erosion_scatter(const float* gathered, size_t strideFloats, const int* zoneChunkIdxOut, int zones, float* layersOut, hipStream_t s)
{
    if (zones <= 0) return 0;
    hipLaunchKernelGGL(mm::k_erosion_scatter, dim3(144, 8, zones), dim3(256), 0, s, gathered, strideFloats, zoneChunkIdxOut, layersOut);
    return (int)hipGetLastError();
}

}  // namespace mmk
