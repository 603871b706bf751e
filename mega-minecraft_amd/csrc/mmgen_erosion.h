// Internal interface of the erosion translation unit (mmgen_erosion.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#ifndef EROSION_K
#define EROSION_K 6         // relaxation passes per launch (temporal blocking; 44 x 44-cell LDS tiles).  Larger K = fewer launches but more ring cells and
                            // LDS.  Measured per step of the bench tile (72 zones), K x row groups per tile: 3 x 19 2.07 ms, 4 x 4 1.85, 4 x 10 1.66, 5 x 14 1.82,
                            // 6 x 4 2.13, 6 x 11 1.54, 6 x 22 1.87, 8 x 12 3.23 (round 2, four row groups: K = 2 / 3 / 4 / 6 / 8 / 12 -> 3.1 / 3.1 / 2.3 / 2.6 / 3.3 / 3.4)
#endif

namespace mm {
struct ErosionPhase {       // state of the relaxation loop of one zone as seen by one launch (host loop of chunk.cu:682-705)
    int layer;              // eroded layer being relaxed, 7 -> 0
    int isFirst;            // this launch starts the layer (its first pass adds the accumulated lift of the layers above)
    int done;               // all 8 layers converged
    int passes;             // relaxation passes the reference's loop would have executed before this launch
    int accSel;             // which accumulator buffer is current
    int fresh;              // written by k_erode_init: the first launch takes it as is
    unsigned sel;           // 2 bits per layer: which of the layer's three planes holds its current start plane (2 = state after its first
                            // pass).  Packed, not an array: a dynamically indexed member would put the whole phase in scratch
    __host__ __device__ int plane(int l) const { return (int)((sel >> (2 * l)) & 3u); }
    __host__ __device__ void setPlane(int l, int v) { sel = (sel & ~(3u << (2 * l))) | ((unsigned)v << (2 * l)); }
};
// Per-zone device-side state machine: launch t reads the phase launch t-1 ran with (slot[(t-1) & 1]) and launch t-1's "changed" mask
// (changed[(t-1) & 3], bit j = pass j altered a column), derives its own phase (every workgroup redundantly, a few scalar ops), and
// workgroup (0,0) stores it to slot[t & 1] for launch t+1.  Kernel boundaries on the stream order everything.
struct ErosionState {
    ErosionPhase slot[2];
    unsigned changed[4];    // round t ORs into changed[t & 3] and clears changed[(t + 1) & 3]
    unsigned barrier;       // arrivals of the zone's workgroups at the end of their rounds (round t is over at perZone * (t + 1))
    unsigned reserved[3];
};
}  // namespace mm

namespace mmk {
size_t erosion_work_bytes(int zones);
size_t erosion_state_bytes(int zones);
// layersOut != null (region path): the kept 12 x 12 chunks of every zone (zoneChunkIdxOut, [zones][144], -1 = skip) go straight into the
// chunk-major layers and `gathered` is left as it was; else the final planes are written back into `gathered` (Chunk::erodeZone's contract)
int erode_zones(float* gathered, size_t strideFloats, int zones, float* work, mm::ErosionState* states, float* accOut, size_t accStride,
                hipStream_t s, int* maxPasses, const int* zoneChunkIdxOut = nullptr, float* layersOut = nullptr, int* maxPassesDev = nullptr,
                hipEvent_t beforeRelaxation = nullptr /* recorded on s right before the persistent launch */,
                // gathered == null (region path): the raw planes are read straight from the chunk-major raw layers / heightfields through
                // the zones' 24 x 24 chunk lists ([zones][576]) - no k_erosion_gather, no packed copy
                const float* rawLayers = nullptr, const float* rawHf = nullptr, const int* zoneChunkIdx = nullptr,
                // workgroups of the persistent launch per FOUR CUs: 0 = as many as fit (the fastest relaxation on its own); a caller that
                // runs another kernel beside it leaves room (the region: the caves' workgroups take the rest of every CU)
                int workgroupsPer4Cu = 0,
                // out (nullable): a device word that counts the persistent launch's workgroups as they start, and the value it reaches
                const unsigned** startedCounter = nullptr, unsigned* startedTarget = nullptr,
                // host-visible (mapped) word that receives the error word of a launch that gave up (k_erode_zones); nullable
                unsigned* errHost = nullptr,
                // region path: clear *maxPassesDev before this batch (the first of a region's batches), and apply
                // Chunk::fixBackwardStratifiedLayers to the kept chunks in the kernel that writes their eroded planes
                bool clearPassesDev = false, bool fixBackward = false,
                // region path, zone cache: the 144 kept chunks' planes of zone z also go to zoneCache + zoneCacheSlot[z] * 144 * 8 * 256 floats
                // (slot -1 = not kept)
                float* zoneCache = nullptr, const int* zoneCacheSlot = nullptr);
// test hook: the next persistent launches wait for `missingWorkgroups` more workgroups than they have (the wait can then never complete) and give up after timeoutMs
// (0, 0 restores the defaults)
void erosion_debug_stall(int missingWorkgroups, int timeoutMs);
// process-wide: persistent relaxations that gave up (as far as the host has learnt of them) and zones the rescue pass relaxed (synchronous calls only)
void erosion_rescue_counts(long long* stalls, long long* zonesRescued);
void erosion_note_stall();
int erosion_gather(const float* layers, const float* hf, const int* zoneChunkIdx, int zones, float* gathered, size_t strideFloats, hipStream_t s);
int erosion_scatter(const float* gathered, size_t strideFloats, const int* zoneChunkIdxOut, int zones, float* layersOut, hipStream_t s);
}  // namespace mmk
