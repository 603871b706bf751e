// Internal interface of the erosion translation unit (mmgen_erosion.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace mm {
struct ErosionState {       // per-zone device-side state machine of the relaxation loop (host loop of chunk.cu:682-705)
    int layer;              // eroded layer being relaxed, 7 -> 0
    int isFirst;            // first pass of this layer (adds the accumulated lift of the layers above)
    int changed;            // any column changed in the pass in flight
    int ticket;             // workgroups finished in the pass in flight
    int done;               // all 8 layers converged
    int passes;             // relaxation passes executed
    int accParity;          // which accumulator buffer is current
    int parity[8];          // which ping-pong plane holds the current start plane of each layer
};
}  // namespace mm

namespace mmk {
size_t erosion_work_bytes(int zones);
size_t erosion_state_bytes(int zones);
int erode_zones(float* gathered, size_t strideFloats, int zones, float* work, mm::ErosionState* states, float* accOut, size_t accStride,
                hipStream_t s, int* maxPasses, void (*prof)(int, hipStream_t, bool));
int erosion_gather(const float* layers, const float* hf, const int* zoneChunkIdx, int zones, float* gathered, size_t strideFloats, hipStream_t s);
int erosion_scatter(const float* gathered, size_t strideFloats, const int* zoneChunkIdxOut, int zones, float* layersOut, hipStream_t s);
}  // namespace mmk
