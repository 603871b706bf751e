// Internal interface of the erosion translation unit (mmgen_erosion.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace mm {
struct ErosionPhase {       // state of the relaxation loop of one zone as seen by one pass (host loop of chunk.cu:682-705)
    int layer;              // eroded layer being relaxed, 7 -> 0
    int isFirst;            // first pass of this layer (adds the accumulated lift of the layers above)
    int done;               // all 8 layers converged
    int passes;             // relaxation passes executed before this one
    int accParity;          // which accumulator buffer is current
    int fresh;              // written by k_erode_init: the first pass takes it as is
    int parity[8];          // which ping-pong plane holds the current start plane of each layer
};
// Per-zone device-side state machine WITHOUT same-address atomics: pass t reads the phase pass t-1 ran with (slot[(t-1) & 1]) and
// pass t-1's "some column changed" word (changed[(t-1) & 3]), derives its own phase (every workgroup redundantly, a few scalar
// ops), and workgroup (0,0) stores it to slot[t & 1] for pass t+1.  Kernel boundaries on the stream order everything.
struct ErosionState {
    ErosionPhase slot[2];
    int changed[4];         // changed[t & 3] = 1 if pass t altered any column; pass t clears changed[(t + 1) & 3]
};
}  // namespace mm

namespace mmk {
size_t erosion_work_bytes(int zones);
size_t erosion_state_bytes(int zones);
int erode_zones(float* gathered, size_t strideFloats, int zones, float* work, mm::ErosionState* states, float* accOut, size_t accStride,
                hipStream_t s, int* maxPasses);
int erosion_gather(const float* layers, const float* hf, const int* zoneChunkIdx, int zones, float* gathered, size_t strideFloats, hipStream_t s);
int erosion_scatter(const float* gathered, size_t strideFloats, const int* zoneChunkIdxOut, int zones, float* layersOut, hipStream_t s);
}  // namespace mmk
