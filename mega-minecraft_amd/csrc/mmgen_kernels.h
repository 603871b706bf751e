// Host-side launchers of the mmgen HIP kernels (internal to libmmgen; the public surface is include/mmgen.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mmgen_types.h"

namespace mmk {
int launch_heightfield(const int32_t* pos, int n, float* hf, float* bw, float* gathered /*nullable*/, hipStream_t s);
int launch_layers(const float* gathered, const float* bw, const int32_t* pos, int n, float* layers, hipStream_t s,
                  float* stratifiedCopy = nullptr /*the first nCopy chunks' twelve stratified layers are also stored here ([chunk][20][256] layout)*/,
                  int nCopy = 0);
int launch_fix_backward(float* layers, int n, hipStream_t s);
// colInfoScratch: cave_scratch_bytes(colInfoChunks) of device scratch, colInfoChunks > every chunk index the launch uses
size_t cave_scratch_bytes(int chunks);
int launch_caves(const float* hf, const float* bw, const int32_t* pos, int n, mmgen_cave_layer* caveLayers, float* colInfoScratch, int colInfoChunks,
                 const int* chunkList /*nullable: chunks to process*/, const uint8_t* colNeed /*nullable: [chunk][256], lazy ring*/, hipStream_t s,
                 hipEvent_t afterVoxels = nullptr /*recorded once the layers' extents are final (before their biomes)*/,
                 int biomeWorkgroupsPerCu = 0 /*0 = as many as fit; fewer leave room for a kernel that runs beside k_cave_biomes*/,
                 hipEvent_t beforeVoxels = nullptr /*the stream waits for it between k_cave_columns and k_cave_voxels*/,
                 const unsigned* waitCounter = nullptr, unsigned waitTarget = 0 /*... and then until *waitCounter >= waitTarget (bounded)*/);
int launch_fill(const float* hf, const float* bw, const float* layers, const mmgen_cave_layer* caveLayers, const int32_t* pos, int n,
                uint8_t* blocks, const int* srcIdx /*nullable: input chunk of each output chunk*/,
                unsigned* lushQueue /*nullable device scratch: deferred clay / moss voxels*/, size_t lushQueueBytes, bool allInPruneDomain /* every chunk within MM_PRUNE_DOMAIN blocks of the origin: k_fill_far is not launched */, hipStream_t s,
                bool countersCleared = false /* launch_fill_clear already ran on this scratch */,
                // out (nullable): a device word that counts k_fill_cave's persistent workgroups as they start, and the value it reaches
                const unsigned** startedCounter = nullptr, unsigned* startedTarget = nullptr, hipEvent_t beforeCave = nullptr);
// one lane waits (bounded: ~3 ms) until *counter >= target: orders a launch behind the START of a persistent one on another stream
int launch_wait_counter(const unsigned* counter, unsigned target, hipStream_t s);
int launch_fill_clear(int n, unsigned* lushQueue, size_t lushQueueBytes, hipStream_t s);
size_t fill_queue_bytes(int n);      // launch_fill's scratch for n chunks: lush queue (2 048 deferred voxels per chunk on average; overflow is evaluated in place) + row lists (393 KB per chunk, at most 8 192 chunks' worth) + work counters
void debug_set_lush_queue_cap(int entries);      // test-only (include/mmgen.h mmgen_debug_set_lush_queue_cap)
int launch_probe(int fn, const float* in, int n, float* out, hipStream_t s);
int prepare_kernels();      // builds this translation unit's noise-table image on the current device (called from mmgen_init)
}  // namespace mmk
