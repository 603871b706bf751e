// Internal interface of the feature translation unit (mmgen_features.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mmgen_types.h"

namespace mmk {
int prepare_features();    // builds this translation unit's noise-table image on the current device (called from mmgen_init)
int launch_feature_placements(const float* hf, const float* bw, const float* layers, const mmgen_cave_layer* cl, const int32_t* pos, int n,
                              mmgen_feature_placement* fp, mmgen_cave_feature_placement* cfp, int* counts, const int* chunkList,
                              const uint8_t* colNeed /*nullable*/, hipStream_t s);
int launch_ring_need(const float* bw, const int32_t* pos, const int* chunkList, int n, const uint8_t* cellLazy, int rx0, int rz0, int rx1, int rz1,
                     uint8_t* colNeed, hipStream_t s);
int launch_gather_placements(const mmgen_feature_placement* fp, const mmgen_cave_feature_placement* cfp, const int* counts, const int* target,
                             int nOut, int gridW, int gridH, mmgen_feature_placement* gfp, mmgen_cave_feature_placement* gcfp, int* bounds,
                             const int32_t* gridPos, hipStream_t s, int* maxGathered = nullptr,
                             // the region folds three small launches into this one: the capacity check of every source cell's cave list
                             // (capHost: host-visible word, capMax: device word) and the clearing of the rasterisers' work counters
                             int* capHost = nullptr, int* capMax = nullptr, unsigned* zeroWords = nullptr, int nZeroWords = 0,
                             // ... and the copy of the targets' heightfields out of the grid (hfGrid [grid][256] -> hfOut [nOut][256])
                             const float* hfGrid = nullptr, float* hfOut = nullptr);
// workCounter: apply_work_bytes() of device scratch that nothing else uses while the kernel runs (the waves draw their work from it)
size_t apply_work_bytes();
int launch_apply_features(uint8_t* blocks, const int32_t* pos, int n, const mmgen_feature_placement* gfp, const mmgen_cave_feature_placement* gcfp,
                          const int* bounds, const int* srcIdx, unsigned* workCounter, hipStream_t s, bool workCleared = false);
int launch_decorators(uint8_t* blocks, const float* hf, const float* bw, const mmgen_cave_layer* cl, const int32_t* pos, int n, const int* srcIdx, hipStream_t s);
int table_dump_floats();
int launch_dump_tables(float* out, hipStream_t s);
int launch_feature_box(int isCave, int feature, const int* fpos, int layerHeight, const int* boxMin, const int* boxSize, uint8_t* out, hipStream_t s);
}  // namespace mmk
