// Per-kernel timing (HIP events on the launch stream) and roctx stage ranges, shared by every translation unit of libmmgen.
//   * MMK_LAUNCH(kid, kernel, grid, block, stream, args...) launches `kernel` and, when profiling is enabled
//     (mmgen_profile_enable), brackets it with an event pair recorded on the SAME stream the kernel runs on - bench.py reads the
//     per-kernel averages from these inside its timed region (torch.cuda.Event would only see torch's current stream);
//   * mmk::StageRange names a pipeline stage for rocprofv3 --marker-trace (roctx push / pop; a few ns when no tool is attached).
#pragma once
#include <hip/hip_runtime.h>

namespace mmk {

enum KernelId {
    KID_HEIGHTFIELD, KID_LAYERS, KID_FIX_BACKWARD, KID_CAVE_COLUMNS, KID_CAVE_VOXELS, KID_CAVE_BIOMES, KID_FILL, KID_FILL_FAR, KID_FILL_LUSH, KID_FILL_BASE, KID_FILL_SCAN, KID_PROBE,
    KID_EROSION_GATHER, KID_ERODE_INIT, KID_ERODE_PASS, KID_ERODE_WRITEBACK, KID_EROSION_SCATTER,
    KID_FEATURE_PLACEMENTS, KID_GATHER_PLACEMENTS, KID_APPLY_FEATURES, KID_DECORATORS, KID_FEATURE_BOX,
    KID_SELECT, KID_RING_NEED, KID_COPY_PLACEMENTS, KID_RING_PACK, KID_RING_UNPACK,
    KID_MESH_COUNT, KID_MESH_FILL, KID_PACK_COUNT, KID_PACK_FILL, KID_UNPACK, KID_ERODE_RESCUE,
    KID_COUNT
};

void profile_enable(bool on);
bool profile_enabled();
int profile_num_kernels();
const char* profile_kernel_name(int id);
// Synchronises the recorded events, accumulates total milliseconds and launch counts per kernel id, and clears the records.
int profile_collect(double* total_ms, long long* counts);
void profile_begin(int kid, hipStream_t s);
void profile_end(hipStream_t s);

// compute units of the CURRENT device (cached per device: a process may drive several GPUs); 0 on error
int device_cus();

struct StageRange {
    explicit StageRange(const char* name);
    ~StageRange();
    StageRange(const StageRange&) = delete;
    StageRange& operator=(const StageRange&) = delete;
};

}  // namespace mmk

#define MMK_LAUNCH_NORET(KID, KERNEL, GRID, BLOCK, STREAM, ...)                               \
    do {                                                                                      \
        const bool prof_ = mmk::profile_enabled();                                            \
        if (prof_) mmk::profile_begin((KID), (STREAM));                                       \
        hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, (STREAM), __VA_ARGS__);                    \
        if (prof_) mmk::profile_end((STREAM));                                                \
    } while (0)

#define MMK_LAUNCH(KID, KERNEL, GRID, BLOCK, STREAM, ...)                                     \
    do {                                                                                      \
        MMK_LAUNCH_NORET(KID, KERNEL, GRID, BLOCK, STREAM, __VA_ARGS__);                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) return (int)e_;                                                 \
    } while (0)
