// libmmgen C ABI (include/mmgen.h): argument checks, scratch ownership, stream plumbing.  No torch, no C++ types across.
#include "../../include/mmgen.h"
#include "mmgen_kernels.h"
#include "mmgen_erosion.h"
#include "mmgen_features.h"
#include "mmgen_prof.h"
#include <map>
#include <mutex>
#include <utility>
#include <cstring>
#include <cstdio>

namespace {
// Library-owned scratch of the per-stage calls, keyed by (device, stream): two calls on different streams or devices never share a
// buffer (calls on ONE stream are ordered by the stream, so they may).  Grow-only; a buffer is only replaced after its stream is idle.
struct Scratch {
    float* colInfo = nullptr; size_t colInfoChunks = 0;            // [chunks][256] float2 per-column cave info
    float* erodeWork = nullptr; mm::ErosionState* erodeState = nullptr; int erodeZones = 0;
    unsigned* fillQueue = nullptr; size_t fillQueueBytes = 0;        // deferred clay / moss voxels, then apply_features hand-over flags
};
std::mutex g_mu;
std::map<std::pair<int, void*>, Scratch> g_scratch;
int g_device = -1;

Scratch* scratch_for(void* stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    return &g_scratch[{dev, stream}];                                // std::map nodes are stable: the pointer stays valid
}

template <class T> int regrow(T*& p, size_t bytes, void* stream)
{
    if (p) {
        hipError_t e = hipStreamSynchronize((hipStream_t)stream);   // the old buffer may still be in use by queued work of this stream
        if (e != hipSuccess) return (int)e;
        e = hipFree(p);
        if (e != hipSuccess) return (int)e;
        p = nullptr;
    }
    return (int)hipMalloc((void**)&p, bytes);
}

int ensure_erosion(Scratch& sc, int zones, void* stream)
{
    if (zones <= sc.erodeZones) return 0;
    sc.erodeZones = 0;
    int e = regrow(sc.erodeWork, mmk::erosion_work_bytes(zones), stream);
    if (e) return e;
    e = regrow(sc.erodeState, mmk::erosion_state_bytes(zones), stream);
    if (e) return e;
    sc.erodeZones = zones;
    return 0;
}

int ensure_fill_queue(Scratch& sc, int n, void* stream)
{
    const size_t want = mmk::fill_queue_bytes(n);
    if (want <= sc.fillQueueBytes) return 0;
    sc.fillQueueBytes = 0;
    const int e = regrow(sc.fillQueue, want, stream);
    if (e) return e;
    sc.fillQueueBytes = want;
    return 0;
}

int ensure_col_info(Scratch& sc, int n, void* stream)
{
    if ((size_t)n <= sc.colInfoChunks) return 0;
    sc.colInfoChunks = 0;
    const int e = regrow(sc.colInfo, mmk::cave_scratch_bytes(n), stream);
    if (e) return e;
    sc.colInfoChunks = (size_t)n;
    return 0;
}
}  // namespace

extern "C" {

int mmgen_init(int device)
{
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) return (int)e;
    if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "mmgen: device %d is %s, this library carries gfx950 code only\n", device, p.gcnArchName);
        return (int)hipErrorNoBinaryForGpu;
    }
    g_device = device;
    // per-device noise-table images of the translation units that evaluate simplex noise: built here once, so that stage calls
    // after init only launch (no synchronisation on their first use, safe under stream capture)
    int ne = mmk::prepare_kernels();
    if (ne) return ne;
    return mmk::prepare_features();
}

const char* mmgen_error_string(int code)
{
    if (code == MMGEN_ERROR_PLACEMENT_OVERFLOW)
        return "a chunk's cave placement list exceeded MMGEN_CFP_CAP and lost entries (mmgen_region_max_cave_placements acknowledges)";
    if (code == MMGEN_ERROR_EROSION_STALL)
        return "the erosion relaxation gave up: a zone's workgroups did not meet within MMGEN_EROSION_TIMEOUT_MS (since round 6 the rescue pass finishes such zones; no call returns this any more)";
    return hipGetErrorString((hipError_t)code);
}

// Library scratch of the per-stage calls is kept per (device, stream) and only grows.  A host that creates and destroys streams calls
// mmgen_release(stream) before destroying one (a recycled handle would otherwise inherit the old buffers, and the entry would stay
// for good); mmgen_release_all() frees everything, e.g. before unloading the library.  Both synchronise the streams they release.
int mmgen_release(void* stream)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    // the stream first: if that fails the entry (and the buffers queued work may still use) stays where it is
    e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    Scratch sc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_scratch.find({dev, stream});
        if (it == g_scratch.end()) return 0;
        sc = it->second;
        g_scratch.erase(it);
    }
    // every buffer is freed even when one hipFree fails; the first error is reported
    hipError_t first = hipSuccess;
    for (void* p : {(void*)sc.colInfo, (void*)sc.erodeWork, (void*)sc.erodeState, (void*)sc.fillQueue})
        if (p && (e = hipFree(p)) != hipSuccess && first == hipSuccess) first = e;
    return (int)first;
}

int mmgen_release_all()
{
    hipError_t e = hipDeviceSynchronize();                      // first: on failure nothing has been taken out of the map
    if (e != hipSuccess) return (int)e;
    std::map<std::pair<int, void*>, Scratch> all;
    { std::lock_guard<std::mutex> lk(g_mu); all.swap(g_scratch); }
    hipError_t first = hipSuccess;
    for (auto& kv : all)
        for (void* p : {(void*)kv.second.colInfo, (void*)kv.second.erodeWork, (void*)kv.second.erodeState, (void*)kv.second.fillQueue})
            if (p && (e = hipFree(p)) != hipSuccess && first == hipSuccess) first = e;
    return (int)first;
}

int mmgen_reserve(int max_chunks_per_call, void* stream)
{
    Scratch* sc = scratch_for(stream);
    if (!sc) return (int)hipErrorInvalidDevice;
    const int e = ensure_col_info(*sc, max_chunks_per_call, stream);
    return e ? e : ensure_fill_queue(*sc, max_chunks_per_call, stream);
}

int mmgen_generate_heightfields(const int32_t* d_pos, int n, float* d_hf, float* d_bw, void* stream)
{
    if (n < 0 || (n > 0 && (!d_pos || !d_hf || !d_bw))) return (int)hipErrorInvalidValue;
    return mmk::launch_heightfield(d_pos, n, d_hf, d_bw, nullptr, (hipStream_t)stream);
}

int mmgen_generate_heightfields_gathered(const int32_t* d_pos, int n, float* d_hf, float* d_bw, float* d_gathered, void* stream)
{
    if (n < 0 || (n > 0 && (!d_pos || !d_hf || !d_bw || !d_gathered))) return (int)hipErrorInvalidValue;
    return mmk::launch_heightfield(d_pos, n, d_hf, d_bw, d_gathered, (hipStream_t)stream);
}

int mmgen_generate_layers(const float* d_gathered, const float* d_bw, const int32_t* d_pos, int n, float* d_layers, void* stream)
{
    if (n < 0 || (n > 0 && (!d_gathered || !d_bw || !d_pos || !d_layers))) return (int)hipErrorInvalidValue;
    return mmk::launch_layers(d_gathered, d_bw, d_pos, n, d_layers, (hipStream_t)stream);
}

int mmgen_fix_backward_layers(float* d_layers, int n, void* stream)
{
    if (n < 0 || (n > 0 && !d_layers)) return (int)hipErrorInvalidValue;
    return mmk::launch_fix_backward(d_layers, n, (hipStream_t)stream);
}

int mmgen_erode_zones(float* d_gathered, int num_zones, float* d_acc, int* max_passes, void* stream)
{
    if (num_zones < 0 || (num_zones > 0 && !d_gathered)) return (int)hipErrorInvalidValue;
    if (num_zones == 0) return 0;
    Scratch* sc = scratch_for(stream);
    if (!sc) return (int)hipErrorInvalidDevice;
    int e = ensure_erosion(*sc, num_zones, stream);
    if (e) return e;
    int dummy = 0;       // asking for the pass count is what makes the call synchronous (the contract of the per-stage entry points)
    return mmk::erode_zones(d_gathered, (size_t)MMGEN_GATHERED_LAYERS_SIZE, num_zones, sc->erodeWork, sc->erodeState, d_acc,
                            (size_t)MMGEN_EROSION_GRID_NUM_COLS, (hipStream_t)stream, max_passes ? max_passes : &dummy);
}

int mmgen_erode_zone(float* d_gathered, float* d_acc, void* stream) { return mmgen_erode_zones(d_gathered, 1, d_acc, nullptr, stream); }

int mmgen_erosion_stalls(long long* stalls, long long* zones_rescued)
{
    mmk::erosion_rescue_counts(stalls, zones_rescued);
    return 0;
}

int mmgen_debug_erosion_stall(int missing_workgroups, int timeout_ms)
{
    mmk::erosion_debug_stall(missing_workgroups, timeout_ms);
    return 0;
}

int mmgen_generate_caves(const float* d_hf, const float* d_bw, const int32_t* d_pos, int n, mmgen_cave_layer* d_cl, void* stream)
{
    if (n < 0 || (n > 0 && (!d_hf || !d_bw || !d_pos || !d_cl))) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    Scratch* sc = scratch_for(stream);
    if (!sc) return (int)hipErrorInvalidDevice;
    int e = ensure_col_info(*sc, n, stream);
    if (e) return e;
    return mmk::launch_caves(d_hf, d_bw, d_pos, n, d_cl, sc->colInfo, (int)sc->colInfoChunks, nullptr, nullptr, (hipStream_t)stream);
}

int mmgen_fill(const float* d_hf, const float* d_bw, const float* d_layers, const mmgen_cave_layer* d_cl, const int32_t* d_pos, int n,
               const mmgen_feature_placement* d_fp, const mmgen_cave_feature_placement* d_cfp, const int32_t* d_bounds, uint8_t* d_blocks,
               void* stream)
{
    if (n < 0 || (n > 0 && (!d_hf || !d_bw || !d_layers || !d_cl || !d_pos || !d_blocks))) return (int)hipErrorInvalidValue;
    if ((d_fp || d_cfp) && !d_bounds) return (int)hipErrorInvalidValue;
    Scratch* sc = scratch_for(stream);
    if (!sc) return (int)hipErrorInvalidDevice;
    int e = ensure_fill_queue(*sc, n, stream);
    if (e) return e;
    e = mmk::launch_fill(d_hf, d_bw, d_layers, d_cl, d_pos, n, d_blocks, nullptr, sc->fillQueue, mmk::fill_queue_bytes(n), false /* positions live on the device */, (hipStream_t)stream);
    if (e || !(d_fp || d_cfp)) return e;
    return mmk::launch_apply_features(d_blocks, d_pos, n, d_fp, d_cfp, d_bounds, nullptr, sc->fillQueue /* k_fill_lush is done with it */, (hipStream_t)stream);
}

int mmgen_generate_feature_placements(const float* d_hf, const float* d_bw, const float* d_layers, const mmgen_cave_layer* d_cl, const int32_t* d_pos,
                                      int n, mmgen_feature_placement* d_fp, mmgen_cave_feature_placement* d_cfp, int32_t* d_counts, void* stream)
{
    if (n < 0 || (n > 0 && (!d_hf || !d_bw || !d_layers || !d_cl || !d_pos || !d_fp || !d_cfp || !d_counts))) return (int)hipErrorInvalidValue;
    return mmk::launch_feature_placements(d_hf, d_bw, d_layers, d_cl, d_pos, n, d_fp, d_cfp, d_counts, nullptr, nullptr, (hipStream_t)stream);
}

int mmgen_gather_feature_placements(const mmgen_feature_placement* d_fp, const mmgen_cave_feature_placement* d_cfp, const int32_t* d_counts,
                                    const int32_t* d_targets, int num_targets, int grid_w, int grid_h, mmgen_feature_placement* d_gfp,
                                    mmgen_cave_feature_placement* d_gcfp, int32_t* d_bounds, void* stream)
{
    if (num_targets < 0 || grid_w <= 0 || grid_h <= 0) return (int)hipErrorInvalidValue;
    if (num_targets > 0 && (!d_fp || !d_cfp || !d_counts || !d_targets || !d_gfp || !d_gcfp || !d_bounds)) return (int)hipErrorInvalidValue;
    return mmk::launch_gather_placements(d_fp, d_cfp, d_counts, d_targets, num_targets, grid_w, grid_h, d_gfp, d_gcfp, d_bounds, nullptr, (hipStream_t)stream);
}

int mmgen_place_decorators(uint8_t* d_blocks, const float* d_hf, const float* d_bw, const mmgen_cave_layer* d_cl, const int32_t* d_pos, int n,
                           void* stream)
{
    if (n < 0 || (n > 0 && (!d_blocks || !d_hf || !d_bw || !d_cl || !d_pos))) return (int)hipErrorInvalidValue;
    return mmk::launch_decorators(d_blocks, d_hf, d_bw, d_cl, d_pos, n, nullptr, (hipStream_t)stream);
}

int mmgen_debug_feature_box(int is_cave, int feature, const int32_t* h_feature_pos, int layer_height, const int32_t* h_box_min,
                            const int32_t* h_box_size, uint8_t* d_out, void* stream)
{
    if (!h_feature_pos || !h_box_min || !h_box_size || !d_out) return (int)hipErrorInvalidValue;
    return mmk::launch_feature_box(is_cave, feature, h_feature_pos, layer_height, h_box_min, h_box_size, d_out, (hipStream_t)stream);
}

int mmgen_debug_set_lush_queue_cap(int entries) { mmk::debug_set_lush_queue_cap(entries); return 0; }

int mmgen_debug_tables(float* d_out, int capacity_floats, void* stream)
{
    if (!d_out) return mmk::table_dump_floats();                 // query: number of floats
    if (capacity_floats < mmk::table_dump_floats()) return (int)hipErrorInvalidValue;
    return mmk::launch_dump_tables(d_out, (hipStream_t)stream);
}

int mmgen_debug_probe(int fn, const float* d_in, int n, float* d_out, void* stream)
{
    if (n < 0 || (n > 0 && (!d_in || !d_out))) return (int)hipErrorInvalidValue;
    return mmk::launch_probe(fn, d_in, n, d_out, (hipStream_t)stream);
}

void mmgen_profile_enable(int on) { mmk::profile_enable(on != 0); }
int mmgen_profile_num_kernels(void) { return mmk::profile_num_kernels(); }
const char* mmgen_profile_kernel_name(int id) { return mmk::profile_kernel_name(id); }
int mmgen_profile_collect(double* total_ms, long long* counts) { return mmk::profile_collect(total_ms, counts); }

}  // extern "C"
