// Per-kernel HIP-event timing + roctx stage ranges (see mmgen_prof.h).  Host code only.
#include "mmgen_prof.h"
#include <rocprofiler-sdk-roctx/roctx.h>
#include <atomic>
#include <mutex>
#include <vector>

namespace mmk {

namespace {
struct ProfRec { int id; hipEvent_t a, b; };
std::mutex g_mu;
std::atomic<bool> g_prof{false};      // read on every launch by any thread, written under g_mu
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
thread_local ProfRec t_open;

const char* const kKernelNames[KID_COUNT] = {
    "k_heightfield", "k_layers", "k_fix_backward", "k_cave_columns", "k_cave_voxels", "k_cave_biomes", "k_fill_cave", "k_fill_far", "k_fill_lush", "k_fill_base", "k_fill_scan", "k_probe",
    "k_erosion_gather", "k_erode_init", "k_erode_zones", "k_erode_writeback", "k_erosion_scatter",
    "k_feature_placements", "k_gather_placements", "k_apply_features", "k_decorators", "k_feature_box",
    "k_select", "k_ring_need", "k_copy_placements", "k_ring_pack", "k_ring_unpack",
    "k_mesh_count", "k_mesh_fill", "k_pack_count", "k_pack_fill", "k_unpack", "k_erode_rescue"};

hipEvent_t get_event()
{
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

int device_cus()
{
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int c = cached[dev].load(std::memory_order_relaxed);
    if (c) return c;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 1) return 0;
    cached[dev].store(c, std::memory_order_relaxed);
    return c;
}

void profile_enable(bool on) { std::lock_guard<std::mutex> lk(g_mu); g_prof.store(on, std::memory_order_relaxed); }
bool profile_enabled() { return g_prof.load(std::memory_order_relaxed); }
int profile_num_kernels() { return KID_COUNT; }
const char* profile_kernel_name(int id) { return (id >= 0 && id < KID_COUNT) ? kKernelNames[id] : ""; }

void profile_begin(int kid, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_mu);
    t_open.id = kid; t_open.a = get_event(); t_open.b = get_event();
    (void)hipEventRecord(t_open.a, s);
}

void profile_end(hipStream_t s)
{
    (void)hipEventRecord(t_open.b, s);
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs.push_back(t_open);
}

int profile_collect(double* total_ms, long long* counts)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < KID_COUNT; ++i) { total_ms[i] = 0; counts[i] = 0; }
    for (auto& r : g_recs) {
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e != hipSuccess) return (int)e;
        total_ms[r.id] += ms; counts[r.id] += 1;
        g_pool.push_back(r.a); g_pool.push_back(r.b);
    }
    g_recs.clear();
    return 0;
}

StageRange::StageRange(const char* name) { (void)roctxRangePushA(name); }
StageRange::~StageRange() { (void)roctxRangePop(); }

}  // namespace mmk
