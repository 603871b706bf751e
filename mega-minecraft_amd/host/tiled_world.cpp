// see tiled_world.hpp
#include "tiled_world.hpp"
#include <cstdio>

namespace mmhost {

#define TW_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)
#define TW_MM(expr) do { int e_ = (expr); if (e_) return e_; } while (0)
#define TW_NCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { std::fprintf(stderr, "RCCL: %s\n", ncclGetErrorString(r_)); return 1000 + (int)r_; } } while (0)

TiledWorld::TiledWorld(const TileLayout& lay, int rk, ncclComm_t c, bool loop, int wpc)
    : layout(lay), rank(rk), comm(c), plan(lay, rk, loop), mask(loop ? std::vector<uint8_t>((size_t)lay.gridW() * lay.gridH(), 1) : lay.localMask(rk)),
      wordsPerCell(wpc), loopback(loop)
{
    initStatus = ((loop && lay.worldSize() != 1) || wpc < 1) ? (int)hipErrorInvalidValue : init();
}

// message layout of include/mmgen.h mmgen_ring_pack_messages (distributed.py::message_layout): per peer [2 n lengths][n * wordsPerCell]
static void message_layout(const std::vector<int>& seg, int wordsPerCell, std::vector<size_t>& bounds, std::vector<int32_t>& slots)
{
    bounds.assign(1, 0);
    slots.clear();
    for (size_t k = 0; k + 1 < seg.size(); ++k) {
        const int a = seg[k], n = seg[k + 1] - seg[k];
        const size_t base = bounds.back();
        for (int i = 0; i < n; ++i) {
            slots.push_back((int32_t)(base + 2 * (size_t)i)); slots.push_back((int32_t)(base + 2 * (size_t)n));
            slots.push_back(a); slots.push_back(wordsPerCell * n);
        }
        bounds.push_back(base + 2 * (size_t)n + (size_t)wordsPerCell * n);
    }
}

int TiledWorld::init()
{
    TW_MM(mmgen_region_create(&region));
    TW_HIP(hipStreamCreateWithFlags(&sMain, hipStreamNonBlocking));
    TW_HIP(hipStreamCreateWithFlags(&sComm, hipStreamNonBlocking));
    TW_HIP(hipEventCreateWithFlags(&evPacked, hipEventDisableTiming));
    TW_HIP(hipEventCreateWithFlags(&evArrived, hipEventDisableTiming));
    TW_HIP(hipMalloc((void**)&d_overflow, 4));
    TW_HIP(hipMemset(d_overflow, 0, 4));
    std::vector<int32_t> slots;
    auto side = [&](const std::vector<int>& cells, const std::vector<int>& seg, std::vector<size_t>& bounds, int32_t*& d_cells, int32_t*& d_slots, int32_t*& d_scratch,
                    int32_t*& d_msg) -> int {
        const size_t n = cells.size();
        message_layout(seg, wordsPerCell, bounds, slots);
        if (!n) return 0;
        if (bounds.back() >= (size_t)1 << 31) return (int)hipErrorInvalidValue;      // word indices travel as int32
        TW_HIP(hipMalloc((void**)&d_cells, 4 * n)); TW_HIP(hipMalloc((void**)&d_slots, 16 * n)); TW_HIP(hipMalloc((void**)&d_scratch, 4 * (3 * n + 1)));
        TW_HIP(hipMalloc((void**)&d_msg, 4 * bounds.back()));
        TW_HIP(hipMemcpy(d_cells, cells.data(), 4 * n, hipMemcpyHostToDevice));
        TW_HIP(hipMemcpy(d_slots, slots.data(), 16 * n, hipMemcpyHostToDevice));
        return 0;
    };
    TW_MM(side(plan.sendCells, plan.sendSeg, msgS, d_sendCells, d_sendSlots, d_scratchS, d_msgS));
    TW_MM(side(plan.recvCells, plan.recvSeg, msgR, d_recvCells, d_recvSlots, d_scratchR, d_msgR));
    return 0;
}

TiledWorld::~TiledWorld()
{
    if (sMain) (void)hipStreamSynchronize(sMain);
    if (sComm) (void)hipStreamSynchronize(sComm);
    for (void* p : {(void*)d_sendCells, (void*)d_recvCells, (void*)d_sendSlots, (void*)d_recvSlots, (void*)d_scratchS, (void*)d_scratchR, (void*)d_msgS, (void*)d_msgR,
                    (void*)d_overflow})
        if (p) (void)hipFree(p);
    if (evPacked) (void)hipEventDestroy(evPacked);
    if (evArrived) (void)hipEventDestroy(evArrived);
    if (sComm) (void)hipStreamDestroy(sComm);
    if (sMain) (void)hipStreamDestroy(sMain);
    if (region) mmgen_region_destroy(region);
}

// One grouped point-to-point phase with every peer at once (<= 8 peers): fixed-size messages, list lengths in-band.  Nothing here reads
// device memory on the host or synchronises a stream.
int TiledWorld::exchange(uint8_t* d_blocks)
{
    mmgen_feature_placement* fp; mmgen_cave_feature_placement* cfp; int32_t* counts;
    TW_MM(mmgen_region_placement_buffers(region, &fp, &cfp, &counts, nullptr, nullptr, nullptr, nullptr));
    const int ns = (int)plan.sendCells.size(), nr = (int)plan.recvCells.size(), np = (int)plan.peers.size();
    TW_MM(mmgen_ring_pack_messages(fp, cfp, counts, d_sendCells, d_sendSlots, ns, d_scratchS, d_msgS, d_overflow, sMain));
    if (loopback) {
        // the messages are packed: wipe the ring's list lengths in the placement grid (top / bottom 3 rows, left / right 3 columns of the
        // rows between), only the wire can restore them
        const int w = layout.gridW(), h = layout.gridH(), R = TileLayout::RING;
        TW_HIP(hipMemsetAsync(counts, 0, 8 * (size_t)w * R, sMain));
        TW_HIP(hipMemsetAsync(counts + 2 * (size_t)w * (h - R), 0, 8 * (size_t)w * R, sMain));
        TW_HIP(hipMemset2DAsync(counts + 2 * (size_t)w * R, 8 * (size_t)w, 0, 8 * (size_t)R, (size_t)(h - 2 * R), sMain));
        TW_HIP(hipMemset2DAsync(counts + 2 * ((size_t)w * R + (w - R)), 8 * (size_t)w, 0, 8 * (size_t)R, (size_t)(h - 2 * R), sMain));
    }
    TW_HIP(hipEventRecord(evPacked, sMain));
    TW_HIP(hipStreamWaitEvent(sComm, evPacked, 0));
    TW_NCCL(ncclGroupStart());
    for (int k = 0; k < np; ++k) {
        if (msgS[k + 1] > msgS[k]) TW_NCCL(ncclSend(d_msgS + msgS[k], msgS[k + 1] - msgS[k], ncclInt32, plan.peers[k], comm, sComm));
        if (msgR[k + 1] > msgR[k]) TW_NCCL(ncclRecv(d_msgR + msgR[k], msgR[k + 1] - msgR[k], ncclInt32, plan.peers[k], comm, sComm));
    }
    TW_NCCL(ncclGroupEnd());
    TW_HIP(hipEventRecord(evArrived, sComm));
    // the base fill needs nothing from the ring: begin has issued it on the region's own stream (mmgen_region_set_output), this is a no-op then
    TW_MM(mmgen_region_fill(region, d_blocks, sMain));
    TW_HIP(hipStreamWaitEvent(sMain, evArrived, 0));
    TW_MM(mmgen_ring_unpack_messages(d_msgR, d_recvCells, d_recvSlots, nr, d_scratchR, fp, cfp, counts, d_overflow, sMain));
    haloBytes = 4 * msgR.back();
    return 0;
}

int TiledWorld::generateAsync(unsigned flags, uint8_t* d_blocks, float* d_heightfields)
{
    const auto r = layout.region(rank);
    if (initStatus) return initStatus;
    const bool exch = (flags & MMGEN_REGION_FEATURES) && (layout.worldSize() > 1 || loopback) && !plan.peers.empty();
    haloBytes = 0;
    TW_MM(mmgen_region_set_output(region, d_blocks));       // the base fill starts as soon as the caves' extents and the eroded layers exist
    TW_MM(mmgen_region_begin(region, r[0], r[1], r[2], r[3], flags, (flags & MMGEN_REGION_FEATURES) ? mask.data() : nullptr, sMain));
    if (exch) {
        if (!comm) return (int)hipErrorInvalidValue;
        TW_MM(exchange(d_blocks));
    }
    TW_MM(mmgen_region_finish(region, d_blocks, d_heightfields, nullptr, nullptr, sMain));
    return 0;
}

int TiledWorld::finishStep()
{
    int32_t need = 0;
    // the verdict is the MAXIMUM over the ranks: an oversized message is only seen by its sender and its receiver, and a rank that returned
    // kRingOverflow alone would leave the others waiting in their next exchange (collective: every rank calls finishStep, or none)
    if (comm && layout.worldSize() > 1 && !loopback) {
        const ncclResult_t nr = ncclAllReduce(d_overflow, d_overflow, 1, ncclInt32, ncclMax, comm, sMain);
        if (nr != ncclSuccess) return 1000 + (int)nr;
    }
    TW_HIP(hipMemcpyAsync(&need, d_overflow, 4, hipMemcpyDeviceToHost, sMain));
    TW_HIP(hipStreamSynchronize(sMain));
    if (need) {
        TW_HIP(hipMemset(d_overflow, 0, 4));
        std::fprintf(stderr, "mmgen: a ring message needed %d payload words, the budget is %d per cell on average: construct TiledWorld with a larger wordsPerCell\n",
                     need, wordsPerCell);
        return kRingOverflow;
    }
    return 0;
}

int TiledWorld::generate(unsigned flags, uint8_t* d_blocks, float* d_heightfields)
{
    TW_MM(generateAsync(flags, d_blocks, d_heightfields));
    return finishStep();
}

}  // namespace mmhost
