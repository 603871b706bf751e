// see tiled_world.hpp
#include "tiled_world.hpp"
#include <cstdio>

namespace mmhost {

#define TW_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)
#define TW_MM(expr) do { int e_ = (expr); if (e_) return e_; } while (0)
#define TW_NCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { std::fprintf(stderr, "RCCL: %s\n", ncclGetErrorString(r_)); return 1000 + (int)r_; } } while (0)

TiledWorld::TiledWorld(const TileLayout& lay, int rk, ncclComm_t c, bool loop)
    : layout(lay), rank(rk), comm(c), plan(lay, rk, loop), mask(loop ? std::vector<uint8_t>((size_t)lay.gridW() * lay.gridH(), 1) : lay.localMask(rk)), loopback(loop)
{
    initStatus = (loop && lay.worldSize() != 1) ? (int)hipErrorInvalidValue : init();
}

int TiledWorld::init()
{
    TW_MM(mmgen_region_create(&region));
    TW_HIP(hipStreamCreateWithFlags(&sMain, hipStreamNonBlocking));
    TW_HIP(hipStreamCreateWithFlags(&sComm, hipStreamNonBlocking));
    TW_HIP(hipEventCreateWithFlags(&evPacked, hipEventDisableTiming));
    TW_HIP(hipEventCreateWithFlags(&evArrived, hipEventDisableTiming));
    const size_t ns = plan.sendCells.size(), nr = plan.recvCells.size();
    if (ns) {
        TW_HIP(hipMalloc((void**)&d_sendCells, 4 * ns)); TW_HIP(hipMalloc((void**)&d_hdrS, 8 * ns)); TW_HIP(hipMalloc((void**)&d_offS, 4 * (ns + 1)));
        TW_HIP(hipMemcpy(d_sendCells, plan.sendCells.data(), 4 * ns, hipMemcpyHostToDevice));
    }
    if (nr) {
        TW_HIP(hipMalloc((void**)&d_recvCells, 4 * nr)); TW_HIP(hipMalloc((void**)&d_hdrR, 8 * nr)); TW_HIP(hipMalloc((void**)&d_offR, 4 * (nr + 1)));
        TW_HIP(hipMemcpy(d_recvCells, plan.recvCells.data(), 4 * nr, hipMemcpyHostToDevice));
    }
    return 0;
}

TiledWorld::~TiledWorld()
{
    if (sMain) (void)hipStreamSynchronize(sMain);
    if (sComm) (void)hipStreamSynchronize(sComm);
    for (void* p : {(void*)d_sendCells, (void*)d_recvCells, (void*)d_hdrS, (void*)d_hdrR, (void*)d_offS, (void*)d_offR, (void*)d_payS, (void*)d_payR})
        if (p) (void)hipFree(p);
    if (evPacked) (void)hipEventDestroy(evPacked);
    if (evArrived) (void)hipEventDestroy(evArrived);
    if (sComm) (void)hipStreamDestroy(sComm);
    if (sMain) (void)hipStreamDestroy(sMain);
    if (region) mmgen_region_destroy(region);
}

// Two grouped point-to-point phases with every peer at once (<= 8 peers): list lengths, then the entries that exist.
int TiledWorld::exchange(uint8_t* d_blocks)
{
    mmgen_feature_placement* fp; mmgen_cave_feature_placement* cfp; int32_t* counts;
    TW_MM(mmgen_region_placement_buffers(region, &fp, &cfp, &counts, nullptr, nullptr, nullptr, nullptr));
    const int ns = (int)plan.sendCells.size(), nr = (int)plan.recvCells.size(), np = (int)plan.peers.size();
    TW_MM(mmgen_ring_header(counts, d_sendCells, ns, d_hdrS, sMain));
    TW_NCCL(ncclGroupStart());
    for (int k = 0; k < np; ++k) {
        const int a = plan.sendSeg[k], b = plan.sendSeg[k + 1], c = plan.recvSeg[k], d = plan.recvSeg[k + 1];
        if (b > a) TW_NCCL(ncclSend(d_hdrS + 2 * a, 2 * (size_t)(b - a), ncclInt32, plan.peers[k], comm, sMain));
        if (d > c) TW_NCCL(ncclRecv(d_hdrR + 2 * c, 2 * (size_t)(d - c), ncclInt32, plan.peers[k], comm, sMain));
    }
    TW_NCCL(ncclGroupEnd());
    TW_MM(mmgen_ring_offsets(d_hdrS, ns, d_offS, sMain));
    TW_MM(mmgen_ring_offsets(d_hdrR, nr, d_offR, sMain));
    // the one host read of the step: message boundaries in words
    std::vector<int32_t> offS(ns + 1), offR(nr + 1);
    TW_HIP(hipMemcpyAsync(offS.data(), d_offS, 4 * (size_t)(ns + 1), hipMemcpyDeviceToHost, sMain));
    TW_HIP(hipMemcpyAsync(offR.data(), d_offR, 4 * (size_t)(nr + 1), hipMemcpyDeviceToHost, sMain));
    TW_HIP(hipStreamSynchronize(sMain));
    const size_t totS = (size_t)offS[ns], totR = (size_t)offR[nr];
    if (totS > payCapS) { if (d_payS) TW_HIP(hipFree(d_payS)); payCapS = totS + totS / 4 + 1024; TW_HIP(hipMalloc((void**)&d_payS, 4 * payCapS)); }
    if (totR > payCapR) { if (d_payR) TW_HIP(hipFree(d_payR)); payCapR = totR + totR / 4 + 1024; TW_HIP(hipMalloc((void**)&d_payR, 4 * payCapR)); }
    if (totS) TW_MM(mmgen_ring_pack(fp, cfp, d_sendCells, d_hdrS, d_offS, ns, d_payS, sMain));
    if (loopback) {
        // the packed payload is on its way: wipe the ring's list lengths in the placement grid (top / bottom 3 rows, left / right 3
        // columns of the rows between), only the wire can restore them
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
        const size_t a = (size_t)offS[plan.sendSeg[k]], b = (size_t)offS[plan.sendSeg[k + 1]];
        const size_t c = (size_t)offR[plan.recvSeg[k]], d = (size_t)offR[plan.recvSeg[k + 1]];
        if (b > a) TW_NCCL(ncclSend(d_payS + a, b - a, ncclInt32, plan.peers[k], comm, sComm));
        if (d > c) TW_NCCL(ncclRecv(d_payR + c, d - c, ncclInt32, plan.peers[k], comm, sComm));
    }
    TW_NCCL(ncclGroupEnd());
    TW_HIP(hipEventRecord(evArrived, sComm));
    // the base fill needs nothing from the ring: it runs while the payload travels
    TW_MM(mmgen_region_fill(region, d_blocks, sMain));
    TW_HIP(hipStreamWaitEvent(sMain, evArrived, 0));
    if (totR) TW_MM(mmgen_ring_unpack(d_payR, d_hdrR, d_offR, d_recvCells, nr, fp, cfp, counts, sMain));
    haloBytes = 8 * (size_t)nr + 4 * totR;
    return 0;
}

int TiledWorld::generate(unsigned flags, uint8_t* d_blocks, float* d_heightfields)
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
    TW_HIP(hipStreamSynchronize(sMain));
    return 0;
}

}  // namespace mmhost
