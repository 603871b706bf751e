// mmgen host side — one rank's share of a spatially tiled world, C++ over the C ABI + RCCL (one process per GPU).
//
// north_star: "Host code stays C++ calling HIP through a thin C-ABI ... the world is tiled spatially across the 8 GPUs of one node with
// RCCL halo exchange over xGMI".  This is the C++ twin of mega-minecraft_amd/distributed.py::generate_tile: same layout, same wire
// protocol (include/mmgen.h mmgen_ring_pack_messages: one fixed-size message per peer, the cells' list lengths in-band, then the entries that
// exist - no host read and no stream synchronisation inside the step), same overlap (the base fill of the tile, issued by
// mmgen_region_begin, runs while the messages are in flight on a second stream), no collective on the data path.  The reference itself has no multi-GPU path; the
// neighbourhoods that define what must travel are terrain.cpp:471-522 (erosion padding: recomputed locally) and chunk.cu:1158-1196
// (placement lists of the 3-chunk ring: exchanged).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdint>
#include <vector>
#include "../../include/mmgen.h"
#include "tile_layout.hpp"

namespace mmhost {

class TiledWorld {
public:
    // comm may be null when layout.worldSize() == 1.  The communicator and the device are the caller's (one rank = one GPU).
    // loopback (single tile, a communicator of one rank): every ring cell is computed here in full AND shipped rank -> rank through the
    // real exchange (pack, grouped ncclSend / ncclRecv to self on the two streams, unpack) after its local copy has been wiped - the
    // rehearsal of the transport on a box with one GPU; the tile must equal mmgen_region_generate's.
    // wordsPerCell: payload budget of a ring message per cell on average (a typical cell carries 250 - 750 words, 7 424 can never overflow)
    TiledWorld(const TileLayout& layout, int rank, ncclComm_t comm, bool loopback = false, int wordsPerCell = 2048);
    // 0, or the first error of the constructor (allocation, stream / event / region creation): check before generate()
    int status() const { return initStatus; }
    ~TiledWorld();
    TiledWorld(const TiledWorld&) = delete;
    TiledWorld& operator=(const TiledWorld&) = delete;

    // Generates this rank's tile through all stages selected by `flags` (MMGEN_REGION_*): d_blocks [tile_nx * tile_nz][98304],
    // d_heightfields [tile_nx * tile_nz][256] (nullable), z-major.  Synchronous on return.  Returns 0, a hipError_t, 1000 + ncclResult_t, or
    // kRingOverflow when a ring message did not fit its budget (the tile is then incomplete: construct with a larger wordsPerCell).
    int generate(unsigned flags, uint8_t* d_blocks, float* d_heightfields);
    // the same without the final synchronisation and overflow check: everything is enqueued on stream() and the caller pipelines steps;
    // call finishStep() before trusting (or re-using) the outputs.  finishStep is collective over the communicator (the ring-overflow verdict
    // is agreed with one 4-byte all-reduce, so that every rank returns kRingOverflow in the same call): all ranks call it the same number of times
    int generateAsync(unsigned flags, uint8_t* d_blocks, float* d_heightfields);
    int finishStep();
    hipStream_t stream() const { return sMain; }
    static constexpr int kRingOverflow = 2000;
    size_t lastHaloBytesReceived() const { return haloBytes; }

private:
    TileLayout layout;
    int rank;
    ncclComm_t comm;
    ExchangePlan plan;
    std::vector<uint8_t> mask;
    mmgen_region* region = nullptr;
    hipStream_t sMain = nullptr, sComm = nullptr;
    hipEvent_t evPacked = nullptr, evArrived = nullptr;
    int32_t *d_sendCells = nullptr, *d_recvCells = nullptr, *d_sendSlots = nullptr, *d_recvSlots = nullptr, *d_scratchS = nullptr, *d_scratchR = nullptr;
    int32_t *d_msgS = nullptr, *d_msgR = nullptr, *d_overflow = nullptr;
    std::vector<size_t> msgS, msgR;            // message boundaries in words, per peer
    int wordsPerCell;
    size_t haloBytes = 0;
    bool loopback = false;
    int initStatus = 0;
    int init();
    int exchange(uint8_t* d_blocks);
};

}  // namespace mmhost
