// mmgen host side — one rank's share of a spatially tiled world, C++ over the C ABI + RCCL (one process per GPU).
//
// north_star: "Host code stays C++ calling HIP through a thin C-ABI ... the world is tiled spatially across the 8 GPUs of one node with
// RCCL halo exchange over xGMI".  This is the C++ twin of mega-minecraft_amd/distributed.py::generate_tile: same layout, same wire
// protocol (include/mmgen.h mmgen_ring_*: headers, then the entries that exist), same overlap (the base fill of the tile runs while the
// payload is in flight on a second stream), no collective on the data path.  The reference itself has no multi-GPU path; the
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
    TiledWorld(const TileLayout& layout, int rank, ncclComm_t comm, bool loopback = false);
    // 0, or the first error of the constructor (allocation, stream / event / region creation): check before generate()
    int status() const { return initStatus; }
    ~TiledWorld();
    TiledWorld(const TiledWorld&) = delete;
    TiledWorld& operator=(const TiledWorld&) = delete;

    // Generates this rank's tile through all stages selected by `flags` (MMGEN_REGION_*): d_blocks [tile_nx * tile_nz][98304],
    // d_heightfields [tile_nx * tile_nz][256] (nullable), z-major.  Synchronous on return.  Returns 0 or a hipError_t / 1000 + ncclResult_t.
    int generate(unsigned flags, uint8_t* d_blocks, float* d_heightfields);
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
    int32_t *d_sendCells = nullptr, *d_recvCells = nullptr, *d_hdrS = nullptr, *d_hdrR = nullptr, *d_offS = nullptr, *d_offR = nullptr;
    int32_t *d_payS = nullptr, *d_payR = nullptr;
    size_t payCapS = 0, payCapR = 0, haloBytes = 0;
    bool loopback = false;
    int initStatus = 0;
    int init();
    int exchange(uint8_t* d_blocks);
};

}  // namespace mmhost
