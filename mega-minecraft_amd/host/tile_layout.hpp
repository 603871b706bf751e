// mmgen host side — spatial tiling of a chunk world over the GPUs of a node: the layout / exchange plan (pure host C++, no HIP).
//
// The reference is single-GPU; what crosses a tile border is fixed by its neighbourhoods: erosion padding (terrain.cpp:471-522, recomputed
// locally from RAW layers) and the placement lists of the 3-chunk ring (chunk.cu:1158-1196, exchanged).  This header is the C++ twin
// of mega-minecraft_amd/distributed.py (TileLayout / TileContext): the two sides of a link must agree on which cells travel and in
// which order, whatever language the ranks are written in — tests/test_distributed_cpu.py holds the two plans to each other.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <map>
#include <vector>

namespace mmhost {

struct TileLayout {
    static constexpr int RING = 3;
    int world_cx0, world_cz0, tiles_x, tiles_z, tile_nx, tile_nz;

    int worldSize() const { return tiles_x * tiles_z; }
    std::array<int, 4> region(int rank) const
    {
        const int tx = rank % tiles_x, tz = rank / tiles_x;
        return {world_cx0 + tx * tile_nx, world_cz0 + tz * tile_nz, tile_nx, tile_nz};
    }
    // rank whose tile contains chunk (cx, cz), or -1 outside the world rectangle
    int owner(int cx, int cz) const
    {
        const int x = cx - world_cx0, z = cz - world_cz0;
        if (x < 0 || z < 0 || x >= tiles_x * tile_nx || z >= tiles_z * tile_nz) return -1;
        return (x / tile_nx) + tiles_x * (z / tile_nz);
    }
    int gridW() const { return tile_nx + 2 * RING; }
    int gridH() const { return tile_nz + 2 * RING; }
    int cellOf(int rank, int cx, int cz) const
    {
        const auto r = region(rank);
        return (cx - r[0] + RING) + gridW() * (cz - r[1] + RING);
    }
    template <class F> void forRing(int rank, F&& f) const          // f(cell, cx, cz, owner) over the ring cells only, z-major
    {
        const auto r = region(rank);
        const int w = gridW(), h = gridH();
        for (int z = 0; z < h; ++z)
            for (int x = 0; x < w; ++x) {
                if (x >= RING && x < w - RING && z >= RING && z < h - RING) continue;
                const int cx = r[0] - RING + x, cz = r[1] - RING + z;
                f(x + w * z, cx, cz, owner(cx, cz));
            }
    }
    // mmgen_region_begin's mask: 0 for ring cells that arrive from the peer that owns them, 2 (computed here, lazily: the lists go nowhere
    // else) for ring cells beyond the world's border, 1 for the tile's own cells
    std::vector<uint8_t> localMask(int rank) const
    {
        std::vector<uint8_t> m((size_t)gridW() * gridH(), 1);
        forRing(rank, [&](int cell, int, int, int own) { m[cell] = (own >= 0 && own != rank) ? 0 : 2; });
        return m;
    }
};

// Everything about one rank's tile that does not change from step to step: the peers in ascending order, and per peer the grid cells
// received from it / sent to it, both sorted by (cz, cx) so that the two ends of a link enumerate the same chunks in the same order.
struct ExchangePlan {
    std::vector<int> peers;
    std::vector<int32_t> sendCells, recvCells;        // concatenated over peers
    std::vector<int> sendSeg, recvSeg;                // [peers + 1] boundaries into the two lists

    // loopback: one-rank rehearsal of the transport - the only peer is the rank itself, every ring cell is both sent and received
    // (distributed.py TileContext(loopback=True) is the same plan)
    ExchangePlan(const TileLayout& lay, int rank, bool loopback = false)
    {
        if (loopback) {
            peers.push_back(rank);
            lay.forRing(rank, [&](int cell, int, int, int) { sendCells.push_back(cell); recvCells.push_back(cell); });
            sendSeg = {0, (int)sendCells.size()}; recvSeg = {0, (int)recvCells.size()};
            return;
        }
        struct Key { int cz, cx, cell; bool operator<(const Key& o) const { return cz != o.cz ? cz < o.cz : cx < o.cx; } };
        std::map<int, std::pair<std::vector<Key>, std::vector<Key>>> plan;        // peer -> (recv, send)
        lay.forRing(rank, [&](int cell, int cx, int cz, int own) { if (own >= 0 && own != rank) plan[own].first.push_back({cz, cx, cell}); });
        for (int peer = 0; peer < lay.worldSize(); ++peer) {
            if (peer == rank) continue;
            lay.forRing(peer, [&](int, int cx, int cz, int own) { if (own == rank) plan[peer].second.push_back({cz, cx, lay.cellOf(rank, cx, cz)}); });
        }
        sendSeg.push_back(0); recvSeg.push_back(0);
        for (auto& kv : plan) {
            peers.push_back(kv.first);
            std::sort(kv.second.first.begin(), kv.second.first.end());
            std::sort(kv.second.second.begin(), kv.second.second.end());
            for (const Key& k : kv.second.first) recvCells.push_back(k.cell);
            for (const Key& k : kv.second.second) sendCells.push_back(k.cell);
            sendSeg.push_back((int)sendCells.size()); recvSeg.push_back((int)recvCells.size());
        }
    }
};

}  // namespace mmhost
