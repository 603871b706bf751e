// Prints the exchange plan of every rank of a layout (pure host C++): tests/test_distributed_cpu.py compares it with the plan of
// mega-minecraft_amd/distributed.py.   usage: tile_plan_dump cx0 cz0 tiles_x tiles_z tile_nx tile_nz
#include <cstdio>
#include <cstdlib>
#include "tile_layout.hpp"

int main(int argc, char** argv)
{
    if (argc != 7) { std::fprintf(stderr, "usage: %s cx0 cz0 tiles_x tiles_z tile_nx tile_nz\n", argv[0]); return 2; }
    mmhost::TileLayout lay{std::atoi(argv[1]), std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]), std::atoi(argv[6])};
    for (int rank = 0; rank < lay.worldSize(); ++rank) {
        mmhost::ExchangePlan p(lay, rank);
        const auto mask = lay.localMask(rank);
        int local = 0;
        for (uint8_t m : mask) local += m != 0;
        std::printf("rank %d local %d peers %zu\n", rank, local, p.peers.size());
        for (size_t k = 0; k < p.peers.size(); ++k) {
            std::printf("peer %d recv", p.peers[k]);
            for (int i = p.recvSeg[k]; i < p.recvSeg[k + 1]; ++i) std::printf(" %d", p.recvCells[i]);
            std::printf(" send");
            for (int i = p.sendSeg[k]; i < p.sendSeg[k + 1]; ++i) std::printf(" %d", p.sendCells[i]);
            std::printf("\n");
        }
    }
    return 0;
}
