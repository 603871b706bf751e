// CPU sanitizer job (ASan + UBSan; GPU sanitizers are not available on this pool): `make -C oracle asan`.
//   1. the oracle's full region pipeline on a small region (every stage, erosion of a zone, features, decorators) under the sanitizers;
//   2. the host logic of the product that needs no device: the wire-format decoder mmgen_unpack_chunk_host fed with a valid stream built
//      from the oracle's blocks, then with 20 000 truncated / bit-flipped / random variants (must return -1 or 0, never touch memory
//      outside `blocks`), and the tile layout / exchange plan of the C++ multi-GPU host on several layouts.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/mmgen.h"
#include "../mega-minecraft_amd/host/tile_layout.hpp"

extern "C" void mmo_generate_region(int cx0, int cz0, int nx, int nz, int flags, uint8_t* out_blocks, float* out_hf, float* out_layers, void* out_cave,
                                    int nthreads, double* stage_seconds);

static std::vector<uint8_t> pack_chunk(const uint8_t* blocks)          // numpy-free statement of the format: u16 runs[256], then (id, len - 1) pairs
{
    std::vector<uint8_t> head(512), body;
    for (int col = 0; col < 256; ++col) {
        unsigned runs = 0;
        for (int y = 0; y < 384;) {
            const uint8_t id = blocks[384 * col + y];
            int len = 1;
            while (y + len < 384 && len < 256 && blocks[384 * col + y + len] == id) ++len;
            body.push_back(id); body.push_back((uint8_t)(len - 1));
            y += len; ++runs;
        }
        head[2 * col] = (uint8_t)(runs & 255); head[2 * col + 1] = (uint8_t)(runs >> 8);
    }
    head.insert(head.end(), body.begin(), body.end());
    return head;
}

int main()
{
    std::vector<uint8_t> blocks((size_t)2 * 98304);
    std::vector<float> hf(2 * 256), layers(2 * 5120);
    std::vector<uint8_t> cave((size_t)2 * 8192 * 12);
    mmo_generate_region(1488, -1110, 2, 1, 7, blocks.data(), hf.data(), layers.data(), cave.data(), 4, nullptr);
    std::printf("oracle region under sanitizers: ok (first column top block %d)\n", blocks[383]);

    unsigned rng = 12345u;
    auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    int ok = 0, rejected = 0;
    for (int c = 0; c < 2; ++c) {
        const std::vector<uint8_t> good = pack_chunk(blocks.data() + (size_t)98304 * c);
        std::vector<uint8_t> out(98304);
        if (mmgen_unpack_chunk_host(good.data(), good.size(), out.data()) != 0 || std::memcmp(out.data(), blocks.data() + (size_t)98304 * c, 98304)) {
            std::fprintf(stderr, "round trip of a valid stream failed\n");
            return 1;
        }
        for (int it = 0; it < 10000; ++it) {
            std::vector<uint8_t> bad = good;
            const unsigned kind = next() % 4;
            if (kind == 0) bad.resize(next() % (bad.size() + 1));                                   // truncated
            else if (kind == 1) for (int k = 0; k < 1 + (int)(next() % 8); ++k) bad[next() % bad.size()] ^= (uint8_t)(1u << (next() % 8));
            else if (kind == 2) { bad.resize(512 + next() % 4096); for (auto& b : bad) b = (uint8_t)next(); }   // noise
            else { const size_t extra = next() % 64; bad.resize(bad.size() + extra, 7); }            // trailing bytes
            // exact-size heap copy: the sanitizer sees any read past the end
            uint8_t* heap = (uint8_t*)std::malloc(bad.size() ? bad.size() : 1);
            std::memcpy(heap, bad.data(), bad.size());
            const int rc = mmgen_unpack_chunk_host(heap, bad.size(), out.data());
            std::free(heap);
            if (rc == 0) ++ok; else if (rc == -1) ++rejected; else { std::fprintf(stderr, "unexpected return %d\n", rc); return 1; }
        }
    }
    std::printf("unpack fuzz: %d accepted, %d rejected, no memory error\n", ok, rejected);

    for (const mmhost::TileLayout lay : {mmhost::TileLayout{-5, 7, 2, 2, 4, 5}, mmhost::TileLayout{-128, -128, 4, 2, 64, 128}, mmhost::TileLayout{0, 0, 3, 3, 1, 2}}) {
        size_t cells = 0;
        for (int r = 0; r < lay.worldSize(); ++r) { mmhost::ExchangePlan p(lay, r); cells += p.sendCells.size() + p.recvCells.size(); (void)lay.localMask(r); }
        std::printf("exchange plan %dx%d tiles of %dx%d: %zu cell transfers\n", lay.tiles_x, lay.tiles_z, lay.tile_nx, lay.tile_nz, cells);
    }
    return 0;
}
