// One line per drawable chunk, written by every headless scheduler main (mmgen_terrain_demo, mmgen_region_terrain_demo and
// tests/refdrop's ref_terrain_dropin) so that tests can hold the chunks a scheduler produced to the CPU oracle:
//     cx cz digest(blocks) vertexCount digest(vertex bytes) digest(index bytes)
// digest = sum over the buffer's 8-byte words w of word * (2 w + 1) * 0x9E3779B97F4A7C15 (mod 2^64) - for the blocks this is
// mega-minecraft_amd.distributed.chunk_digests, the value tests/golden/world_digests.npz holds for the oracle's chunks.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>

inline uint64_t mmhostDigest(const void* p, size_t bytes)          // bytes: a multiple of 8 (98 304 blocks; 40 B vertices; 24 B per quad of indices)
{
    uint64_t s = 0;
    for (uint64_t w = 0; w < bytes / 8; ++w) {
        uint64_t v;
        std::memcpy(&v, (const char*)p + 8 * w, 8);
        s += v * ((2 * w + 1) * 0x9E3779B97F4A7C15ull);
    }
    return s;
}

template <class ChunkT>
inline void mmhostWriteChunkDigest(FILE* f, const ChunkT* c)
{
    std::fprintf(f, "%d %d %016llx %zu %016llx %016llx\n", c->worldChunkPos.x, c->worldChunkPos.y, (unsigned long long)mmhostDigest(c->blocks.data(), c->blocks.size()),
                 c->verts.size(), (unsigned long long)mmhostDigest(c->verts.data(), c->verts.size() * sizeof(c->verts[0])),
                 (unsigned long long)mmhostDigest(c->idx.data(), c->idx.size() * sizeof(c->idx[0])));
}
