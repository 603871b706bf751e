// mmgen region wire format for gfx950 ("none exists in the reference; a compact block format makes the D2H of
// 96 KB per chunk the next bottleneck to remove").  Terrain columns are long vertical runs of one block id, so a chunk is stored
// column by column (the reference's blocks[] order: column = x + 16 z, y ascending) as run-length pairs:
//
//   chunk  :=  u16 runsOfColumn[256]   then, for column 0 .. 255 in order, its runs
//   run    :=  u8 blockId, u8 length - 1              (1 .. 256 voxels; a longer run continues in the next pair)
//
// A chunk packs to 512 + 2 R bytes (R = total runs, typically 2 - 4 k: 5 - 9 KB instead of 98 304).  Chunks are byte-addressed by a
// caller-computed exclusive prefix of their sizes, like the mesher's vertex offsets.
//   k_pack_count   lane = column: runs per column -> d_col_runs[n][256], packed bytes per chunk -> d_chunk_bytes[n]
//   k_pack_fill    LDS scan of the 256 run counts, every lane writes its column's pairs at its own offset
//   k_unpack       inverse: lane = column, expands its pairs back into 384 bytes
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mmgen.h"
#include "mmgen_prof.h"

namespace mm {

// walks one column; EMIT: writes (id, len - 1) pairs to out; returns the number of pairs
template <bool EMIT>
static __device__ __forceinline__ uint32_t pack_column(const uint8_t* __restrict__ col, uint8_t* __restrict__ out)
{
    uint32_t runs = 0;
    int cur = -1, len = 0;
    for (int w = 0; w < 24; ++w) {
        const uint4 v = ((const uint4*)col)[w];
        const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int id = (int)((q[k] >> (8 * b)) & 255u);
                if (id == cur && len < 256) { ++len; continue; }
                if (len > 0) {
                    if (EMIT) { out[2 * runs] = (uint8_t)cur; out[2 * runs + 1] = (uint8_t)(len - 1); }
                    ++runs;
                }
                cur = id; len = 1;
            }
        }
    }
    if (EMIT) { out[2 * runs] = (uint8_t)cur; out[2 * runs + 1] = (uint8_t)(len - 1); }
    return runs + 1;
}

__global__ void __launch_bounds__(256)
k_pack_count(const uint8_t* __restrict__ blocks, const int32_t* __restrict__ chunkIdx, uint16_t* __restrict__ colRuns, uint32_t* __restrict__ chunkBytes)
{
    __shared__ uint32_t s_total;
    const int o = blockIdx.x, t = threadIdx.x;
    const int c = chunkIdx ? chunkIdx[o] : o;
    if (t == 0) s_total = 0;
    __syncthreads();
    const uint32_t r = pack_column<false>(blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * c + 384 * t, nullptr);
    colRuns[256 * o + t] = (uint16_t)r;
    atomicAdd(&s_total, r);
    __syncthreads();
    if (t == 0) chunkBytes[o] = 512u + 2u * s_total;
}

__global__ void __launch_bounds__(256)
k_pack_fill(const uint8_t* __restrict__ blocks, const int32_t* __restrict__ chunkIdx, const uint16_t* __restrict__ colRuns,
            const uint64_t* __restrict__ chunkOffset, uint8_t* __restrict__ out)
{
    __shared__ uint32_t s_scan[256];
    const int o = blockIdx.x, t = threadIdx.x;
    const int c = chunkIdx ? chunkIdx[o] : o;
    const uint32_t mine = colRuns[256 * o + t];
    s_scan[t] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const uint32_t v = t >= off ? s_scan[t - off] : 0u;
        __syncthreads();
        s_scan[t] += v;
        __syncthreads();
    }
    uint8_t* dst = out + chunkOffset[o];
    ((uint16_t*)dst)[t] = (uint16_t)mine;                 // chunk offsets are even (512 + 2 R): the header is 2-byte aligned
    pack_column<true>(blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * c + 384 * t, dst + 512 + 2 * (size_t)(s_scan[t] - mine));
}

__global__ void __launch_bounds__(256)
k_unpack(const uint8_t* __restrict__ packed, const uint64_t* __restrict__ chunkOffset, uint8_t* __restrict__ blocks)
{
    __shared__ uint32_t s_scan[256];
    const int c = blockIdx.x, t = threadIdx.x;
    const uint8_t* src = packed + chunkOffset[c];
    const uint32_t mine = ((const uint16_t*)src)[t];
    s_scan[t] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const uint32_t v = t >= off ? s_scan[t - off] : 0u;
        __syncthreads();
        s_scan[t] += v;
        __syncthreads();
    }
    const uint8_t* runs = src + 512 + 2 * (size_t)(s_scan[t] - mine);
    uint8_t* col = blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * c + 384 * t;
    int y = 0;
    for (uint32_t r = 0; r < mine && y < 384; ++r) {
        const uint8_t id = runs[2 * r];
        const int len = (int)runs[2 * r + 1] + 1;
        for (int k = 0; k < len && y < 384; ++k) col[y++] = id;
    }
    for (; y < 384; ++y) col[y] = 0;                      // malformed input: never read out of bounds, never leave bytes unset
}

}  // namespace mm

extern "C" {

int mmgen_pack_count(const uint8_t* d_blocks, const int32_t* d_chunk_idx, int n, uint16_t* d_col_runs, uint32_t* d_chunk_bytes, void* stream)
{
    if (n < 0 || (n > 0 && (!d_blocks || !d_col_runs || !d_chunk_bytes))) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    MMK_LAUNCH_NORET(mmk::KID_PACK_COUNT, mm::k_pack_count, dim3(n), dim3(256), (hipStream_t)stream, d_blocks, d_chunk_idx, d_col_runs, d_chunk_bytes);
    return (int)hipGetLastError();
}

int mmgen_pack_fill(const uint8_t* d_blocks, const int32_t* d_chunk_idx, int n, const uint16_t* d_col_runs, const uint64_t* d_chunk_offset, uint8_t* d_out,
                    void* stream)
{
    if (n < 0 || (n > 0 && (!d_blocks || !d_col_runs || !d_chunk_offset || !d_out))) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    MMK_LAUNCH_NORET(mmk::KID_PACK_FILL, mm::k_pack_fill, dim3(n), dim3(256), (hipStream_t)stream, d_blocks, d_chunk_idx, d_col_runs, d_chunk_offset, d_out);
    return (int)hipGetLastError();
}

int mmgen_unpack(const uint8_t* d_packed, const uint64_t* d_chunk_offset, int n, uint8_t* d_blocks, void* stream)
{
    if (n < 0 || (n > 0 && (!d_packed || !d_chunk_offset || !d_blocks))) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    MMK_LAUNCH_NORET(mmk::KID_UNPACK, mm::k_unpack, dim3(n), dim3(256), (hipStream_t)stream, d_packed, d_chunk_offset, d_blocks);
    return (int)hipGetLastError();
}

}  // extern "C"
