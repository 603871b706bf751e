// mmgen mesher for gfx950: the mesh build that follows the generation path.
// Behavioural spec: Chunk::createVBOs (chunk.cu:1751-2003): per voxel in z, x, y order an X-shaped block emits 8 vertices +
// 12 indices, a cube emits 4 vertices + 6 indices per face whose neighbour lets it show, in DirectionEnums::dirVecs order;
// Vertex = {vec3 pos, vec3 nor, vec2 uv, size_t material} (rendering/structs.hpp:25-31), indices local to the chunk.
//
// The reference walks 98 304 voxels x 6 neighbours per chunk on the host with std::vector pushes (the most expensive action of
// its scheduler).  Here: one workgroup per chunk, one lane per column (the reference's z, x order IS the column index), two
// kernels sharing one traversal template:
//   k_mesh_count  vertices per column -> columnVerts[n][256], per chunk -> chunkVerts[n]   (indices = 3/2 vertices, always)
//   k_mesh_fill   scan of the 256 column counts in LDS; every lane appends its column's QUAD RECORDS (32 bits: y, face, column,
//                 tile, uv rotation / flip) to an LDS stage at its scanned offset - the reference's order by construction, no
//                 atomics, no sorting - and then all lanes expand the records into the vertex and index streams, one vertex /
//                 one index per lane: consecutive lanes write consecutive 40-byte vertices, a wave covers 2.5 KB of contiguous
//                 output per pass instead of 64 scattered 160-byte runs (the mesher is HBM-bound: 0.67 MB written per chunk).
// Columns are read as 16-byte words (24 per column) together with the 4 neighbouring columns' words; the per-block render data
// (140 packed words, mm_blockdata.cuh) sits in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mm_math.cuh"
#include "mm_noise.cuh"
#include "mm_blockdata.cuh"
#include "../../include/mmgen.h"

namespace mm {

#define MESH_TRANS(d) ((d) >> 30)
enum { T_OPAQUE = 0, T_SEMI = 1, T_TRANSPARENT = 2, T_XSHAPED = 3 };

MM_DEV int mesh_material(int b)      // switch of chunk.cu:1797-1829
{
    switch (b) {
    case MMB_WATER: return 1;
    case MMB_CYAN_CRYSTAL: case MMB_GREEN_CRYSTAL: case MMB_MAGENTA_CRYSTAL: return 2;
    case MMB_MARBLE: case MMB_QUARTZ: case MMB_ICE: case MMB_PACKED_ICE: case MMB_BLUE_ICE: return 3;
    case MMB_SNOW: case MMB_SNOWY_GRASS_BLOCK: return 4;
    case MMB_SAND: case MMB_GRAVEL: return 5;
    default: return 0;
    }
}

__device__ constexpr int kMeshDir[6][3] = {{0, 0, 1}, {1, 0, 0}, {0, 0, -1}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}};      // enums.hpp:43-50
__device__ constexpr int kMeshDirVert[24][3] = {      // directionVertPositions, chunk.cu:1768-1775
    {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}, {1, 0, 1}, {1, 0, 0}, {1, 1, 0}, {1, 1, 1}, {1, 0, 0}, {0, 0, 0}, {0, 1, 0}, {1, 1, 0},
    {0, 0, 0}, {0, 0, 1}, {0, 1, 1}, {0, 1, 0}, {0, 1, 1}, {1, 1, 1}, {1, 1, 0}, {0, 1, 0}, {0, 0, 0}, {1, 0, 0}, {1, 0, 1}, {0, 0, 1}};

// byte i of a 16-byte word without indexing registers dynamically
MM_DEV int byte_at(const uint4& v, int i)
{
    const uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    return (int)(((i < 8 ? lo : hi) >> (8 * (i & 7))) & 255u);
}

// Quad record (one displayed cube face, or one of the two quads of an X-shaped block), 32 bits:
//   y 0-8 | kind 9-11 (0-5 = face direction, 6 / 7 = X quad 1 / 2) | column 12-19 | tile u 20-23 | tile v 24-27 |
//   uv start 28-29 | flip u 30 | flip v 31            (material is recomputed from the block id kept in a parallel byte array)
MM_DEV uint32_t quad_record(int y, int kind, int column, int su, int sv, int uvStart, int uvFlip)
{
    const int fu = (uvFlip != -1) && (uvFlip & 1), fv = (uvFlip != -1) && (uvFlip & 2);
    return (uint32_t)y | ((uint32_t)kind << 9) | ((uint32_t)column << 12) | ((uint32_t)su << 20) | ((uint32_t)sv << 24) | ((uint32_t)(uvStart & 3) << 28) |
           ((uint32_t)fu << 30) | ((uint32_t)fv << 31);
}

// One lane = one column.  EMIT = false: returns the column's quad count.  EMIT = true: appends the column's quad records to
// recs / mats (LDS) in the reference's order (y ascending, faces in dirVecs order).
template <bool EMIT>
MM_DEV uint32_t mesh_column(const uint8_t* __restrict__ col, const uint8_t* __restrict__ colN /*+z*/, const uint8_t* __restrict__ colE /*+x*/,
                            const uint8_t* __restrict__ colS /*-z*/, const uint8_t* __restrict__ colW /*-x*/, const uint32_t* s_data, int column,
                            int wbx, int wbz, uint32_t* recs, uint8_t* mats)
{
    uint32_t nq = 0;
    const int x = column & 15, z = column >> 4;
    uint8_t prev = 0;                                            // block below the current 16-voxel word (unused at y = 0)
    for (int w = 0; w < 24; ++w) {
        const uint4 me4 = ((const uint4*)col)[w];
        // cheap exit: a word of AIR emits nothing
        if ((me4.x | me4.y | me4.z | me4.w) == 0u) { prev = 0; continue; }
        uint4 n4[4];
        const uint8_t* nbp[4] = {colN, colE, colS, colW};
#pragma unroll
        for (int k = 0; k < 4; ++k) n4[k] = nbp[k] ? ((const uint4*)nbp[k])[w] : make_uint4(0, 0, 0, 0);
        const uint8_t next = (w < 23) ? col[16 * (w + 1)] : (uint8_t)0;
        for (int i = 0; i < 16; ++i) {
            const int b = byte_at(me4, i);
            const int y = 16 * w + i;
            if (b == MMB_AIR) continue;
            const uint32_t bd = s_data[b];
            const int trans = MESH_TRANS(bd);
            if (trans == T_XSHAPED) {
                if (EMIT) {
                    recs[nq] = quad_record(y, 6, column, bd & 15, (bd >> 4) & 15, 0, -1);
                    recs[nq + 1] = quad_record(y, 7, column, bd & 15, (bd >> 4) & 15, 0, -1);
                    mats[nq] = mats[nq + 1] = (uint8_t)mesh_material(b);
                }
                nq += 2;
                continue;
            }
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                bool show = true;
                const int ny = y + kMeshDir[d][1];
                if (ny >= 0 && ny < 384) {
                    int nb;
                    if (d < 4) {
                        if (!nbp[d]) continue;                      // neighbouring chunk absent: the face is skipped (chunk.cu:1906-1909)
                        nb = byte_at(n4[d], i);
                    } else if (d == 4) nb = (i < 15) ? byte_at(me4, i + 1) : next;
                    else nb = (i > 0) ? byte_at(me4, i - 1) : prev;
                    const int nt = MESH_TRANS(s_data[nb]);
                    show = (trans == T_TRANSPARENT) ? (nb == MMB_AIR || nt == T_SEMI) : (nt != T_OPAQUE);
                }
                if (!show) continue;
                if (EMIT) {
                    const int which = d == 4 ? 1 : (d == 5 ? 2 : 0);      // 0 side, 1 top, 2 bottom
                    const int su = (bd >> (8 * which)) & 15, sv = (bd >> (8 * which + 4)) & 15;
                    const bool rot = (bd >> (24 + which)) & 1, flip = (bd >> (27 + which)) & 1;
                    int uvStart = 0, uvFlip = -1;
                    if (rot || flip) {
                        MinStd rng = rng4(wbx + x, y, wbz + z, d);
                        if (rot) uvStart = (int)((rng.u01() * (4.f - 0.f)) + 0.f);      // uniform_real_distribution<float>(0, 4)
                        if (flip) uvFlip = (int)((rng.u01() * (4.f - 0.f)) + 0.f);
                    }
                    recs[nq] = quad_record(y, d, column, su, sv, uvStart, uvFlip);
                    mats[nq] = (uint8_t)mesh_material(b);
                }
                nq += 1;
            }
        }
        prev = (uint8_t)(me4.w >> 24);
    }
    return nq;
}

// neighbour columns of column (x, z) of chunk `c`: inside the chunk, or the facing border column of the neighbouring chunk
MM_DEV void mesh_neighbours(const uint8_t* __restrict__ blocks, const int32_t* __restrict__ neighborIdx, int o, int c, int x, int z,
                            const uint8_t*& colN, const uint8_t*& colE, const uint8_t*& colS, const uint8_t*& colW)
{
    const uint8_t* base = blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * c;
    auto chunk_of = [&](int k) -> const uint8_t* {
        const int n = neighborIdx ? neighborIdx[4 * o + k] : -1;
        return n < 0 ? nullptr : blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * n;
    };
    const uint8_t* cN = z < 15 ? base : chunk_of(0);
    const uint8_t* cE = x < 15 ? base : chunk_of(1);
    const uint8_t* cS = z > 0 ? base : chunk_of(2);
    const uint8_t* cW = x > 0 ? base : chunk_of(3);
    colN = cN ? cN + 384 * (x + 16 * ((z + 1) & 15)) : nullptr;
    colE = cE ? cE + 384 * (((x + 1) & 15) + 16 * z) : nullptr;
    colS = cS ? cS + 384 * (x + 16 * ((z + 15) & 15)) : nullptr;
    colW = cW ? cW + 384 * (((x + 15) & 15) + 16 * z) : nullptr;
}

__global__ void __launch_bounds__(256)
k_mesh_count(const uint8_t* __restrict__ blocks, const int32_t* __restrict__ chunkIdx, const int32_t* __restrict__ neighborIdx,
             uint32_t* __restrict__ columnVerts, uint32_t* __restrict__ chunkVerts)
{
    __shared__ uint32_t s_data[MMB_NUM_BLOCKS];
    __shared__ uint32_t s_total;
    const int o = blockIdx.x, t = threadIdx.x;
    const int c = chunkIdx ? chunkIdx[o] : o;           // o = position in the work list (outputs), c = chunk in the block array
    if (t < MMB_NUM_BLOCKS) s_data[t] = kBlockData[t];
    if (t == 0) s_total = 0;
    __syncthreads();
    const int x = t & 15, z = t >> 4;
    const uint8_t *colN, *colE, *colS, *colW;
    mesh_neighbours(blocks, neighborIdx, o, c, x, z, colN, colE, colS, colW);
    const uint32_t n = 4u * mesh_column<false>(blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * c + 384 * t, colN, colE, colS, colW, s_data, t, 0, 0, nullptr, nullptr);
    columnVerts[256 * o + t] = n;
    atomicAdd(&s_total, n);
    __syncthreads();
    if (t == 0) chunkVerts[o] = s_total;
}

// Vertex j (0-3) of quad record `r` (material m) as 10 dwords: pos xyz, nor xyz, uv, material lo / hi.  Table lookups go to the LDS
// copies s_dv (directionVertPositions) and s_dir (dirVecs): a dynamically indexed constant array would be a global load per lane.
struct VertexWords { uint4 a, b; uint2 c; };
MM_DEV VertexWords vertex_words(uint32_t r, int m, int j, const float2* s_jitter, const int* s_dv /*[24][3]*/, const int* s_dir /*[6][3]*/)
{
    const int y = r & 511, kind = (r >> 9) & 7, column = (r >> 12) & 255;
    const int x = column & 15, z = column >> 4;
    // uv = (tile + corner) / 16, corner rotated by uvStart and flipped (chunk.cu:1975-1990)
    const int corner = (((r >> 28) & 3) + j) & 3;
    int ou = (corner == 1 || corner == 2) ? 1 : 0, ov = corner >> 1;         // uvOffsets: (0,0) (1,0) (1,1) (0,1)
    if ((r >> 30) & 1) ou = 1 - ou;
    if ((r >> 31) & 1) ov = 1 - ov;
    const float u = (float)((int)((r >> 20) & 15) + ou) * 0.0625f, v = (float)((int)((r >> 24) & 15) + ov) * 0.0625f;
    float px, py, pz, nx, ny, nz;
    if (kind < 6) {
        const int* dv = s_dv + 3 * (4 * kind + j);
        px = (float)(x + dv[0]); py = (float)(y + dv[1]); pz = (float)(z + dv[2]);
        nx = (float)s_dir[3 * kind]; ny = (float)s_dir[3 * kind + 1]; nz = (float)s_dir[3 * kind + 2];
    } else {
        // X-shaped plant (chunk.cu:1753-1766, 1835-1871): two crossed quads around a jittered cell centre
        const float kXOff = 0x1.6a09e6p-2f;                          // 0.5f * sin(radians(45)), correctly rounded
        const float kInvSqrt2 = 1.f / __builtin_sqrtf(2.f);          // glm::normalize(vec3(1, 0, +-1)) = v * inversesqrt(dot(v, v))
        const int i = 4 * (kind - 6) + j;                            // vertex 0-7 of the plant
        const float2 jit = s_jitter[column];
        const bool negx = (i == 1 || i == 2 || i == 4 || i == 7), negz = (i == 1 || i == 2 || i == 5 || i == 6);
        px = (((float)x + 0.5f) + jit.x) + (negx ? -kXOff : kXOff);
        py = (float)y + ((j >= 2) ? 1.f : 0.f);
        pz = (((float)z + 0.5f) + jit.y) + (negz ? -kXOff : kXOff);
        nx = kInvSqrt2 * 1.f; ny = kInvSqrt2 * 0.f; nz = (i < 4) ? kInvSqrt2 * -1.f : kInvSqrt2 * 1.f;
    }
    VertexWords w;
    w.a = make_uint4(__float_as_uint(px), __float_as_uint(py), __float_as_uint(pz), __float_as_uint(nx));
    w.b = make_uint4(__float_as_uint(ny), __float_as_uint(nz), __float_as_uint(u), __float_as_uint(v));
    w.c = make_uint2((uint32_t)m, 0u);
    return w;
}

#define MESH_CAP 4096      // quads staged per batch (a column has at most 384 * 6)
__global__ void __launch_bounds__(256)
k_mesh_fill(const uint8_t* __restrict__ blocks, const int32_t* __restrict__ chunkIdx, const int32_t* __restrict__ neighborIdx,
            const int2* __restrict__ chunkWorldBlockPos, const uint32_t* __restrict__ columnVerts, const uint64_t* __restrict__ vertOffset, mmgen_vertex* __restrict__ verts,
            uint32_t* __restrict__ idx)
{
    __shared__ uint32_t s_data[MMB_NUM_BLOCKS];
    __shared__ uint32_t s_scan[256];                     // inclusive scan of the columns' quad counts
    __shared__ uint32_t s_rec[MESH_CAP];
    __shared__ uint8_t s_mat[MESH_CAP];
    __shared__ float2 s_jitter[256];
    __shared__ int s_dv[72], s_dir[18];
    __shared__ int s_end;
    const int o = blockIdx.x, t = threadIdx.x;
    const int c = chunkIdx ? chunkIdx[o] : o;
    if (t < MMB_NUM_BLOCKS) s_data[t] = kBlockData[t];
    if (t < 72) s_dv[t] = kMeshDirVert[t / 3][t % 3];
    if (t < 18) s_dir[t] = kMeshDir[t / 3][t % 3];
    const uint32_t mine = columnVerts[256 * o + t] / 4u;
    s_scan[t] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {            // Hillis-Steele
        const uint32_t v = t >= off ? s_scan[t - off] : 0u;
        __syncthreads();
        s_scan[t] += v;
        __syncthreads();
    }
    const uint32_t total = s_scan[255];
    if (total == 0) return;
    const int2 wb = chunkWorldBlockPos[o];
    {   // X-shaped jitter of this column (rand2From2 of the world block xz, chunk.cu:1838-1841); cheap enough to do for every column
        const f2 r = rand2from2((float)(wb.x + (t & 15)), (float)(wb.y + (t >> 4)));
        s_jitter[t] = make_float2(0.4f * (r.x - 0.5f), 0.4f * (r.y - 0.5f));
    }
    const uint64_t vbase = vertOffset[o];
    uint32_t* vout = (uint32_t*)(verts + vbase);
    uint32_t* iout = idx + (vbase / 4) * 6;
    const uint8_t* col = blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * c + 384 * t;
    const uint8_t *colN, *colE, *colS, *colW;
    mesh_neighbours(blocks, neighborIdx, o, c, t & 15, t >> 4, colN, colE, colS, colW);

    // batches of whole columns whose quads fit the LDS stage
    int start = 0;
    uint32_t qbase = 0;                                   // quads before column `start`
    while (start < 256) {
        if (t == 0) s_end = 256;
        __syncthreads();
        if (t >= start && s_scan[t] - qbase > MESH_CAP) atomicMin(&s_end, t);
        __syncthreads();
        const int end = s_end;
        if (t >= start && t < end && mine)
            mesh_column<true>(col, colN, colE, colS, colW, s_data, t, wb.x, wb.y, s_rec + (s_scan[t] - mine - qbase), s_mat + (s_scan[t] - mine - qbase));
        __syncthreads();
        const uint32_t nq = s_scan[end - 1] - qbase;
        // vertex stream: consecutive lanes write consecutive 40-byte vertices (16 + 16 + 8 bytes): a wave covers 2 560 contiguous bytes
        for (uint32_t k = t; k < nq * 4u; k += 256u) {
            const uint32_t q = k >> 2;
            const VertexWords w = vertex_words(s_rec[q], s_mat[q], (int)(k & 3u), s_jitter, s_dv, s_dir);
            uint32_t* o = vout + ((size_t)qbase * 4u + k) * 10u;
#ifndef MESH_NO_VSTORE
            *(uint4*)o = w.a; *(uint4*)(o + 4) = w.b; *(uint2*)(o + 8) = w.c;
#else
            if (w.a.x == 0x12345678u) *(uint4*)o = w.a;
#endif
        }
        // index stream: 6 per quad, chunk-local vertex numbers
        for (uint32_t k = t; k < nq * 6u; k += 256u) {
            const uint32_t q = k / 6u, r = k - 6u * q;
            const uint32_t pat = (r == 0u || r == 3u) ? 0u : (r == 1u ? 1u : (r == 5u ? 3u : 2u));      // 0 1 2 0 2 3
            iout[(size_t)qbase * 6u + k] = 4u * (qbase + q) + pat;
        }
        __syncthreads();
        qbase += nq;
        start = end;
    }
}

}  // namespace mm

extern "C" {

int mmgen_mesh_count(const uint8_t* d_blocks, const int32_t* d_chunk_idx, const int32_t* d_neighbor_idx, int n, uint32_t* d_column_verts,
                     uint32_t* d_chunk_verts, void* stream)
{
    if (n < 0 || (n > 0 && (!d_blocks || !d_column_verts || !d_chunk_verts))) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    hipLaunchKernelGGL(mm::k_mesh_count, dim3(n), dim3(256), 0, (hipStream_t)stream, d_blocks, d_chunk_idx, d_neighbor_idx, d_column_verts, d_chunk_verts);
    return (int)hipGetLastError();
}

int mmgen_mesh_fill(const uint8_t* d_blocks, const int32_t* d_chunk_idx, const int32_t* d_neighbor_idx, const int32_t* d_chunk_world_block_pos, int n,
                    const uint32_t* d_column_verts, const uint64_t* d_vert_offset, mmgen_vertex* d_verts, uint32_t* d_idx, void* stream)
{
    if (n < 0 || (n > 0 && (!d_blocks || !d_chunk_world_block_pos || !d_column_verts || !d_vert_offset || !d_verts || !d_idx)))
        return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    hipLaunchKernelGGL(mm::k_mesh_fill, dim3(n), dim3(256), 0, (hipStream_t)stream, d_blocks, d_chunk_idx, d_neighbor_idx, (const int2*)d_chunk_world_block_pos,
                       d_column_verts, d_vert_offset, d_verts, d_idx);
    return (int)hipGetLastError();
}

}  // extern "C"
