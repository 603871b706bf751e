// mmgen mesher for gfx950: the mesh build that follows the generation path.
// Behavioural spec: Chunk::createVBOs (chunk.cu:1751-2003): per voxel in z, x, y order an X-shaped block emits 8 vertices +
// 12 indices, a cube emits 4 vertices + 6 indices per face whose neighbour lets it show, in DirectionEnums::dirVecs order;
// Vertex = {vec3 pos, vec3 nor, vec2 uv, size_t material} (rendering/structs.hpp:25-31), indices local to the chunk.
//
// The reference walks 98 304 voxels x 6 neighbours per chunk on the host with std::vector pushes (the most expensive action of
// its scheduler).  Here: one workgroup per chunk, one lane per column (the reference's z, x order IS the column index), two
// kernels sharing one traversal template:
//   k_mesh_count  vertices per column -> columnVerts[n][256], per chunk -> chunkVerts[n]   (indices = 3/2 vertices, always)
//   k_mesh_fill   exclusive scan of the 256 column counts in LDS, then every lane writes its column's vertices and indices at
//                 its own offset: output order is the reference's by construction, no atomics, no sorting.
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
__device__ constexpr int kMeshUvOff[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};

// byte i of a 16-byte word without indexing registers dynamically
MM_DEV int byte_at(const uint4& v, int i)
{
    const uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    return (int)(((i < 8 ? lo : hi) >> (8 * (i & 7))) & 255u);
}

MM_DEV void put_vertex(mmgen_vertex* v, float px, float py, float pz, float nx, float ny, float nz, float u, float w, int mat)
{
    v->pos[0] = px; v->pos[1] = py; v->pos[2] = pz;
    v->nor[0] = nx; v->nor[1] = ny; v->nor[2] = nz;
    v->uv[0] = u; v->uv[1] = w;
    v->material = (uint64_t)mat;
}

// One lane = one column.  FILL = false: returns the column's vertex count.  FILL = true: writes vertices / indices from
// vertex index `v0` (chunk-local) on; verts / idx point at the chunk's first vertex / index.
template <bool FILL>
MM_DEV uint32_t mesh_column(const uint8_t* __restrict__ col, const uint8_t* __restrict__ colN /*+z*/, const uint8_t* __restrict__ colE /*+x*/,
                            const uint8_t* __restrict__ colS /*-z*/, const uint8_t* __restrict__ colW /*-x*/, const uint32_t* s_data, int x, int z,
                            int wbx, int wbz, uint32_t v0, mmgen_vertex* __restrict__ verts, uint32_t* __restrict__ idx)
{
    uint32_t nv = v0;
    const float kXOff = 0x1.6a09e6p-2f;                          // 0.5f * sin(radians(45)), correctly rounded (chunk.cu:1753)
    const float kInvSqrt2 = 1.f / __builtin_sqrtf(2.f);          // glm::normalize(vec3(1, 0, +-1)) = v * inversesqrt(dot(v, v))
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
            if (b != MMB_AIR) {
                const uint32_t bd = s_data[b];
                const int trans = MESH_TRANS(bd);
                const int mat = mesh_material(b);
                if (trans == T_XSHAPED) {
                    if (FILL) {
                        const f2 r = rand2from2((float)(wbx + x), (float)(wbz + z));
                        const float bx = ((float)x + 0.5f) + 0.4f * (r.x - 0.5f), by = (float)y, bz = ((float)z + 0.5f) + 0.4f * (r.y - 0.5f);
                        const float su = (float)(bd & 15), sv = (float)((bd >> 4) & 15);
                        const float px[8] = {kXOff, -kXOff, -kXOff, kXOff, -kXOff, kXOff, kXOff, -kXOff};
                        const float pz[8] = {kXOff, -kXOff, -kXOff, kXOff, kXOff, -kXOff, -kXOff, kXOff};
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            put_vertex(verts + nv + j, bx + px[j], by + ((j & 3) >= 2 ? 1.f : 0.f), bz + pz[j], kInvSqrt2 * 1.f, kInvSqrt2 * 0.f,
                                       j < 4 ? kInvSqrt2 * -1.f : kInvSqrt2 * 1.f, (su + (float)kMeshUvOff[j & 3][0]) * 0.0625f,
                                       (sv + (float)kMeshUvOff[j & 3][1]) * 0.0625f, mat);
                        uint32_t* ip = idx + (nv / 4) * 6;
                        const uint32_t q[12] = {0, 1, 2, 0, 2, 3, 4, 5, 6, 4, 6, 7};
#pragma unroll
                        for (int k = 0; k < 12; ++k) ip[k] = nv + q[k];
                    }
                    nv += 8;
                } else {
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
                        if (FILL) {
                            const int which = d == 4 ? 1 : (d == 5 ? 2 : 0);      // 0 side, 1 top, 2 bottom
                            const int su = (bd >> (8 * which)) & 15, sv = (bd >> (8 * which + 4)) & 15;
                            const bool rot = (bd >> (24 + which)) & 1, flip = (bd >> (27 + which)) & 1;
                            int uvStart = 0, uvFlip = -1;
                            if (rot || flip) {
                                MinStd rng = rng4(wbx + x, y, wbz + z, d);
                                if (rot) uvStart = (int)((rng.u01() * (4.f - 0.f)) + 0.f);      // uniform_real_distribution<float>(0, 4)
                                if (flip) uvFlip = (int)((rng.u01() * (4.f - 0.f)) + 0.f);
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                int ou = kMeshUvOff[(uvStart + j) & 3][0], ov = kMeshUvOff[(uvStart + j) & 3][1];
                                if (uvFlip != -1) {
                                    if (uvFlip & 1) ou = 1 - ou;
                                    if (uvFlip & 2) ov = 1 - ov;
                                }
                                put_vertex(verts + nv + j, (float)(x + kMeshDirVert[4 * d + j][0]), (float)(y + kMeshDirVert[4 * d + j][1]),
                                           (float)(z + kMeshDirVert[4 * d + j][2]), (float)kMeshDir[d][0], (float)kMeshDir[d][1], (float)kMeshDir[d][2],
                                           (float)(su + ou) * 0.0625f, (float)(sv + ov) * 0.0625f, mat);
                            }
                            uint32_t* ip = idx + (nv / 4) * 6;
                            ip[0] = nv; ip[1] = nv + 1; ip[2] = nv + 2; ip[3] = nv; ip[4] = nv + 2; ip[5] = nv + 3;
                        }
                        nv += 4;
                    }
                }
            }
        }
        prev = (uint8_t)(me4.w >> 24);
    }
    return nv - v0;
}

// neighbour columns of column (x, z) of chunk `c`: inside the chunk, or the facing border column of the neighbouring chunk
MM_DEV void mesh_neighbours(const uint8_t* __restrict__ blocks, const int32_t* __restrict__ neighborIdx, int c, int x, int z, const uint8_t*& colN,
                            const uint8_t*& colE, const uint8_t*& colS, const uint8_t*& colW)
{
    const uint8_t* base = blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * c;
    auto chunk_of = [&](int k) -> const uint8_t* {
        const int n = neighborIdx ? neighborIdx[4 * c + k] : -1;
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
k_mesh_count(const uint8_t* __restrict__ blocks, const int32_t* __restrict__ neighborIdx, uint32_t* __restrict__ columnVerts,
             uint32_t* __restrict__ chunkVerts)
{
    __shared__ uint32_t s_data[MMB_NUM_BLOCKS];
    __shared__ uint32_t s_total;
    const int c = blockIdx.x, t = threadIdx.x;
    if (t < MMB_NUM_BLOCKS) s_data[t] = kBlockData[t];
    if (t == 0) s_total = 0;
    __syncthreads();
    const int x = t & 15, z = t >> 4;
    const uint8_t *colN, *colE, *colS, *colW;
    mesh_neighbours(blocks, neighborIdx, c, x, z, colN, colE, colS, colW);
    const uint32_t n = mesh_column<false>(blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * c + 384 * t, colN, colE, colS, colW, s_data, x, z, 0, 0, 0, nullptr, nullptr);
    columnVerts[256 * c + t] = n;
    atomicAdd(&s_total, n);
    __syncthreads();
    if (t == 0) chunkVerts[c] = s_total;
}

__global__ void __launch_bounds__(256)
k_mesh_fill(const uint8_t* __restrict__ blocks, const int32_t* __restrict__ neighborIdx, const int2* __restrict__ chunkWorldBlockPos,
            const uint32_t* __restrict__ columnVerts, const uint64_t* __restrict__ vertOffset, mmgen_vertex* __restrict__ verts,
            uint32_t* __restrict__ idx)
{
    __shared__ uint32_t s_data[MMB_NUM_BLOCKS];
    __shared__ uint32_t s_scan[256];
    const int c = blockIdx.x, t = threadIdx.x;
    if (t < MMB_NUM_BLOCKS) s_data[t] = kBlockData[t];
    const uint32_t mine = columnVerts[256 * c + t];
    s_scan[t] = mine;
    __syncthreads();
    // inclusive Hillis-Steele scan over the 256 columns
    for (int off = 1; off < 256; off <<= 1) {
        const uint32_t v = t >= off ? s_scan[t - off] : 0u;
        __syncthreads();
        s_scan[t] += v;
        __syncthreads();
    }
    if (mine == 0) return;
    const uint32_t v0 = s_scan[t] - mine;
    const int x = t & 15, z = t >> 4;
    const uint8_t *colN, *colE, *colS, *colW;
    mesh_neighbours(blocks, neighborIdx, c, x, z, colN, colE, colS, colW);
    const uint64_t base = vertOffset[c];
    const int2 wb = chunkWorldBlockPos[c];
    mesh_column<true>(blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * c + 384 * t, colN, colE, colS, colW, s_data, x, z, wb.x, wb.y, v0, verts + base,
                      idx + (base / 4) * 6);
}

}  // namespace mm

extern "C" {

int mmgen_mesh_count(const uint8_t* d_blocks, const int32_t* d_neighbor_idx, int n, uint32_t* d_column_verts, uint32_t* d_chunk_verts, void* stream)
{
    if (n < 0 || (n > 0 && (!d_blocks || !d_column_verts || !d_chunk_verts))) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    hipLaunchKernelGGL(mm::k_mesh_count, dim3(n), dim3(256), 0, (hipStream_t)stream, d_blocks, d_neighbor_idx, d_column_verts, d_chunk_verts);
    return (int)hipGetLastError();
}

int mmgen_mesh_fill(const uint8_t* d_blocks, const int32_t* d_neighbor_idx, const int32_t* d_chunk_world_block_pos, int n,
                    const uint32_t* d_column_verts, const uint64_t* d_vert_offset, mmgen_vertex* d_verts, uint32_t* d_idx, void* stream)
{
    if (n < 0 || (n > 0 && (!d_blocks || !d_chunk_world_block_pos || !d_column_verts || !d_vert_offset || !d_verts || !d_idx)))
        return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    hipLaunchKernelGGL(mm::k_mesh_fill, dim3(n), dim3(256), 0, (hipStream_t)stream, d_blocks, d_neighbor_idx, (const int2*)d_chunk_world_block_pos,
                       d_column_verts, d_vert_offset, d_verts, d_idx);
    return (int)hipGetLastError();
}

}  // extern "C"
