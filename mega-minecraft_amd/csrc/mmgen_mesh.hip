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
#include "mmgen_prof.h"

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

// ---------------------------------------------------------------------------------------------------------
// Face visibility as bit arithmetic.  Every block belongs to one of five classes (AIR, OPAQUE, SEMI_TRANSPARENT, TRANSPARENT,
// X_SHAPED) and the display rule of chunk.cu:1912-1928 only looks at classes:
//     opaque / semi-transparent block : face shows  <=>  neighbour is not OPAQUE
//     transparent block               : face shows  <=>  neighbour is AIR or SEMI_TRANSPARENT
// So each column is classified ONCE into three bit planes of its class code per 16-voxel word (LDS, 3 x 16 bits per word), the
// facing border column of a present neighbour chunk likewise, and the six face masks of a word are a handful of AND / OR / shifts
// on 16-bit masks - instead of 98 304 x 6 table lookups and branches per chunk.  Only voxels that really emit a quad are visited.
// ---------------------------------------------------------------------------------------------------------
enum { C_AIR = 0, C_OPAQUE = 1, C_SEMI = 2, C_TRANS = 3, C_XSHAPED = 4 };
#define MESH_COLS 320                                     // 256 own columns + 4 x 16 facing border columns of the neighbour chunks
struct Planes { uint16_t p0[MESH_COLS][24], p1[MESH_COLS][24], p2[MESH_COLS][24]; };      // 46 080 bytes

MM_DEV void classify_word(const uint4& v, const uint8_t* s_cls, Planes& P, int slot, int w)
{
    // terrain is long vertical runs: most words hold one block id 16 times - one lookup instead of 16
    if (v.x == v.y && v.x == v.z && v.x == v.w && v.x == ((v.x << 8) | (v.x >> 24))) {
        const uint32_t cls = s_cls[v.x & 255u];
        P.p0[slot][w] = (cls & 1u) ? 0xffffu : 0u; P.p1[slot][w] = (cls & 2u) ? 0xffffu : 0u; P.p2[slot][w] = (cls & 4u) ? 0xffffu : 0u;
        return;
    }
    const uint32_t q[4] = {v.x, v.y, v.z, v.w};
    uint32_t a = 0, b = 0, c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t cls = s_cls[(q[k] >> (8 * j)) & 255u];
            const int i = 4 * k + j;
            a |= (cls & 1u) << i; b |= ((cls >> 1) & 1u) << i; c |= (cls >> 2) << i;
        }
    P.p0[slot][w] = (uint16_t)a; P.p1[slot][w] = (uint16_t)b; P.p2[slot][w] = (uint16_t)c;
}

MM_DEV void classify_column(const uint8_t* __restrict__ col, const uint8_t* s_cls, Planes& P, int slot)
{
    // 8 independent 16-byte loads in flight per lane (a load-use-load chain would cost 24 memory latencies per column)
    for (int w0 = 0; w0 < 24; w0 += 8) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ((const uint4*)col)[w0 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) classify_word(v[k], s_cls, P, slot, w0 + k);
    }
}

struct WordMasks { uint32_t opq, semi, air, trn, xsh; };
MM_DEV WordMasks word_masks(const Planes& P, int slot, int w)
{
    const uint32_t a = P.p0[slot][w], b = P.p1[slot][w], c = P.p2[slot][w];
    WordMasks m;
    m.opq = a & ~b & ~c; m.semi = ~a & b & ~c & 0xffffu; m.air = ~(a | b | c) & 0xffffu; m.trn = a & b; m.xsh = c;
    return m;
}

// the six face masks (dirVecs order: +z, +x, -z, -x, +y, -y) and the X mask of word w of column t
struct WordFaces { uint32_t f[6], xsh; };
MM_DEV WordFaces word_faces(const Planes& P, int t, int w, const int nbSlot[4], const bool nbPresent[4])
{
    const WordMasks me = word_masks(P, t, w);
    const uint32_t solidish = me.opq | me.semi;
    WordFaces r;
    r.xsh = me.xsh;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (!nbPresent[d]) { r.f[d] = 0; continue; }      // neighbouring chunk absent: the face is skipped (chunk.cu:1906-1909)
        const WordMasks nb = word_masks(P, nbSlot[d], w);
        r.f[d] = (solidish & ~nb.opq) | (me.trn & (nb.air | nb.semi));
    }
    // vertical neighbours: the column itself shifted by one voxel; beyond the world (y = -1, 384) the face always shows, which is
    // what a virtual AIR neighbour gives under both rules
    uint32_t upOpq = me.opq >> 1, upAir = me.air >> 1, upSemi = me.semi >> 1;
    if (w < 23) { const WordMasks n = word_masks(P, t, w + 1); upOpq |= (n.opq & 1u) << 15; upAir |= (n.air & 1u) << 15; upSemi |= (n.semi & 1u) << 15; }
    else upAir |= 1u << 15;
    uint32_t dnOpq = (me.opq << 1) & 0xffffu, dnAir = (me.air << 1) & 0xffffu, dnSemi = (me.semi << 1) & 0xffffu;
    if (w > 0) { const WordMasks n = word_masks(P, t, w - 1); dnOpq |= n.opq >> 15; dnAir |= n.air >> 15; dnSemi |= n.semi >> 15; }
    else dnAir |= 1u;
    r.f[4] = (solidish & ~upOpq) | (me.trn & (upAir | upSemi));
    r.f[5] = (solidish & ~dnOpq) | (me.trn & (dnAir | dnSemi));
    return r;
}

// workgroup-wide: block class table, own columns, facing border columns of the present neighbour chunks.  Ends with a barrier.
MM_DEV void mesh_classify(const uint8_t* __restrict__ blocks, const int32_t* __restrict__ neighborIdx, int o, int c, uint8_t* s_cls, Planes& P,
                          int nbSlot[4], bool nbPresent[4])
{
    const int t = threadIdx.x, x = t & 15, z = t >> 4;
    if (t < MMB_NUM_BLOCKS) {
        const int tr = MESH_TRANS(kBlockData[t]);
        s_cls[t] = (uint8_t)(t == MMB_AIR ? C_AIR : (tr == T_OPAQUE ? C_OPAQUE : (tr == T_SEMI ? C_SEMI : (tr == T_TRANSPARENT ? C_TRANS : C_XSHAPED))));
    }
    if (t >= MMB_NUM_BLOCKS) s_cls[t] = C_OPAQUE;         // ids beyond the enum never occur; BlockData{} default is OPAQUE (block.hpp)
    __syncthreads();
    const uint8_t* base = blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * c;
    classify_column(base + 384 * t, s_cls, P, t);
    // neighbour of (x, z) in direction d: inside the chunk, or border slot 256 + 16 d + (position along the edge)
    const int inside[4] = {z < 15, x < 15, z > 0, x > 0};
    const int insideSlot[4] = {t + 16, t + 1, t - 16, t - 1};
    const int along[4] = {x, z, x, z};
    const int facing[4] = {x + 16 * 0, 0 + 16 * z, x + 16 * 15, 15 + 16 * z};      // the neighbour chunk's column that touches this one
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (inside[d]) { nbSlot[d] = insideSlot[d]; nbPresent[d] = true; continue; }
        const int n = neighborIdx ? neighborIdx[4 * o + d] : -1;
        nbSlot[d] = 256 + 16 * d + along[d];
        nbPresent[d] = n >= 0;
        if (n >= 0) classify_column(blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * n + 384 * facing[d], s_cls, P, nbSlot[d]);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256)
k_mesh_count(const uint8_t* __restrict__ blocks, const int32_t* __restrict__ chunkIdx, const int32_t* __restrict__ neighborIdx,
             uint32_t* __restrict__ columnVerts, uint32_t* __restrict__ chunkVerts)
{
    __shared__ Planes P;
    __shared__ uint8_t s_cls[256];
    __shared__ uint32_t s_total;
    const int o = blockIdx.x, t = threadIdx.x;
    const int c = chunkIdx ? chunkIdx[o] : o;           // o = position in the work list (outputs), c = chunk in the block array
    if (t == 0) s_total = 0;
    int nbSlot[4]; bool nbPresent[4];
    mesh_classify(blocks, neighborIdx, o, c, s_cls, P, nbSlot, nbPresent);
    uint32_t quads = 0;
    for (int w = 0; w < 24; ++w) {
        if ((P.p0[t][w] | P.p1[t][w] | P.p2[t][w]) == 0) continue;      // a word of AIR emits nothing
        const WordFaces f = word_faces(P, t, w, nbSlot, nbPresent);
        quads += __popc(f.f[0]) + __popc(f.f[1]) + __popc(f.f[2]) + __popc(f.f[3]) + __popc(f.f[4]) + __popc(f.f[5]) + 2 * __popc(f.xsh);
    }
    columnVerts[256 * o + t] = 4u * quads;
    atomicAdd(&s_total, 4u * quads);
    __syncthreads();
    if (t == 0) chunkVerts[o] = s_total;
}

// appends column t's quad records to recs / mats in the reference's order (y ascending, faces in dirVecs order)
MM_DEV void emit_column(const uint8_t* __restrict__ col, const Planes& P, const uint32_t* s_data, int t, const int nbSlot[4], const bool nbPresent[4],
                        int wbx, int wbz, uint32_t* recs, uint8_t* mats)
{
    const int x = t & 15, z = t >> 4;
    uint32_t nq = 0;
    for (int w = 0; w < 24; ++w) {
        if ((P.p0[t][w] | P.p1[t][w] | P.p2[t][w]) == 0) continue;      // a word of AIR emits nothing
        const WordFaces f = word_faces(P, t, w, nbSlot, nbPresent);
        uint32_t any = f.f[0] | f.f[1] | f.f[2] | f.f[3] | f.f[4] | f.f[5] | f.xsh;
        if (!any) continue;
        const uint4 me4 = ((const uint4*)col)[w];
        while (any) {
            const int i = __builtin_ctz(any);
            any &= any - 1;
            const int y = 16 * w + i;
            const int b = byte_at(me4, i);
            const uint32_t bd = s_data[b];
            const uint8_t mat = (uint8_t)mesh_material(b);
            if ((f.xsh >> i) & 1u) {
                recs[nq] = quad_record(y, 6, t, bd & 15, (bd >> 4) & 15, 0, -1);
                recs[nq + 1] = quad_record(y, 7, t, bd & 15, (bd >> 4) & 15, 0, -1);
                mats[nq] = mats[nq + 1] = mat;
                nq += 2;
                continue;
            }
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                if (!((f.f[d] >> i) & 1u)) continue;
                const int which = d == 4 ? 1 : (d == 5 ? 2 : 0);      // 0 side, 1 top, 2 bottom
                const int su = (bd >> (8 * which)) & 15, sv = (bd >> (8 * which + 4)) & 15;
                const bool rot = (bd >> (24 + which)) & 1, flip = (bd >> (27 + which)) & 1;
                int uvStart = 0, uvFlip = -1;
                if (rot || flip) {
                    MinStd rng = rng4(wbx + x, y, wbz + z, d);
                    if (rot) uvStart = (int)((rng.u01() * (4.f - 0.f)) + 0.f);      // uniform_real_distribution<float>(0, 4)
                    if (flip) uvFlip = (int)((rng.u01() * (4.f - 0.f)) + 0.f);
                }
                recs[nq] = quad_record(y, d, t, su, sv, uvStart, uvFlip);
                mats[nq] = mat;
                nq += 1;
            }
        }
    }
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
            uint32_t* __restrict__ idx, int parts /*workgroups per chunk (gridDim.y): part p emits columns [256 p / parts, 256 (p + 1) / parts)*/,
            unsigned long long capacityVerts /*0 = the caller sized the buffers from the counts; else a chunk that would end beyond it writes nothing*/,
            const uint32_t* __restrict__ chunkVerts /*non-null (a strip: gridDim.x <= 256): the chunk's offset is summed here from the chunks' counts;
                                                       vertOffsetOut[chunk] and totalOut receive what mmgen_mesh_offsets would have written*/,
            uint64_t* __restrict__ vertOffsetOut, uint64_t* __restrict__ totalOut)
{
    __shared__ Planes P;
    __shared__ uint8_t s_cls[256];
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
    __shared__ unsigned long long s_base;
    if (chunkVerts) {
        // exclusive prefix of the chunks before this one (and, in the last chunk's first part, the total): <= 256 counts, one per lane
        const int nChunks = (int)gridDim.x;
        const unsigned long long cv = t < nChunks ? chunkVerts[t] : 0u;
        unsigned long long before = t < o ? cv : 0ull, all = cv;
        for (int d = 32; d > 0; d >>= 1) { before += __shfl_xor(before, d); all += __shfl_xor(all, d); }
        __shared__ unsigned long long s_part[2][4];
        if ((t & 63) == 0) { s_part[0][t >> 6] = before; s_part[1][t >> 6] = all; }
        __syncthreads();
        if (t == 0) {
            const unsigned long long b = s_part[0][0] + s_part[0][1] + s_part[0][2] + s_part[0][3];
            s_base = b;
            if (blockIdx.y == 0) {
                vertOffsetOut[o] = b;
                if (o == nChunks - 1) *totalOut = s_part[1][0] + s_part[1][1] + s_part[1][2] + s_part[1][3];
            }
        }
    }
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
    const uint64_t vbase = chunkVerts ? (uint64_t)s_base : vertOffset[o];
    if (capacityVerts && vbase + 4ull * total > capacityVerts) return;      // (workgroup-uniform: every part of the chunk leaves)
    uint32_t* vout = (uint32_t*)(verts + vbase);
    uint32_t* iout = idx + (vbase / 4) * 6;
    const uint8_t* col = blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * c + 384 * t;
    int nbSlot[4]; bool nbPresent[4];
    mesh_classify(blocks, neighborIdx, o, c, s_cls, P, nbSlot, nbPresent);

    // batches of whole columns whose quads fit the LDS stage.  A small launch (a streaming strip: 33 chunks) gives every chunk to several
    // workgroups, each emitting a range of the columns at the offset the scan says: one workgroup writes a chunk's ~1.3 MB of vertices and
    // indices in 0.24 ms whatever the rest of the chip does (every workgroup classifies the whole chunk: 98 KB out of L2)
    const int colEnd = 256 * ((int)blockIdx.y + 1) / parts;
    int start = 256 * (int)blockIdx.y / parts;
    uint32_t qbase = start ? s_scan[start - 1] : 0u;      // quads before column `start`
    while (start < colEnd) {
        if (t == 0) s_end = colEnd;
        __syncthreads();
        if (t >= start && s_scan[t] - qbase > MESH_CAP) atomicMin(&s_end, t);
        __syncthreads();
        const int end = s_end;
        if (t >= start && t < end && mine)
            emit_column(col, P, s_data, t, nbSlot, nbPresent, wb.x, wb.y, s_rec + (s_scan[t] - mine - qbase), s_mat + (s_scan[t] - mine - qbase));
        __syncthreads();
        const uint32_t nq = s_scan[end - 1] - qbase;
        // vertex stream: consecutive lanes write consecutive 40-byte vertices (16 + 16 + 8 bytes): a wave covers 2 560 contiguous bytes
        for (uint32_t k = t; k < nq * 4u; k += 256u) {
            const uint32_t q = k >> 2;
            const VertexWords w = vertex_words(s_rec[q], s_mat[q], (int)(k & 3u), s_jitter, s_dv, s_dir);
            uint32_t* o = vout + ((size_t)qbase * 4u + k) * 10u;
            *(uint4*)o = w.a; *(uint4*)(o + 4) = w.b; *(uint2*)(o + 8) = w.c;
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

// exclusive prefix of the chunks' vertex counts (one workgroup: a streaming strip has 33 chunks, a pool a few thousand)
__global__ void __launch_bounds__(256)
k_mesh_offsets(const uint32_t* __restrict__ chunkVerts, int n, uint64_t* __restrict__ vertOffset, uint64_t* __restrict__ total)
{
    __shared__ unsigned long long s_scan[256];
    __shared__ unsigned long long s_carry;
    const int t = threadIdx.x;
    if (t == 0) s_carry = 0ull;
    __syncthreads();
    for (int base = 0; base < n; base += 256) {
        const unsigned long long mine = base + t < n ? chunkVerts[base + t] : 0u;
        s_scan[t] = mine;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const unsigned long long v = t >= off ? s_scan[t - off] : 0ull;
            __syncthreads();
            s_scan[t] += v;
            __syncthreads();
        }
        const unsigned long long carry = s_carry;
        if (base + t < n) vertOffset[base + t] = carry + s_scan[t] - mine;
        __syncthreads();
        if (t == 255) s_carry = carry + s_scan[255];
        __syncthreads();
    }
    if (t == 0) *total = s_carry;
}

}  // namespace mm

extern "C" {

int mmgen_mesh_count(const uint8_t* d_blocks, const int32_t* d_chunk_idx, const int32_t* d_neighbor_idx, int n, uint32_t* d_column_verts,
                     uint32_t* d_chunk_verts, void* stream)
{
    if (n < 0 || (n > 0 && (!d_blocks || !d_column_verts || !d_chunk_verts))) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    MMK_LAUNCH_NORET(mmk::KID_MESH_COUNT, mm::k_mesh_count, dim3(n), dim3(256), (hipStream_t)stream, d_blocks, d_chunk_idx, d_neighbor_idx, d_column_verts, d_chunk_verts);
    return (int)hipGetLastError();
}

static int mesh_fill_launch(const uint8_t* d_blocks, const int32_t* d_chunk_idx, const int32_t* d_neighbor_idx, const int32_t* d_chunk_world_block_pos, int n,
                            const uint32_t* d_column_verts, const uint64_t* d_vert_offset, unsigned long long capacity, mmgen_vertex* d_verts, uint32_t* d_idx, void* stream,
                            const uint32_t* d_chunk_verts = nullptr, uint64_t* d_vert_offset_out = nullptr, uint64_t* d_total_out = nullptr)
{
    if (n < 0 || (n > 0 && (!d_blocks || !d_chunk_world_block_pos || !d_column_verts || !(d_vert_offset || d_chunk_verts) || !d_verts || !d_idx)))
        return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    int parts = 1;                                        // ~512 workgroups at least, at most 8 per chunk
    while (parts < 8 && n * parts * 2 <= 512) parts *= 2;
    MMK_LAUNCH_NORET(mmk::KID_MESH_FILL, mm::k_mesh_fill, dim3(n, parts), dim3(256), (hipStream_t)stream, d_blocks, d_chunk_idx, d_neighbor_idx, (const int2*)d_chunk_world_block_pos,
                       d_column_verts, d_vert_offset, d_verts, d_idx, parts, capacity, d_chunk_verts, d_vert_offset_out, d_total_out);
    return (int)hipGetLastError();
}

int mmgen_mesh_fill(const uint8_t* d_blocks, const int32_t* d_chunk_idx, const int32_t* d_neighbor_idx, const int32_t* d_chunk_world_block_pos, int n,
                    const uint32_t* d_column_verts, const uint64_t* d_vert_offset, mmgen_vertex* d_verts, uint32_t* d_idx, void* stream)
{
    return mesh_fill_launch(d_blocks, d_chunk_idx, d_neighbor_idx, d_chunk_world_block_pos, n, d_column_verts, d_vert_offset, 0ull, d_verts, d_idx, stream);
}

int mmgen_mesh_fill_capped(const uint8_t* d_blocks, const int32_t* d_chunk_idx, const int32_t* d_neighbor_idx, const int32_t* d_chunk_world_block_pos, int n,
                           const uint32_t* d_column_verts, const uint64_t* d_vert_offset, uint64_t capacity_verts, mmgen_vertex* d_verts, uint32_t* d_idx,
                           void* stream)
{
    if (capacity_verts == 0) return (int)hipErrorInvalidValue;
    return mesh_fill_launch(d_blocks, d_chunk_idx, d_neighbor_idx, d_chunk_world_block_pos, n, d_column_verts, d_vert_offset, capacity_verts, d_verts, d_idx, stream);
}

int mmgen_mesh_fill_strip(const uint8_t* d_blocks, const int32_t* d_chunk_idx, const int32_t* d_neighbor_idx, const int32_t* d_chunk_world_block_pos, int n,
                          const uint32_t* d_column_verts, const uint32_t* d_chunk_verts, uint64_t* d_vert_offset, uint64_t* d_total, uint64_t capacity_verts,
                          mmgen_vertex* d_verts, uint32_t* d_idx, void* stream)
{
    if (capacity_verts == 0 || n < 1 || n > 256 || !d_chunk_verts || !d_vert_offset || !d_total) return (int)hipErrorInvalidValue;
    return mesh_fill_launch(d_blocks, d_chunk_idx, d_neighbor_idx, d_chunk_world_block_pos, n, d_column_verts, nullptr, capacity_verts, d_verts, d_idx, stream,
                            d_chunk_verts, d_vert_offset, d_total);
}

int mmgen_mesh_offsets(const uint32_t* d_chunk_verts, int n, uint64_t* d_vert_offset, uint64_t* d_total, void* stream)
{
    if (n < 0 || !d_total || (n > 0 && (!d_chunk_verts || !d_vert_offset))) return (int)hipErrorInvalidValue;
    MMK_LAUNCH_NORET(mmk::KID_MESH_COUNT, mm::k_mesh_offsets, dim3(1), dim3(256), (hipStream_t)stream, d_chunk_verts, n, d_vert_offset, d_total);
    return (int)hipGetLastError();
}

}  // extern "C"
