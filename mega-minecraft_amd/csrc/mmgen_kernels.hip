// mmgen HIP kernels for gfx950 (MI355X): heightfield + biome weights, layers, cave carving, chunk fill.
// One wave = 64 lanes; all kernels are FP32-VALU bound (no dense contraction → no MFMA), so the design rules are:
// one lane per column (2D stages) or per voxel with y fastest (3D stages) for coalesced plane / byte stores, per-column
// invariants hoisted into tiny pre-pass kernels, Worley cell points staged once per workgroup in LDS, wave ballots for
// run-length extraction.  Compile with -ffp-contract=off (bit-exact contract, see mm_math.cuh).
//
// Behavioural spec (reference file:line): kernGenerateHeightfield chunk.cu:150-185, kernGenerateLayers :322-415,
// shouldGenerateCaveAtBlock :755-810, kernGenerateCaves :812-937, chunkFillPlaceBlock :1202-1380, kernFill :1382-1510.
#include "mm_biome.cuh"
#include "mmgen_kernels.h"
#include "mmgen_prof.h"

namespace mm {

// XCD-aware workgroup order for the kernels that run 64 workgroups (4-column groups) per chunk.  The dispatcher deals consecutive
// workgroup ids round-robin over the 8 XCDs, each with its own L2: neighbouring groups of a chunk read the same 128-byte lines of
// the attribute planes (16 bytes each), so with the plain order every line is fetched by up to 8 L2s.  This map gives every XCD
// whole chunks: of each run of 8 chunks (512 workgroup ids), XCD k takes the k-th.  MEASURED (profiles/README.md, r01h): k_fill
// read traffic 117 MB -> 37 MB per launch (= algorithmic), but launch time 1.01 -> 1.11 ms; these kernels are VALU-issue bound at
// < 2 % of HBM peak, so the plain order stays the default and the map is kept behind MM_XCD_SWIZZLE for HBM-bound variants.
#ifndef MM_XCD_SWIZZLE
#define MM_XCD_SWIZZLE 0
#endif
MM_DEV int xcd_block(int b, int n) { return (!MM_XCD_SWIZZLE || (n & 511)) ? b : (((b >> 9) * 8 + (b & 7)) << 6) + ((b >> 3) & 63); }


// =========================================================================================================
// K1 — heightfield + 24 biome weights.  GATHERED variant also produces the 18x18 ring the layer stage needs
// (the ring heights are a pure function of position, so no neighbour chunk is read).
// =========================================================================================================
#ifndef MM_HF_WAVES
#define MM_HF_WAVES 6           // latency bound (a lane = one column's chain of table lookups): 0.81 / 0.70 / 0.56 ms at 4 / 5 / 6 waves per SIMD
#endif
template <bool GATHERED>
__attribute__((amdgpu_waves_per_eu(MM_HF_WAVES, MM_HF_WAVES)))
__global__ void __launch_bounds__(GATHERED ? 384 : 256)
k_heightfield(const int2* __restrict__ chunkPos, float* __restrict__ hf, float* __restrict__ bw, float* __restrict__ gathered)
{
    noise_tables_init();
    const int chunk = blockIdx.x;
    const int t = threadIdx.x;
    int x, z;
    if (GATHERED) {
        if (t >= 324) return;
        x = (t % 18) - 1;
        z = (t / 18) - 1;
    } else {
        x = t & 15;
        z = t >> 4;
    }
    const int2 cp = chunkPos[chunk];
    const bool interior = (x >= 0) & (x < 16) & (z >= 0) & (z < 16);
    const int idx = x + 16 * z;

    const float wx = (float)(cp.x + x), wz = (float)(cp.y + z);
    const BiomeNoise bn = biome_noise(wx, wz);
    // The weights first, then the heights of the biomes that have one: the loop of the 24 height functions (the register peak of the
    // kernel) keeps a mask of the positive weights instead of the biome noise, and reads each weight back from where it was just
    // stored - its plane for the chunk's own columns, an LDS row for the 68 ring columns of the gathered variant.
    __shared__ float s_ringW[GATHERED ? MMGEN_NUM_BIOMES * 68 : 1];
    float* wout = bw + (size_t)MMGEN_BIOME_WEIGHTS_SIZE * chunk + idx;
    int stride = 256;
    if (GATHERED && !interior) {
        const int ord = z == -1 ? x + 1 : (z == 16 ? 19 + x : (x == -1 ? 36 + z : 52 + z));      // 18 + 18 + 16 + 16 ring columns
        wout = s_ringW + ord;
        stride = 68;
    }
    unsigned positive = 0u;
    for (int b = 0; b < MMGEN_NUM_BIOMES; ++b) {
        const float w = biome_weight(b, bn);
        if (w > 0.f) positive |= 1u << b;
        wout[stride * b] = w;
    }
    float height = 0.f;
    for (int b = 0; b < MMGEN_NUM_BIOMES; ++b) {
        if (!((positive >> b) & 1u)) continue;
        height += wout[stride * b] * biome_height(b, wx, wz);
    }
    if (interior) hf[(size_t)256 * chunk + idx] = height;
    if (GATHERED) gathered[(size_t)MMGEN_GATHERED_HEIGHTFIELD_SIZE * chunk + t] = height;
}

// =========================================================================================================
// K2 — layers.  Input: gathered 18x18 heights, biome weights.  Output: 20 layer-start planes.
// =========================================================================================================
MM_DEV float stratified_thickness(int layer, float weight, float wx, float wz)
{
    if (weight > 0.f) {
        const float s = kMaterialScaleOrMaxSlope[layer];
        const float o = (float)layer * 5283.64f;
        const float v = kMaterialThickness[layer] + kMaterialAmpOrTan[layer] * fbm2<5>(wx * s + o, wz * s + o);
        return gmax(0.f, v) * weight;
    }
    return 0.f;
}

__global__ void __launch_bounds__(256)
k_layers(const float* __restrict__ gathered, const float* __restrict__ bw, const int2* __restrict__ chunkPos, float* __restrict__ layers,
         float* __restrict__ stratifiedCopy /*nullable*/, int nCopy)
{
    noise_tables_init();
    __shared__ float s_h[MMGEN_GATHERED_HEIGHTFIELD_SIZE];
    const int chunk = blockIdx.x;
    const int t = threadIdx.x;
    const int x = t & 15, z = t >> 4;
    const float* g = gathered + (size_t)MMGEN_GATHERED_HEIGHTFIELD_SIZE * chunk;
    s_h[t] = g[t];
    if (t + 256 < MMGEN_GATHERED_HEIGHTFIELD_SIZE) s_h[t + 256] = g[t + 256];
    __syncthreads();

    const int2 cp = chunkPos[chunk];
    const float wx = (float)(cp.x + x), wz = (float)(cp.y + z);

    float tw[MMGEN_NUM_MATERIALS];
#pragma unroll
    for (int m = 0; m < MMGEN_NUM_MATERIALS; ++m) tw[m] = 0.f;
    const float* cbw = bw + (size_t)MMGEN_BIOME_WEIGHTS_SIZE * chunk + t;
#pragma unroll
    for (int b = 0; b < MMGEN_NUM_BIOMES; ++b) {
        const float w = cbw[256 * b];
#pragma unroll
        for (int m = 0; m < MMGEN_NUM_MATERIALS; ++m) tw[m] += w * kMatWeights.w[b][m];
    }

    const int c18 = (x + 1) + 18 * (z + 1);
    const float maxHeight = s_h[c18];
    float slope = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float nh = s_h[c18 + kDirX[i] + 18 * kDirZ[i]];
        slope = gmax(slope, __builtin_fabsf(nh - maxHeight) * ((i & 1) ? MM_SQRT_2 : 1.f));
    }

    float* out = layers + (size_t)MMGEN_LAYERS_SIZE * chunk + t;
    // region path with erosion: the first nCopy chunks' twelve stratified layers also go straight to the buffer the eroded planes will join
    // (what a device copy of all twenty layers did on the erosion branch's critical path)
    float* out2 = (stratifiedCopy && chunk < nCopy) ? stratifiedCopy + (size_t)MMGEN_LAYERS_SIZE * chunk + t : nullptr;

    // forward stratified layers 0..9: start = running height; stop accumulating once above the surface.  Layers after
    // the stop carry the running height (the reference leaves them unwritten; they never influence a block).
    float height = 0.f;
    bool stopped = false;
#pragma unroll
    for (int l = 0; l < MMGEN_NUM_FORWARD_MATERIALS; ++l) {
        out[256 * l] = height;
        if (out2) out2[256 * l] = height;
        if (!stopped) {
            if (height > maxHeight || l == MMGEN_NUM_FORWARD_MATERIALS - 1) stopped = true;
            else height += stratified_thickness(l, tw[l], wx, wz);
        }
    }
    // backward stratified layers 11, 10: cumulative thickness (turned into a start height by the fix-up after erosion)
    height = 0.f;
#pragma unroll
    for (int l = MMGEN_NUM_STRATIFIED_MATERIALS - 1; l >= MMGEN_NUM_FORWARD_MATERIALS; --l) {
        height += stratified_thickness(l, tw[l], wx, wz);
        out[256 * l] = height;
        if (out2) out2[256 * l] = height;
    }
    // eroded layers 19..12 from the surface down, thinned by slope
    height = maxHeight;
#pragma unroll
    for (int l = MMGEN_NUM_MATERIALS - 1; l >= MMGEN_NUM_STRATIFIED_MATERIALS; --l) {
        const float ms = kMaterialScaleOrMaxSlope[l];
        const float lh = gmax(0.f, kMaterialThickness[l] * ((ms - slope) / ms)) * tw[l];
        height -= lh;
        out[256 * l] = height;
    }
}

// E3 — fixBackwardStratifiedLayers (chunk.cu:725-749): layers[10], layers[11] = start_12 - cumulative thickness
__global__ void __launch_bounds__(256) k_fix_backward(float* __restrict__ layers)
{
    float* col = layers + (size_t)MMGEN_LAYERS_SIZE * blockIdx.x + threadIdx.x;
    const float erodedStart = col[256 * MMGEN_NUM_STRATIFIED_MATERIALS];
    col[256 * 10] = erodedStart - col[256 * 10];
    col[256 * 11] = erodedStart - col[256 * 11];
}

// =========================================================================================================
// K4 — caves.
//   k_cave_columns : per column, everything of shouldGenerateCaveAtBlock that does not depend on y
//                    (ocean+beach weight, the whole ravine branch → one y threshold)
//   k_cave_voxels  : one workgroup (4 waves) per 16-column row, four dense phases over row-long voxel lists; Worley cell points of the row's
//                    reachable 8x8x7 cell box staged in LDS; solid/air bits → LDS bit words → popcount ranks → (start,end) runs
//   k_cave_biomes  : the occupied layer slots' (bottom, top) cave-biome evaluations, streamed by persistent waves 64 at a time
// =========================================================================================================
// one wave, one lane working: returns when *counter >= target or after ~3 ms (the wait is an optimisation, never a condition)
__global__ void __launch_bounds__(64) k_wait_counter(const unsigned* counter, unsigned target)
{
    if (threadIdx.x != 0) return;
    for (int i = 0; i < 4096; ++i) {
        if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return;
        __builtin_amdgcn_s_sleep(32);
    }
}

__global__ void __launch_bounds__(256)
k_cave_columns(const float* __restrict__ bw, const int2* __restrict__ chunkPos, float2* __restrict__ colInfo, const int* __restrict__ chunkList,
               unsigned* __restrict__ clearWords /*k_cave_biomes' work counters, 16 * CB_COUNTERS words: cleared here, two launches ahead of their use*/)
{
    if (blockIdx.x == 0) clearWords[threadIdx.x] = 0u;
    noise_tables_init();
    const int chunk = chunkList ? chunkList[blockIdx.x] : blockIdx.x, t = threadIdx.x;
    const int2 cp = chunkPos[chunk];
    const int wx = cp.x + (t & 15), wz = cp.y + (t >> 4);

    float obw = 0.f;   // canonical order: ascending biome index
    const float* cbw = bw + (size_t)MMGEN_BIOME_WEIGHTS_SIZE * chunk + t;
#pragma unroll
    for (int b = 0; b < MMGEN_NUM_OCEAN_AND_BEACH_BIOMES; ++b) obw += cbw[256 * b];

    const float rx = (float)wx * 0.0015f, rz = (float)wz * 0.0015f;
    const f2 ro = fbm2from2<4>(rx * 10.f, rz * 10.f);
    const Worley2 rw = worley2(rx + 0.03f * ro.x, rz + 0.03f * ro.y);
    const float thr = 0.12f * (1.f - obw);
    float ravineY = __builtin_inff();
    if (rw.d1 < thr) {
        const f3 color = rand3from2(rw.closest.x, rw.closest.y);
        const float ravineTop = 120.f + 24.f * color.x;
        const float ratio = 1.f - (rw.d1 / thr);
        float depth = 60.f + 26.f * fbm2<4>(rx * 8.f + 8391.32f, rz * 8.f + 4821.39f);
        depth *= smoothstep(0.f, 0.3f, ratio);
        const float waveOff = 4.f * fbm2<4>(rx * 3.f + 5129.32f, rz * 3.f + 1392.49f);
        float wave = sinf_((rx + rz) * 15.f + waveOff);
        wave = smoothstep(0.4f, 0.6f, wave);
        depth *= wave;
        if (depth > 0.0001f) ravineY = ravineTop - depth;
    }
    colInfo[(size_t)256 * chunk + t] = make_float2(obw, ravineY);
}

// smoothstep(0.2, 0.4, fbm3<4>(p)) of the cave threshold (chunk.cu shouldGenerateCaveAtBlock).  Exact pruning: the smoothstep is exactly 0
// for an argument <= 0.2 and exactly 1 for one >= 0.4, the octaves still to come after octave i add at most amp_i * B3 in magnitude
// (their amplitudes sum to < amp_i; B3 = MM_SIMPLEX3_BOUND, mm_noise.cuh), so once the partial sum is that far on either
// side the remaining octaves cannot move the result.  The partial sums are the reference's own (same order of additions).  This noise
// varies over thousands of blocks: a workgroup's voxels nearly always leave the loop together.
#ifndef MM_CAVE_HUGE_PRUNE
#define MM_CAVE_HUGE_PRUNE 1
#endif
MM_DEV float cave_huge(float x, float y, float z, float b3 /* MM_SIMPLEX3_BOUND inside the pruning domain, FLT_MAX outside: no early exit */)
{
    float acc = 0.f, amp = 1.f;
    // the gradients' table-domain test once for the four octaves (mm_noise.cuh fbm3): the argument is 0.00035 x the block position
    const bool tables = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(x), __builtin_fabsf(y)), __builtin_fabsf(z)) < (float)(1 << 18);
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        amp *= 0.5f;
        acc += amp * (__builtin_expect(tables, 1) ? simplex3_inl<true>(x, y, z) : simplex3(x, y, z));
        x *= 2.f; y *= 2.f; z *= 2.f;
#if MM_CAVE_HUGE_PRUNE
        const float rest = amp * b3;
        if (acc + rest <= 0.2f - 0.001f) return 0.f;
        if (acc - rest >= 0.4f + 0.001f) return 1.f;
#endif
    }
    return smoothstep(0.2f, 0.4f, acc);
}

// The two height ratios of the cave threshold (chunk.cu shouldGenerateCaveAtBlock).  A smoothstep is exactly 1 once its argument reaches
// the upper edge (t clamps to 1, 1 * 1 * (3 - 2) = 1), which is where most voxels of a list entry's wave are (the lists are ordered by
// depth): the division is only evaluated inside the transitions.
MM_DEV float cave_top_ratio(float x /* fy + obw * 50 */) { return x <= 95.f ? 1.f : smoothstep(142.f, 95.f, x); }
MM_DEV float cave_bottom_ratio(float fy) { return fy >= 20.f ? 1.f : smoothstep(5.f, 20.f, fy); }

#define CELL_NX 8          // 4 adjacent columns share one tile: +1 cell in x over the single-column reach
#define CELL_NY 8
#define CELL_NZ 7
#define CELL_N (CELL_NX * CELL_NY * CELL_NZ)
#define CAVE_YEVAL 144     // voxels y < 144 may need the noise (threshold is 0 once y + 50*obw >= 142)
#define CAVE_ROW 16        // one workgroup = one 16-column row of a chunk
#define CAVE_VOXELS (CAVE_ROW * CAVE_YEVAL)       // 2 304 evaluated voxels per row
#ifndef CAVE_L2_CAP
#define CAVE_L2_CAP 1096                          // list 2 (typically 40 - 45 % of list 1, which has at most 2 304 entries); a full list resolves the surplus in place
#endif
#define CAVE_L3_CAP (CAVE_VOXELS / 3)             // list 3 (typically a third of list 2)
#ifndef CAVE_THREADS
#define CAVE_THREADS 256                          // 4 waves: six workgroups (26.8 KB of LDS each: 26 848 of the 26 880 B a sixth of the CU's LDS is) fill a CU's 24 wave slots
#endif

struct CellTile {
    const float* pts;     // LDS, 3 floats per cell
    int ox, oy, oz;
    static constexpr int kBoxStrideX = 3 * CELL_NY * CELL_NZ, kBoxStrideY = 3 * CELL_NZ;
    // the staged point of cell (ux - 1, uy - 1, uz - 1) when all 27 cells around (ux, uy, uz) are staged, else null
    MM_DEV const float* box27(int ux, int uy, int uz) const
    {
        const int ix = ux - 1 - ox, iy = uy - 1 - oy, iz = uz - 1 - oz;
        if ((unsigned)ix <= CELL_NX - 3 && (unsigned)iy <= CELL_NY - 3 && (unsigned)iz <= CELL_NZ - 3) return pts + 3 * ((ix * CELL_NY + iy) * CELL_NZ + iz);
        return nullptr;
    }
    MM_DEV f3 operator()(int cx, int cy, int cz) const
    {
        const int ix = cx - ox, iy = cy - oy, iz = cz - oz;
        if ((unsigned)ix < CELL_NX && (unsigned)iy < CELL_NY && (unsigned)iz < CELL_NZ) {
            const float* p = pts + 3 * ((ix * CELL_NY + iy) * CELL_NZ + iz);
            return mk3(p[0], p[1], p[2]);
        }
        return rand3from3((float)cx, (float)cy, (float)cz);   // outside the staged box: same value, computed directly
    }
    // the three z-consecutive cells (cx, cy, cz - 1 .. cz + 1): one bounds test and 9 consecutive LDS words when the row is staged
    MM_DEV void row3(int cx, int cy, int cz, f3 (&out)[3]) const
    {
        const int ix = cx - ox, iy = cy - oy, iz = cz - oz - 1;
        if ((unsigned)ix < CELL_NX && (unsigned)iy < CELL_NY && iz >= 0 && iz + 2 < CELL_NZ) {
            const float* p = pts + 3 * ((ix * CELL_NY + iy) * CELL_NZ + iz);
#pragma unroll
            for (int k = 0; k < 3; ++k) out[k] = mk3(p[3 * k], p[3 * k + 1], p[3 * k + 2]);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) out[k] = (*this)(cx, cy, cz - 1 + k);
        }
    }
};

// One workgroup = one 16-column row of a chunk = 16 x 144 evaluated voxels.  Voxels y >= 144 never need noise: solid iff
// y <= min(max((int)h, 128), ravine cut), so their bits are built analytically.  The air / solid bits of all 16 x 384 voxels go to LDS as
// 64-bit words; runs are extracted with popcount prefixes over those words.
// The noise runs in dense phases over LDS-compacted voxel lists of the WHOLE row (order inside a list is irrelevant: voxels are
// independent): a phase loses at most one partial wave at the end of its list, and a row-long list makes that 2 % of a phase where the
// four-column batches of the previous layout lost 9 % (cave noise) to 22 % (threshold) - and paid four times the barriers.
#ifndef MM_CAVE_WAVES
#define MM_CAVE_WAVES 6
#endif
// THREADS: 256 for launches that fill the chip (a row's ~2 000 listed voxels are nine per thread: 6 workgroups x 4 waves per CU).  A SMALL
// launch - a streaming strip: ~1 200 rows, all resident at once - is as slow as its slowest row, nine dependent warp + Worley evaluations
// deep; k_cave_voxels_wide runs the same row with 512 threads (the phases' loops stride by THREADS, the per-column walk stays with the first
// 256), which halves that depth: 0.19 -> 0.1x ms of a 1.07 ms tick.  Same lists, same LDS, same results (list order is irrelevant).
template <int THREADS>
MM_DEV void cave_voxels_body(const float* __restrict__ hf, const float2* __restrict__ colInfo, const int2* __restrict__ chunkPos,
              mmgen_cave_layer* __restrict__ caveLayers, const int* __restrict__ chunkList, const uint8_t* __restrict__ colNeed /*nullable, lazy ring*/)
{
    __shared__ float s_cells[3 * CELL_N];
    __shared__ unsigned long long s_solid[CAVE_ROW][6];       // solid bit of voxel y at word y / 64, bit y % 64 (set / cleared through its 32-bit halves)
    __shared__ float s_obw[CAVE_ROW], s_ravine[CAVE_ROW];      // the row's per-column info (k_cave_columns)
    __shared__ int s_top[CAVE_ROW];                            // max((int)height, SEA_LEVEL); -1 for a column the lazy ring skips
    __shared__ __attribute__((aligned(16))) unsigned short s_list1[CAVE_VOXELS];
    __shared__ unsigned short s_list2[CAVE_L2_CAP];
    __shared__ float s_thr[CAVE_L2_CAP];
    __shared__ int s_count[3];
    // list 3 lives in list 1's memory once phase B is over: index into list 2 + the voxel's `huge`
    static_assert(CAVE_L3_CAP * (sizeof(unsigned short) + sizeof(float)) <= sizeof(unsigned short) * CAVE_VOXELS, "list 3 fits in list 1");
    unsigned short* s_list3 = s_list1;
    float* s_huge = (float*)(s_list1 + CAVE_L3_CAP);

    const int t = threadIdx.x;
    const int bid = xcd_block(blockIdx.x, gridDim.x);
    const int chunk = chunkList ? chunkList[bid >> 4] : (bid >> 4);
    const int row = bid & 15;                                  // z = row
    const int2 cp = chunkPos[chunk];
    const int colBase = 16 * row;                              // column c of the row: x = c, z = row
    // lazy ring: columns that cannot produce a placement reaching the rectangle get no cave noise: no solid bit is ever set for them,
    // so no flip is found and their 32 layers stay at the default {384, 384, NONE, NONE}
    unsigned rowNeed = 0xffffu;
    if (colNeed) {
        const uint4 nd = *(const uint4*)(colNeed + (size_t)256 * chunk + 16 * row);
        rowNeed = 0u;
        const unsigned w[4] = {nd.x, nd.y, nd.z, nd.w};
        for (int k = 0; k < 16; ++k) rowNeed |= ((w[k >> 2] >> (8 * (k & 3))) & 255u) ? (1u << k) : 0u;
    }
    if (!rowNeed) {                                            // nothing to evaluate: all 16 columns keep the default layers
        for (int i = t; i < 16 * 3 * MMGEN_MAX_CAVE_LAYERS_PER_COLUMN; i += THREADS)
            ((int*)(caveLayers + (size_t)MMGEN_MAX_CAVE_LAYERS_PER_COLUMN * (chunk * 256 + 16 * row)))[i] = ((i % 3) == 2) ? 0 : 384;
        return;
    }

    // cell tile of the row: sample position = noisePos * (1, 1.6, 1) + offset with |offset| < 1.8; origin from the row's first column
    // (16 columns are 0.08 cells wide: at most one cell boundary inside the row)
    CellTile tile;
    tile.pts = s_cells;
    tile.ox = (int)__builtin_floorf(((float)cp.x * 0.0050f) * 1.f) - 3;
    tile.oy = -3;
    tile.oz = (int)__builtin_floorf(((float)(cp.y + row) * 0.0050f) * 1.f) - 3;
    for (int i = t; i < CELL_N; i += THREADS) {
        const int iz = i % CELL_NZ, iy = (i / CELL_NZ) % CELL_NY, ix = i / (CELL_NZ * CELL_NY);
        const f3 p = rand3from3((float)(tile.ox + ix), (float)(tile.oy + iy), (float)(tile.oz + iz));
        s_cells[3 * i] = p.x; s_cells[3 * i + 1] = p.y; s_cells[3 * i + 2] = p.z;
    }
    for (int i = t; i < CAVE_ROW * 6; i += THREADS) s_solid[i / 6][i % 6] = 0ull;
    if (t < CAVE_ROW) {
        const int col = chunk * 256 + colBase + t;
        const float2 ci = colInfo[col];
        s_obw[t] = ci.x; s_ravine[t] = ci.y;
        s_top[t] = ((rowNeed >> t) & 1u) ? imax((int)hf[col], MMGEN_SEA_LEVEL) : -1;
    }
    if (t < 3) s_count[t] = 0;
    noise_tables_init<false>();                                // no simplex2 in this kernel; ends with the workgroup barrier

    // Ocean columns need the noise only below y ~ 92 and most voxels that pass the first test pass the second, so without compaction
    // 4 - 40 % of the lanes idle through the 23 simplex evaluations.
    //   A  every voxel: everything that needs no noise; solid bit set as if the noise said "no cave"; voxels that need it -> list 1
    //   B  list 1: position warp (fbm3from3<5>) + Worley = the cave noise; voxels whose noise is below the largest threshold the voxel can
    //      have -> list 2 with their noise
    //   C  list 2: `huge` (<= 4 octaves) and the bound it implies; voxels whose noise is still below it -> list 3 with their huge
    //   D  list 3: the threshold itself (fbm3<4>); "cave" clears the solid bit again.  (With four-column batches this fourth phase measured
    //      0 %: its partial wave cost what it saved.  With row-long lists a third of phase C's lanes no longer idle through four octaves.)
    // every slot's default {384, 384, biomes 0}, whole lines, long before the runs overwrite a few of them (the barriers in between order the stores)
    {
        int* rowLayersEarly = (int*)(caveLayers + (size_t)MMGEN_MAX_CAVE_LAYERS_PER_COLUMN * (chunk * 256 + colBase));
        for (int i = t; i < CAVE_ROW * 3 * MMGEN_MAX_CAVE_LAYERS_PER_COLUMN; i += THREADS) rowLayersEarly[i] = ((i % 3) == 2) ? 0 : 384;
    }
    // analytic part, y in [144, 384): solid iff y <= topSolid and not (y > ravineY)   (topRatio == 0 there)
    if (t < CAVE_ROW * 4) {    // 4 lanes per column fill words 2..5 (word 2 holds y 128..191: bits >= 16 only)
        const int c = t >> 2, w = 2 + (t & 3);
        const float ravineY = s_ravine[c];
        // (float)y > ravineY  <=>  y > floor(ravineY) for an integer y; no ravine = +inf
        const int cut = ravineY >= 383.f ? 383 : (int)__builtin_floorf(ravineY);
        const int lo = imax(CAVE_YEVAL, 64 * w) - 64 * w, hi = imin(imin(s_top[c], cut), 64 * w + 63) - 64 * w;      // bit range inside the word
        if (hi >= lo) {
            // through the same 32-bit halves as the walk below, which ORs y 128 .. 143 into the low half of word 2 with no barrier in between
            const unsigned long long bits = (~0ull >> (63 - hi)) & (~0ull << lo);
            unsigned* half = (unsigned*)&s_solid[c][w];
            if ((unsigned)bits) atomicOr(half, (unsigned)bits);
            if ((unsigned)(bits >> 32)) atomicOr(half + 1, (unsigned)(bits >> 32));
        }
    }
    const bool prune = prune_domain(cp.x, cp.y + row);          // the row's 16 columns: x in [cp.x, cp.x + 15] (cp.x a multiple of 16), z = cp.y + row
    const float b3 = prune ? MM_SIMPLEX3_BOUND : 3.402823466e+38f;
    const float kCaveFaMax = prune ? 0.9375f * MM_SIMPLEX3_BOUND : 1e30f;      // outside the domain: a bound no noise reaches
    static_assert(THREADS >= 256 && THREADS % 64 == 0 && CAVE_ROW == 16 && CAVE_YEVAL == 144, "the walk below: thread t < 256 = column t % 16, y = t / 16 + 16 i, i < 9");
    if (THREADS == 256 || t < 256) {
    // y-major walk (the lists come out ordered by depth): a thread keeps its column, a wave covers four consecutive y of the 16 columns
    // per step.  Nothing touches LDS inside the walk: the solid bits gather in registers (y + 16 i lies in 32-bit word i / 2), the four
    // lanes of a column OR theirs together and one of them writes the words; the list positions come from ballots, with ONE counter
    // update per wave - per step that was 64 conflicting LDS atomics and a counter round trip.
    const int c = t % CAVE_ROW, yb = t / CAVE_ROW, lane = t & 63;
    const float obw = s_obw[c], ravineY = s_ravine[c];
    const int topSolid = s_top[c];
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned bitsW[5] = {0u, 0u, 0u, 0u, 0u};
    unsigned listed = 0u;                                       // bit i: voxel y = yb + 16 i needs the noise
    unsigned char posIn[9];                                     // ... and is the posIn[i]-th such voxel of the wave's step i
    int nBefore = 0;                                            // wave-uniform: list entries of the wave's steps so far
    int before[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int y = yb + 16 * i;
        bool need = false;
        if (topSolid >= 0) {
            const float fy = (float)y;
            const bool inBand = (y != 0) && (y <= topSolid);
            const float topRatio = smoothstep(142.f, 95.f, fy + obw * 50.f);
            // the largest threshold the voxel can have (phase B's `bound`, the same expression): at or below 0.04 there is no noise cave
            // whatever the noise is, so the voxel never enters list 1 (the top few blocks of the band, where topRatio is tiny).
            // y = yb + 16 i >= 32 for i >= 2, where smoothstep(5, 20, y) is exactly 1
            const float bottomRatio = i >= 2 ? 1.f : smoothstep(5.f, 20.f, fy);
            float bound = 0.24f + 0.12f * kCaveFaMax;
            bound *= (1.f + 1.4f * 1.f);
            bound *= topRatio * (0.3f + 0.7f * bottomRatio);
            const bool needThr = inBand && bound > 0.04f;
            // final cave = noise cave || (y != 0 && !inBand) || (inBand && fy > ravineY)   (y == 0 solid, y > topSolid air, ravine cut)
            const bool cave0 = ((y != 0) && !inBand) || (inBand && fy > ravineY);
            if (!cave0) {
                bitsW[i >> 1] |= 1u << (y & 31);
                need = needThr;                                 // a voxel that is a cave anyway needs no noise
            }
        }
        const unsigned long long m = __ballot(need);
        if (need) listed |= 1u << i;
        posIn[i] = (unsigned char)__popcll(m & below);
        before[i] = nBefore;
        nBefore += __popcll(m);
    }
    // solid bits: the column's four lanes of this wave (lane, lane ^ 16, lane ^ 32, lane ^ 48) together, lanes 0 .. 15 write
#pragma unroll
    for (int w = 0; w < 5; ++w) {
        unsigned b = bitsW[w];
        b |= (unsigned)__shfl_xor((int)b, 16);
        b |= (unsigned)__shfl_xor((int)b, 32);
        if (lane < 16 && b) atomicOr((unsigned*)&s_solid[c][0] + w, b);
    }
    // list 1: one reservation per wave, entries in step order
    int base = 0;
    if (lane == 0 && nBefore) base = atomicAdd(&s_count[0], nBefore);
    base = __shfl(base, 0);
#pragma unroll
    for (int i = 0; i < 9; ++i)
        if ((listed >> i) & 1u) s_list1[base + before[i] + posIn[i]] = (unsigned short)(c * CAVE_YEVAL + yb + 16 * i);
    }
    __syncthreads();
    // The reference evaluates  cave = threshold > 0.04 && caveNoise < threshold  with threshold = ((0.24 + 0.12 fa) (1 + 1.4 huge)) T,
    // fa = fbm3<4>, huge in [0, 1], T = topRatio (0.3 + 0.7 bottomRatio) >= 0.  Every operation of that expression is monotone in fa and
    // in huge (IEEE rounding is monotone, the other factors are non-negative), so the SAME expression with fa := kCaveFaMax >= sup |fbm3<4>|
    // = 0.9375 * MM_SIMPLEX3_BOUND and huge := 1 is >= threshold bit for bit: a voxel whose cave noise is not below that bound is solid whatever the
    // two fbm3<4> are, and they are never evaluated for it (56 % of the voxels; half of the rest is decided once huge is known).
    struct VoxelTerms { int c, y; float npx, npy, npz, T; };      // T = topRatio (0.3 + 0.7 bottomRatio)
    auto terms = [&](int e) {
        VoxelTerms v;
        v.c = e / CAVE_YEVAL; v.y = e - v.c * CAVE_YEVAL;
        const int idx2d = colBase + v.c;
        const float obw = s_obw[v.c];
        const int wx = cp.x + (idx2d & 15), wz = cp.y + (idx2d >> 4);
        v.npx = (float)wx * 0.0050f; v.npz = (float)wz * 0.0050f;
        const float fy = (float)v.y;
        v.npy = fy * 0.0050f;
        const float topRatio = cave_top_ratio(fy + obw * 50.f);
        const float bottomRatio = cave_bottom_ratio(fy);
        v.T = topRatio * (0.3f + 0.7f * bottomRatio);
        return v;
    };
    // phase C for one voxel (its noise n is below the largest threshold it can have): false = solid whatever fbm3<4> is
    auto below_huge_bound = [&](const VoxelTerms& v, float n, float& huge) {
        huge = cave_huge(v.npx * 0.0700f, v.npy * 0.0700f, v.npz * 0.0700f, b3);
        float bound = 0.24f + 0.12f * kCaveFaMax;
        bound *= (1.f + 1.4f * huge);
        bound *= v.T;
        return n < bound;
    };
    // phase D for one voxel
    auto carve = [&](const VoxelTerms& v, float n, float huge) {
        float thr = 0.24f + 0.12f * fbm3<4>(v.npx * 4.f, v.npy * 4.f, v.npz * 4.f);
        thr *= (1.f + 1.4f * huge);
        thr *= v.T;
        if (thr > 0.04f && n < thr) atomicAnd((unsigned*)&s_solid[v.c][0] + (v.y >> 5), ~(1u << (v.y & 31)));
    };
    auto resolve = [&](int e, float n) {                       // C + D in place (a list was full)
        const VoxelTerms v = terms(e);
        float huge;
        if (below_huge_bound(v, n, huge)) carve(v, n, huge);
    };
    const int count1 = s_count[0];
    for (int i = t; i < count1; i += THREADS) {
        const int e = s_list1[i];
        const int c = e / CAVE_YEVAL, y = e - c * CAVE_YEVAL;
        const int idx2d = colBase + c;
        const float obw = s_obw[c];
        const int wx = cp.x + (idx2d & 15), wz = cp.y + (idx2d >> 4);
        const float npx = (float)wx * 0.0050f, npz = (float)wz * 0.0050f;
        const float fy = (float)y;
        const float npy = fy * 0.0050f;
        const float topRatio = cave_top_ratio(fy + obw * 50.f);
        const float bottomRatio = cave_bottom_ratio(fy);
        float bound = 0.24f + 0.12f * kCaveFaMax;
        bound *= (1.f + 1.4f * 1.f);
        bound *= topRatio * (0.3f + 0.7f * bottomRatio);
        if (!(bound > 0.04f)) continue;                        // threshold <= bound <= 0.04: no noise cave
        const f3 o = fbm3from3<5>(npx * 0.8000f, npy * 0.8000f, npz * 0.8000f);
        const float n = special_cave_noise(npx * 1.f + o.x * 1.8f, npy * 1.6f + o.y * 1.8f, npz * 1.f + o.z * 1.8f, tile);
        if (n < bound) {
            const int k = atomicAdd(&s_count[1], 1);
            if (k < CAVE_L2_CAP) { s_list2[k] = (unsigned short)e; s_thr[k] = n; }
            else resolve(e, n);                                // list 2 is full (more than half of the row's voxels got here): in place
        }
    }
    __syncthreads();
    const int count2 = imin(s_count[1], CAVE_L2_CAP);
    for (int i = t; i < count2; i += THREADS) {
        const VoxelTerms v = terms(s_list2[i]);
        float huge;
        if (!below_huge_bound(v, s_thr[i], huge)) continue;
        const int k = atomicAdd(&s_count[2], 1);
        if (k < CAVE_L3_CAP) { s_list3[k] = (unsigned short)i; s_huge[k] = huge; }
        else carve(v, s_thr[i], huge);
    }
    __syncthreads();
    const int count3 = imin(s_count[2], CAVE_L3_CAP);
    for (int k = t; k < count3; k += THREADS) {
        const int i = s_list3[k];
        carve(terms(s_list2[i]), s_thr[i], s_huge[k]);
    }
    __syncthreads();

    // flips: solid(y) != solid(y+1), y = 383 compares with "not solid"; the r-th flip of a column (counted from below) is the start
    // (r even) or the end (r odd) of layer r / 2; runs beyond 32 layers are dropped (canonical).  All 16 columns' slots get their default
    // {384, 384, biomes 0} with whole-line stores at the start of the kernel (several barriers before this point); here one thread per
    // (column, 64-voxel word) walks the set bits of its flip word and overwrites the slots its flips belong to (a column has a handful).
    int* rowLayers = (int*)(caveLayers + (size_t)MMGEN_MAX_CAVE_LAYERS_PER_COLUMN * (chunk * 256 + colBase));
    if (t < CAVE_ROW * 6) {
        const int c = t / 6, w = t % 6;
        auto flips = [&](int k) {
            const unsigned long long m = s_solid[c][k];
            const unsigned long long nl = (k < 5) ? (s_solid[c][k + 1] & 1ull) : 0ull;
            return m ^ ((m >> 1) | (nl << 63));
        };
        int rank = 0;
        for (int k = 0; k < w; ++k) rank += __popcll(flips(k));
        unsigned long long f = flips(w);
        int* colLayers = rowLayers + 3 * MMGEN_MAX_CAVE_LAYERS_PER_COLUMN * c;
        while (f && rank < 2 * MMGEN_MAX_CAVE_LAYERS_PER_COLUMN) {
            const int b = (int)__builtin_ctzll(f);
            f &= f - 1ull;
            colLayers[3 * (rank >> 1) + (rank & 1)] = 64 * w + b;
            ++rank;
        }
    }
}

__attribute__((amdgpu_waves_per_eu(MM_CAVE_WAVES, MM_CAVE_WAVES)))
__global__ void __launch_bounds__(CAVE_THREADS)
k_cave_voxels(const float* __restrict__ hf, const float2* __restrict__ colInfo, const int2* __restrict__ chunkPos,
              mmgen_cave_layer* __restrict__ caveLayers, const int* __restrict__ chunkList, const uint8_t* __restrict__ colNeed)
{
    cave_voxels_body<CAVE_THREADS>(hf, colInfo, chunkPos, caveLayers, chunkList, colNeed);
}

#define CAVE_THREADS_WIDE 512
#ifndef CAVE_WIDE_MAX_ROWS
#define CAVE_WIDE_MAX_ROWS 3072                     // launches of at most this many rows (192 chunks) take the wide kernel: two rounds of the chip's 1 536 workgroups
#endif
__attribute__((amdgpu_waves_per_eu(MM_CAVE_WAVES, MM_CAVE_WAVES)))
__global__ void __launch_bounds__(CAVE_THREADS_WIDE)
k_cave_voxels_wide(const float* __restrict__ hf, const float2* __restrict__ colInfo, const int2* __restrict__ chunkPos,
                   mmgen_cave_layer* __restrict__ caveLayers, const int* __restrict__ chunkList, const uint8_t* __restrict__ colNeed)
{
    cave_voxels_body<CAVE_THREADS_WIDE>(hf, colInfo, chunkPos, caveLayers, chunkList, colNeed);
}

// Cave biomes of the layers' end blocks: at most 2 getCaveBiome evaluations per occupied layer slot, and only a few of a column's
// 32 slots are occupied.  Persistent waves, no workgroup barrier after the tables are staged (k_fill_cave's scheme): a wave draws UNITS
// of 64 columns from work counters, walks their slots (lane = column, slot by slot until no column has one left), and streams the
// (column, slot, bottom | top) evaluations that exist through two per-wave LDS buffers: stage 1 = the warped height of 64 items (far
// above the depth bands it settles the biome: NONE), stage 2 = the rest of the evaluation for 64 survivors.  Every stage runs with full
// waves whatever unit its items came from.
#define CB_THREADS 256
#define CB_UNIT_COLS 64
#define CB_COUNTERS 16
#define CB_S1_CAP 192      // < 64 waiting + at most 128 new ones per slot round
#define CB_S2_CAP 128
#ifndef MM_CB_WAVES
#define MM_CB_WAVES 6
#endif
// UNIT_COLS: 64 (lane = column, one slot per step) for launches that give every wave many units; 16 for small ones (a streaming strip has
// ~300 units of 64 columns for 1 024 waves: one round, as long as its slowest unit): a unit is then 16 columns and a step covers FOUR
// consecutive slots of them (lane = column + 16 * slot offset), so there are four times the units, each a quarter of the walk.  The items
// are the same; only which wave evaluates them changes.
template <int UNIT_COLS>
MM_DEV void cave_biomes_body(const float* __restrict__ hf, const int2* __restrict__ chunkPos, mmgen_cave_layer* __restrict__ caveLayers,
              const int* __restrict__ chunkList, int nUnits, unsigned* __restrict__ work)
{
    __shared__ uint2 s_s1[CB_THREADS / 64][CB_S1_CAP];         // .x = list index << 14 | column << 6 | slot << 1 | top, .y = y of the block
    __shared__ uint2 s_s2[CB_THREADS / 64][CB_S2_CAP];
    __shared__ float s_py[CB_THREADS / 64][CB_S2_CAP];         // stage 2: the item's warped height
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    noise_tables_init();                                       // once per (persistent) workgroup; no workgroup barrier after this one
    uint2* s1 = s_s1[wave];
    uint2* s2 = s_s2[wave];
    float* s2py = s_py[wave];
    int n1 = 0, n2 = 0;                                        // wave-uniform
    const unsigned long long below = (1ull << lane) - 1ull;

    struct Item { int chunk, col, wx, wz, seed; float maxHeight; bool near; };
    auto item_of = [&](unsigned id) {
        Item it;
        const int li = (int)(id >> 14);
        it.chunk = chunkList ? chunkList[li] : li;
        it.col = (id >> 6) & 255;
        const int2 cp = chunkPos[it.chunk];
        it.wx = cp.x + (it.col & 15); it.wz = cp.y + (it.col >> 4);
        it.maxHeight = hf[(size_t)256 * it.chunk + it.col];
        it.seed = (id & 1u) ? 4982921 : 329271348;
        it.near = prune_domain(it.wx, it.wz);
        return it;
    };
    auto store = [&](unsigned id, const Item& it, int biome) {
        mmgen_cave_layer* l = caveLayers + ((size_t)256 * it.chunk + it.col) * MMGEN_MAX_CAVE_LAYERS_PER_COLUMN + ((id >> 1) & 31);
        if (id & 1u) l->top_biome = (uint8_t)biome; else l->bottom_biome = (uint8_t)biome;
    };

    int part = (int)((CB_THREADS / 64) * blockIdx.x + wave) % CB_COUNTERS, dry = 0;
    (void)dry;
    unsigned drawn = 0u;
    if (lane == 0) drawn = atomicAdd(&work[16 * part], 1u);
    bool more = true;                                          // units left to draw
    int unit = -1, slot = MMGEN_MAX_CAVE_LAYERS_PER_COLUMN;    // the unit being walked and its next slot round (32 = done)
    for (;;) {
        if (n2 >= 64 || (!more && n1 == 0 && n2 > 0)) {
            // stage 2: the rest of getCaveBiome for items whose warped height did not settle it
            const int n = imin(n2, 64);
            wave_lds_sync();
            if (lane < n) {
                const uint2 d = s2[n2 - n + lane];
                const float py = s2py[n2 - n + lane];
                const Item it = item_of(d.x);
                store(d.x, it, it.near ? cave_biome_rest<true>(it.wx, (int)d.y, it.wz, it.maxHeight, it.seed, true, false, py)
                                       : cave_biome_rest<false>(it.wx, (int)d.y, it.wz, it.maxHeight, it.seed, true, false, py));
            }
            n2 -= n;
            continue;
        }
        if (n1 >= 64 || (!more && n1 > 0)) {
            // stage 1: the warped height (3 simplex3, the same work in every lane)
            const int n = imin(n1, 64);
            wave_lds_sync();
            bool on = false;
            uint2 d = make_uint2(0u, 0u);
            float py = 0.f;
            if (lane < n) {
                d = s1[n1 - n + lane];
                const Item it = item_of(d.x);
                const bool none = it.near ? cave_biome_py<true>(it.wx, (int)d.y, it.wz, it.maxHeight, true, py)
                                          : cave_biome_py<false>(it.wx, (int)d.y, it.wz, it.maxHeight, true, py);
                if (none) store(d.x, it, MMCB_NONE); else on = true;
            }
            n1 -= n;
            const unsigned long long om = __ballot(on);
            if (on) { const int at = n2 + __popcll(om & below); s2[at] = d; s2py[at] = py; }
            n2 += __popcll(om);
            continue;
        }
        if (!more) break;
        if (slot >= MMGEN_MAX_CAVE_LAYERS_PER_COLUMN) {              // next unit
            const int u = __builtin_amdgcn_readfirstlane((int)drawn) * CB_COUNTERS + part;
            if (u >= nUnits) {
#if MM_COUNTER_PROBE_LOADS
                const int nx = next_live_counter(work, CB_COUNTERS, part, nUnits);
                if (nx < 0) { more = false; continue; }
                part = nx;
#else
                if (++dry == CB_COUNTERS) { more = false; continue; }
                part = (part + 1) % CB_COUNTERS;
#endif
            }
            if (lane == 0) drawn = atomicAdd(&work[16 * part], 1u);  // the next draw is in flight while this unit is walked
            if (u >= nUnits) continue;
            unit = u; slot = 0;
        }
        // one slot of the unit's 64 columns (lane = column): a used slot has one item (its bottom block) or two (a layer open to the sky
        // has no top block: top biome NONE); used slots come first in a column, the walk ends when no column has one left
        constexpr int unitsPerChunk = 256 / UNIT_COLS, slotsPerStep = 64 / UNIT_COLS;
        static_assert(UNIT_COLS == 64 || UNIT_COLS == 16, "lane = column + UNIT_COLS * slot offset");
        const int li = unit / unitsPerChunk, col = UNIT_COLS * (unit % unitsPerChunk) + lane % UNIT_COLS;
        const int mySlot = slot + lane / UNIT_COLS;
        const bool inRange = UNIT_COLS == 64 || mySlot < MMGEN_MAX_CAVE_LAYERS_PER_COLUMN;
        const int chunk = chunkList ? chunkList[li] : li;
        mmgen_cave_layer* l = caveLayers + ((size_t)256 * chunk + col) * MMGEN_MAX_CAVE_LAYERS_PER_COLUMN + (inRange ? mySlot : 0);
        const int start = inRange ? l->start : 384, end = l->end;
        const bool used = start != 384, two = used && end != 384;
        const unsigned long long um = __ballot(used), tm = __ballot(two);
        if (um == 0ull) { slot = MMGEN_MAX_CAVE_LAYERS_PER_COLUMN; continue; }      // (used slots come first in a column: none in this step, none later)
        if (used) {
            if (!two) l->top_biome = (uint8_t)MMCB_NONE;
            const unsigned id = ((unsigned)li << 14) | ((unsigned)col << 6) | ((unsigned)mySlot << 1);
            const int at = n1 + __popcll(um & below) + __popcll(tm & below);
            s1[at] = make_uint2(id, (unsigned)start);
            if (two) s1[at + 1] = make_uint2(id | 1u, (unsigned)(end + 1));
        }
        n1 += __popcll(um) + __popcll(tm);
        slot += slotsPerStep;
    }
}

__attribute__((amdgpu_waves_per_eu(MM_CB_WAVES, MM_CB_WAVES)))
__global__ void __launch_bounds__(CB_THREADS)
k_cave_biomes(const float* __restrict__ hf, const int2* __restrict__ chunkPos, mmgen_cave_layer* __restrict__ caveLayers,
              const int* __restrict__ chunkList, int nUnits, unsigned* __restrict__ work)
{
    cave_biomes_body<CB_UNIT_COLS>(hf, chunkPos, caveLayers, chunkList, nUnits, work);
}

#define CB_UNIT_COLS_SMALL 16
#ifndef CB_SMALL_MAX_CHUNKS
#define CB_SMALL_MAX_CHUNKS 512                      // launches of at most this many chunks take the 16-column units
#endif
__attribute__((amdgpu_waves_per_eu(MM_CB_WAVES, MM_CB_WAVES)))
__global__ void __launch_bounds__(CB_THREADS)
k_cave_biomes_small(const float* __restrict__ hf, const int2* __restrict__ chunkPos, mmgen_cave_layer* __restrict__ caveLayers,
                    const int* __restrict__ chunkList, int nUnits, unsigned* __restrict__ work)
{
    cave_biomes_body<CB_UNIT_COLS_SMALL>(hf, chunkPos, caveLayers, chunkList, nUnits, work);
}

// =========================================================================================================
// K6 — fill (shared pieces; the kernels are further down).  Per voxel:
//   1. every voxel gets its base block (bedrock / air / water / cave air / layer material + surface-biome rules): cheap, lane = y;
//   2. the voxels whose block a cave biome could still alter (STONE / DEEPSLATE / BLACKSTONE below ground — the only blocks
//      caveBiomeBlockPostProcess touches) are compacted into an LDS list and processed densely: the cave-biome evaluation
//      (9 simplex3D + 12 simplex2D) is the expensive part of fill, and a dense list keeps every lane of every wave on it instead
//      of interleaving it with lanes that returned "air" long ago.  Order inside the list is irrelevant (voxels are independent).
// =========================================================================================================
// ---- cave voxels of a column as a 384-bit mask: bit y = voxel y is inside a cave layer (start < y <= end).  Eight 64-bit words per
// column, the mask in words 1 .. 6 between two words of zeros, so that the 32-bit windows below never leave the column's words.
#define CAVE_MASK_WORDS 8
MM_DEV void cave_mask_add(unsigned long long* colMask /*LDS, zeroed*/, int start, int end)
{
    if (start == 384) return;
    const int lo = start + 1, hi = imin(end, 383);
#pragma unroll
    for (int w = 0; w < 6; ++w) {
        const int a = imax(lo, 64 * w), b = imin(hi, 64 * w + 63);
        if (a <= b) atomicOr(&colMask[1 + w], (~0ull >> (63 - (b - a))) << (a - 64 * w));
    }
}
MM_DEV bool cave_mask_test(const unsigned long long* colMask, int y) { return (((const unsigned*)colMask)[2 + (y >> 5)] >> (y & 31)) & 1u; }
// the two cave-surface distances of chunkFillPlaceBlock's layer walk (chunk.cu:1264-1292) for a voxel OUTSIDE every cave, as the 5-bit
// codes the cave-biome rules read them through: caveBottomDepth = start - y of the first layer that starts at or above y = the number of
// solid voxels between y and the next cave voxel above; caveTopDepth = y - (end + 1) of the last layer that ends below y = the same
// downwards.  The rules only ask "== 0" and "0 .. threshold" with threshold = 1.5 + 4.5 * simplex3 < 17.25 (|simplex3| < 3.5 by the
// crudest bound: 42 * 4 corners * max((0.6 - r^2)^4 r)), so a distance of 30 or more, or no cave at all in that direction (negative in the
// reference), are the same to them: code 31 = nothing within the 32 voxels of the window.
MM_DEV void depth_codes(const unsigned long long* colMask, int y, unsigned& bdc, unsigned& tdc)
{
    const unsigned* w = (const unsigned*)colMask + 2;                // word i = voxels 32 i .. 32 i + 31; w[-2], w[-1], w[12], w[13] are zero
    const int iu = (y + 1) >> 5, su = (y + 1) & 31;                  // voxels y + 1 .. y + 32, bit 0 = y + 1
    const unsigned up = __funnelshift_r(w[iu], w[iu + 1], su);
    const int id = (y - 32) >> 5, sd = (y - 32) & 31;                // voxels y - 32 .. y - 1, bit 31 = y - 1 (arithmetic shift: -1 for y < 32)
    const unsigned dn = __funnelshift_r(w[id], w[id + 1], sd);
    bdc = up ? (unsigned)imin(__builtin_ctz(up), 30) : 31u;
    tdc = dn ? (unsigned)imin(__builtin_clz(dn), 30) : 31u;
}

struct BaseBlock { uint8_t block; bool needCave; int bottomDepth, topDepth; };

// The column's biome weights in the compact form getRandomBiome (biomeFuncs.hpp:39-53) can be walked in: biome 0 (returned by a draw of
// exactly 0 whatever its weight) plus the biomes of positive weight, ascending.  Subtracting a zero weight changes nothing and cannot
// trigger the "<= 0" exit of a positive remainder, so the walk returns exactly what the 24-entry walk returns - in 2 - 4 steps, not 10 - 23.
#define FILL_NZ_CAP 8
// drawMinY / drawWater: where the biome drawn for a voxel can matter at all.  Besides the grass block of the top voxel, the drawn biome only
// enters through biomeBlockPreProcess / biomeBlockPostProcess (biomeFuncs.hpp:385-600), and those act for nine biomes only, each from a
// fixed height up (or on water): in a column where none of them has weight, or below the lowest of their thresholds, every biome
// gives the same block and the draw (a hash + a walk over the weights per voxel, ~85 % of the voxels of a generated world) is skipped.
struct ColumnBiomes { int n; bool isOcean; const uint8_t* idx; const float* w; const float* all; int drawMinY; bool drawWater; bool layersSorted; };

// lowest y at which biome b changes a block in biome_block_pre / biome_block_post other than through isTop or WATER (384 = never)
MM_DEV int biome_rule_min_y(int b, float height)
{
    switch (b) {
    case MMBIO_ARCHIPELAGO: return MMGEN_SEA_LEVEL;           // wy < SEA_LEVEL returns
    case MMBIO_MESA: return 90;                               // fy < 90 returns
    case MMBIO_SHREKS_SWAMP: return 100;
    case MMBIO_TIANZI_MOUNTAINS: return 90;
    case MMBIO_MOUNTAINS: return 190;
    case MMBIO_CRYSTALS: return height > 176.f ? 0 : 384;     // quartz above a noisy start height (biome_block_pre); its post rule is isTop only
    default: return 384;                                      // TROPICAL_BEACH, BEACH: isTop only; FROZEN_WASTELAND: WATER only; the rest: none
    }
}
MM_DEV int random_biome(const ColumnBiomes& cb, float rand)
{
    if (cb.n > FILL_NZ_CAP) return random_biome(cb.all, 1, rand);          // more than 8 biomes meet in this column: the plain walk
    for (int k = 0; k < cb.n; ++k) {
        rand -= cb.w[k];
        if (rand <= 0.f) return cb.idx[k];
    }
    return MMBIO_PLAINS;
}

// kMaterialBlock[layer] as immediates (8 bits per material): indexed per lane the table is a dependent memory round trip per voxel
MM_DEV uint8_t material_block(int layer)
{
    static_assert(MMGEN_NUM_MATERIALS <= 24, "three 64-bit words");
    unsigned long long w[3] = {0ull, 0ull, 0ull};
#pragma unroll
    for (int m = 0; m < MMGEN_NUM_MATERIALS; ++m) w[m >> 3] |= (unsigned long long)kMaterialBlock[m] << (8 * (m & 7));
    const unsigned long long v = layer < 8 ? w[0] : (layer < 16 ? w[1] : w[2]);
    return (uint8_t)((v >> (8 * (layer & 7))) & 255ull);
}

// MASKED (k_fill_base): s_cl is the column's 384-bit mask of cave voxels (six 64-bit words, cave_mask_add) instead of its layer list, the
// two cave-surface distances are left to the caller (depth_codes, only for the voxels that need them).
template <bool MASKED = false, class CaveT = mmgen_cave_layer>
MM_DEV BaseBlock place_block_base(const ColumnBiomes& cbi, const float* s_lh, const CaveT* s_cl, int y, float height, int wx, int wz)
{
    BaseBlock r; r.needCave = false; r.bottomDepth = -384; r.topDepth = -384;
    if (y == 0) { r.block = MMB_BEDROCK; return r; }
    const float fy = (float)y;
    if (fy > height && y > MMGEN_SEA_LEVEL) { r.block = MMB_AIR; return r; }

    const bool isOcean = cbi.isOcean;

    const bool isTop = fy >= height - 1.f;
    const bool isWater = fy > height && y <= MMGEN_SEA_LEVEL;
    int randBiome = MMBIO_PLAINS;                             // a biome without rules
    if (isTop || y >= cbi.drawMinY || (isWater && cbi.drawWater)) {
        MinStd rng = rng3(wx, y, wz);
        randBiome = random_biome(cbi, rng.u01());
    }

    uint8_t block = MMB_AIR;
    if (isWater) {
        block = MMB_WATER;
        biome_block_post(block, randBiome, wx, y, wz, isTop);
        if (isOcean) { r.block = block; return r; }
    }

    int bottomDepth = -384, topDepth = -384;
    if constexpr (MASKED) {
        // inside a cave: air or lava (see below)
        if (cave_mask_test(s_cl, y)) { r.block = (y <= MMGEN_LAVA_LEVEL) ? MMB_LAVA : MMB_AIR; return r; }
    } else
    for (int k = 0; k < MMGEN_MAX_CAVE_LAYERS_PER_COLUMN; ++k) {
        const int start = s_cl[k].start, end = s_cl[k].end;
        if (start == 384) { bottomDepth = -384; break; }
        bottomDepth = start - y;
        if (y <= start) break;
        if (y <= end) {
            // inside a cave: air or lava.  caveBiomeBlockPostProcess never alters AIR/LAVA (every rule requires
            // STONE/DEEPSLATE/BLACKSTONE), so the cave biome is not evaluated here.
            r.block = (y <= MMGEN_LAVA_LEVEL) ? MMB_LAVA : MMB_AIR;
            return r;
        }
        topDepth = y - (end + 1);
    }

    if (fy > height) { r.block = block; return r; }

    if (biome_block_pre(block, randBiome, wx, y, wz, height)) {
        biome_block_post(block, randBiome, wx, y, wz, isTop);
        r.block = block;
        return r;
    }

    const int l0 = (fy >= s_lh[MMGEN_NUM_FORWARD_MATERIALS]) ? MMGEN_NUM_FORWARD_MATERIALS : 0;
    int layer = -1;
    if (cbi.layersSorted) {
        // Both runs of layer starts - forward 0 .. 9, backward + eroded 10 .. 20 (the last entry is the height) - are non-decreasing in
        // nearly every column.  Then the first l >= l0 with s_lh[l] <= fy < s_lh[l + 1] is the last layer of fy's run that starts at or
        // below fy (earlier ones end at or below fy, later ones start above it, and for l0 = 0 the second run starts above fy altogether):
        // nine independent compares instead of a data-dependent walk of up to twenty
        int cnt = 0;
#pragma unroll
        for (int i = 1; i < MMGEN_NUM_FORWARD_MATERIALS; ++i) cnt += (s_lh[l0 + i] <= fy) ? 1 : 0;
        const int l = l0 + cnt;
        if (s_lh[l0] <= fy && fy < s_lh[l + 1]) layer = l;
    } else
    for (int l = l0; l < MMGEN_NUM_MATERIALS; ++l) {
        if (s_lh[l] <= fy && fy < s_lh[l + 1]) { layer = l; break; }
    }
    block = (layer < 0) ? (uint8_t)MMB_STONE : material_block(layer);   // canonical: no layer (y == height exactly) → STONE
    if (isTop && block == MMB_DIRT) block = kGrassBlock[randBiome];

    biome_block_post(block, randBiome, wx, y, wz, isTop);
    r.block = block;
    r.needCave = cave_post_can_apply(block);
    r.bottomDepth = bottomDepth;
    r.topDepth = topDepth;
    return r;
}

#define FILL_ROW 16          // columns per workgroup: one row of the chunk, staged with whole-line loads
#define FILL_VOX (FILL_ROW * 384)     // 6 144 voxels per row
#define FILL_VBITS 13        // bits of a voxel's position in the row
#define FILL_VMASK ((1 << FILL_VBITS) - 1)
#define FILL_L2_CAP 2048     // voxels of a row whose cave biome has a noise rule (typically a few hundred); a full list evaluates the surplus in place
#define FILL_L3_CAP 512      // deferred lush voxels per row (typically 0 - 100)
#ifndef FILL_THREADS
#define FILL_THREADS 512     // 8 waves: three workgroups (50 KB of LDS each) fill a CU's 24 wave slots
#endif
#ifndef MM_FILL_WAVES
#define MM_FILL_WAVES 6
#endif

// NEAR = the rows inside the pruning domain (mm_noise.cuh), evaluated with the exact prunings; k_fill_far takes the rows beyond it the
// plain way.  Two kernels rather than a flag: the pruned path sits exactly at its register budget, and a row belongs to one of them.
//
// One workgroup = one 16-column row of a chunk, its voxel lists row-long (like k_cave_voxels): a dense phase loses at most one partial
// wave at the end of its list - with the 4-column batches of the previous layout that was 12 % of the cave-biome phase, with a row's
// ~1 800 stone voxels it is 2 % - and the row pays 5 workgroup barriers instead of 20.
template <bool NEAR>
MM_DEV void fill_body(const float* __restrict__ hf, const float* __restrict__ bw, const float* __restrict__ layers,
                      const mmgen_cave_layer* __restrict__ caveLayers, const int2* __restrict__ chunkPos, uint8_t* __restrict__ blocks,
                      const int* __restrict__ srcIdx, unsigned* __restrict__ lushQueue /*[0] = count, entries from [1]; nullable*/, unsigned lushCap)
{
    {
        const int bid0 = xcd_block(blockIdx.x, gridDim.x);
        const int2 cp0 = chunkPos[srcIdx ? srcIdx[bid0 >> 4] : (bid0 >> 4)];
        if (prune_domain(cp0.x, cp0.y + (bid0 & 15)) != NEAR) return;       // (the row's x range is [cp.x, cp.x + 15], cp.x a multiple of 16)
    }
    noise_tables_init();
    __shared__ float s_bw[FILL_ROW][MMGEN_NUM_BIOMES];
    __shared__ float s_lh[FILL_ROW][MMGEN_NUM_MATERIALS + 1];
    __shared__ mmgen_cave_layer s_cl[FILL_ROW][MMGEN_MAX_CAVE_LAYERS_PER_COLUMN];
    __shared__ unsigned int s_list[FILL_VOX];                 // voxel (FILL_VBITS) | block (8) | bottomDepth code (5) | topDepth code (5), from the low bits up
    __shared__ unsigned short s_list2[FILL_L2_CAP];           // index into s_list (FILL_VBITS) | isLush << FILL_VBITS: voxels whose cave biome has a noise rule
    __shared__ unsigned short s_list3[FILL_L3_CAP];           // index into s_list: lush voxels close enough to a cave surface for clay / moss
    __shared__ int s_count[3];
    __shared__ unsigned s_qbase;
    __shared__ float s_nzW[FILL_ROW][FILL_NZ_CAP];
    __shared__ uint8_t s_nzIdx[FILL_ROW][FILL_NZ_CAP], s_nzN[FILL_ROW], s_ocean[FILL_ROW], s_drawWater[FILL_ROW];
    __shared__ short s_drawMinY[FILL_ROW];

    const int t = threadIdx.x;
    const int bid = xcd_block(blockIdx.x, gridDim.x);
    const int outChunk = bid >> 4, row = bid & 15;                 // one workgroup = one 16-column row of a chunk (z = row)
    const int chunk = srcIdx ? srcIdx[outChunk] : outChunk;        // inputs are read at `chunk`, blocks are written densely at outChunk
    const int2 cp = chunkPos[chunk];
    const int idxBase = FILL_ROW * row;

    // the row's plane attributes, once: 24 weights + 20 layer starts + height per column.  A row of a plane is one 64-byte line, read whole
    for (int i = t; i < FILL_ROW * 45; i += FILL_THREADS) {
        const int k = i / FILL_ROW, c = i % FILL_ROW;               // consecutive lanes = consecutive columns of one plane
        const int idx2d = idxBase + c;
        if (k < MMGEN_NUM_BIOMES) s_bw[c][k] = bw[(size_t)MMGEN_BIOME_WEIGHTS_SIZE * chunk + 256 * k + idx2d];
        else if (k < MMGEN_NUM_BIOMES + MMGEN_NUM_MATERIALS) s_lh[c][k - MMGEN_NUM_BIOMES] = layers[(size_t)MMGEN_LAYERS_SIZE * chunk + 256 * (k - MMGEN_NUM_BIOMES) + idx2d];
        else s_lh[c][MMGEN_NUM_MATERIALS] = hf[chunk * 256 + idx2d];
    }
    // the row's cave layers: 16 columns x 32 layers x 12 bytes, contiguous
    for (int i = t; i < FILL_ROW * 96; i += FILL_THREADS)
        ((int*)s_cl)[i] = ((const int*)(caveLayers + (size_t)MMGEN_MAX_CAVE_LAYERS_PER_COLUMN * (chunk * 256 + idxBase)))[i];
    if (t < 3) s_count[t] = 0;
    __syncthreads();
    if (t < FILL_ROW) {
        int n = 0, minY = 384;
        bool ocean = false, water = false;
        for (int b = 0; b < MMGEN_NUM_BIOMES; ++b) {
            const float w = s_bw[t][b];
            if (b < MMGEN_NUM_OCEAN_BIOMES) ocean = ocean || (w > 0.f);
            if (b == 0 || w > 0.f) { if (n < FILL_NZ_CAP) { s_nzIdx[t][n] = (uint8_t)b; s_nzW[t][n] = w; } ++n; }
            // (biome 0 can be drawn at weight 0 and PLAINS is the walk's fall-through: neither has a rule)
            if (w > 0.f) { minY = imin(minY, biome_rule_min_y(b, s_lh[t][MMGEN_NUM_MATERIALS])); water = water || b == MMBIO_FROZEN_WASTELAND; }
        }
        bool sorted = true;
        for (int l = 0; l < MMGEN_NUM_MATERIALS; ++l)
            if (l != MMGEN_NUM_FORWARD_MATERIALS - 1) sorted = sorted && s_lh[t][l] <= s_lh[t][l + 1];
        s_nzN[t] = (uint8_t)n; s_ocean[t] = ocean ? 1 : 0; s_drawMinY[t] = (short)minY; s_drawWater[t] = (water ? 1 : 0) | (sorted ? 2 : 0);
    }
    __syncthreads();

    uint8_t* outBase = blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * outChunk + 384 * idxBase;     // the 16 columns are contiguous: 6 144 bytes

    // phase 1: base blocks.  The two cave-surface distances only matter as "== 0" and "0 .. threshold" with threshold = 1.5 + 4.5 * simplex3
    // (|simplex3| < 3.5 by the crudest bound: 42 * 4 corners * max((0.6 - r^2)^4 r) = 3.5, threshold < 17.25), so they travel as 5-bit
    // codes: negative -> 31, 30 and beyond -> 30.
    // Walk: a wave = 16 consecutive y of four neighbouring columns (16-byte pieces per column in a wave's store), the four column groups
    // of a 16-y block in four consecutive waves: the list that phase 2 walks comes out ordered by depth, and the exits of the cave-biome
    // evaluation - which go by depth zone - retire whole waves instead of idling lanes
    for (int u = t; u < FILL_VOX; u += FILL_THREADS) {
        const int c = 4 * ((u >> 6) & 3) + (u & 3), y = 16 * (u >> 8) + ((u >> 2) & 15);
        const int v = 384 * c + y;                                  // position in the row's 6 144 output bytes
        const int wx = cp.x + c;
        int wz = cp.y + row;
        // wz is the same for the whole workgroup (one row of a chunk): left visible, the compiler hoists (float)wz * scale for each of the six
        // block-rule noises out of the loop into six VGPRs it then has to spill (the kernel sits at its register budget)
        asm volatile("" : "+v"(wz));
        const ColumnBiomes cbi = {s_nzN[c], s_ocean[c] != 0, s_nzIdx[c], s_nzW[c], s_bw[c], s_drawMinY[c], (s_drawWater[c] & 1) != 0, (s_drawWater[c] & 2) != 0};
        const BaseBlock r = place_block_base(cbi, s_lh[c], s_cl[c], y, s_lh[c][MMGEN_NUM_MATERIALS], wx, wz);
        if (r.needCave) {
            const int slot = atomicAdd(&s_count[0], 1);
            const unsigned bdc = r.bottomDepth < 0 ? 31u : (unsigned)imin(r.bottomDepth, 30);
            const unsigned tdc = r.topDepth < 0 ? 31u : (unsigned)imin(r.topDepth, 30);
            s_list[slot] = (unsigned)v | ((unsigned)r.block << FILL_VBITS) | (bdc << (FILL_VBITS + 8)) | (tdc << (FILL_VBITS + 13));
        } else {
            outBase[v] = r.block;
        }
    }
    __syncthreads();

    // phase 3 body: the one simplex3 both noise rules start with; lush voxels within reach of a cave surface go to list 3
    auto noise_rule = [&](int i, int cb) {
        const unsigned e = s_list[i];
        const int v = e & FILL_VMASK;
        uint8_t block = (uint8_t)((e >> FILL_VBITS) & 255);
        const int bdc = (e >> (FILL_VBITS + 8)) & 31, tdc = (e >> (FILL_VBITS + 13)) & 31;
        const int c = v / 384, y = v - 384 * c;
        const int wx = cp.x + c, wz = cp.y + row;
        float ax, ay, az;
        cave_post_noise_pos(cb, wx, y, wz, ax, ay, az);
        const float n = simplex3_inl(ax, ay, az);
        if (cave_post_apply(block, cb, n, wx, y, wz, bdc == 31 ? -1 : bdc, tdc == 31 ? -1 : tdc)) {
            const int slot = atomicAdd(&s_count[2], 1);
            if (slot < FILL_L3_CAP) { s_list3[slot] = (unsigned short)i; return; }
            block = lush_clay_or_moss(wx, y, wz, CellDirect());       // more than FILL_L3_CAP in one row: in place
        }
        outBase[v] = block;
    };

    // phase 2: cave biome of the compacted stone voxels (9 simplex3 + 6 .. 12 simplex2 + 0 .. 2 simplex3 each).  NONE / WARPED / AMBER are
    // final here; CRYSTAL and LUSH voxels go to list 2
    const int count = s_count[0];
    for (int i = t; i < count; i += FILL_THREADS) {
        const unsigned e = s_list[i];
        const int v = e & FILL_VMASK;
        uint8_t block = (uint8_t)((e >> FILL_VBITS) & 255);
        const int bdc = (e >> (FILL_VBITS + 8)) & 31;
        const int c = v / 384, y = v - 384 * c;
        const int wx = cp.x + c, wz = cp.y + row;
        // WARPED / AMBER only act on the top DEEPSLATE / BLACKSTONE block of a cave floor (caveBottomDepth == 0)
        const bool wantDeep = bdc == 0 && (block == MMB_DEEPSLATE || block == MMB_BLACKSTONE);
        // LUSH_CAVES only converts within 1.5 + 4.5 simplex3 <= 1.5 + 4.5 * 1.37 = 7.67 blocks of a cave surface: further away only CRYSTAL_CAVES
        // can change the block (depth codes: 31 = no such surface; MM_SIMPLEX3_BOUND holds inside the pruning domain)
        const int tdc = (e >> (FILL_VBITS + 13)) & 31;
        static_assert(1.5f + 4.5f * MM_SIMPLEX3_BOUND < 8.f, "depth beyond which LUSH_CAVES cannot convert");
        const bool crystalOnly = NEAR && !wantDeep && bdc > 7 && tdc > 7;
        const int cb = cave_biome_t<NEAR>(wx, y, wz, s_lh[c][MMGEN_NUM_MATERIALS], 190249401, wantDeep, crystalOnly);
        if (cb == MMCB_CRYSTAL_CAVES || cb == MMCB_LUSH_CAVES) {
            const int k = atomicAdd(&s_count[1], 1);
            if (k < FILL_L2_CAP) s_list2[k] = (unsigned short)(i | (cb == MMCB_LUSH_CAVES ? (1 << FILL_VBITS) : 0));
            else noise_rule(i, cb);                                   // list 2 is full: in place
            continue;
        }
        if (wantDeep && cb != MMCB_NONE) cave_biome_block_post(block, cb, wx, y, wz, 0, -1);      // WARPED / AMBER re-skin, no noise
        outBase[v] = block;
    }
    __syncthreads();

    // phase 3, densely over list 2
    const int count2 = imin(s_count[1], FILL_L2_CAP);
    for (int k = t; k < count2; k += FILL_THREADS) {
        const int item = s_list2[k];
        noise_rule(item & FILL_VMASK, (item & (1 << FILL_VBITS)) ? MMCB_LUSH_CAVES : MMCB_CRYSTAL_CAVES);
    }
    __syncthreads();

    // phase 4: clay or moss (fbm3from3<3> + a 27-cell Worley search with 81 sin hashes = ~5 600 instructions) for the few lush voxels that
    // got here: typically 0 - 100 per row, which would leave most lanes of the workgroup idle through the longest code path of the
    // kernel.  They are appended to a device-wide queue instead and k_fill_lush evaluates them 64 to a wave; only when the queue is
    // full (or absent) are they evaluated here.
    const int count3 = imin(s_count[2], FILL_L3_CAP);
    if (t == 0 && count3) s_qbase = lushQueue ? atomicAdd(lushQueue, (unsigned)count3) : 0xffffffffu;
    __syncthreads();
    const unsigned qbase = count3 ? s_qbase : 0u;
    const bool queued = lushQueue && qbase <= lushCap && (unsigned)count3 <= lushCap - qbase;
    for (int k = t; k < count3; k += FILL_THREADS) {
        const unsigned e = s_list[s_list3[k]];
        const int v = e & FILL_VMASK;
        const int c = v / 384, y = v - 384 * c;
        const int idx2d = idxBase + c;
        if (queued) { lushQueue[1 + qbase + k] = ((unsigned)outChunk << 17) | ((unsigned)idx2d << 9) | (unsigned)y; continue; }
        // a reservation that straddles the capacity marks its in-range slots as holes (they hold the previous launch's entries)
        if (lushQueue && qbase < lushCap && (unsigned)k < lushCap - qbase) lushQueue[1 + qbase + k] = 0xffffffffu;
        outBase[v] = lush_clay_or_moss(cp.x + c, y, cp.y + row, CellDirect());
    }
}

// ---------------------------------------------------------------------------------------------------------
// K6 for the rows INSIDE the pruning domain (all of them, for any world within 32 768 blocks of the origin): three kernels.
//   k_fill_base  one workgroup = one row: every voxel's base block, written; the stone voxels a cave biome could still alter are appended
//                to the row's list in global memory (4 bytes each).  No noise tables unless a biome with a noise rule has weight in the
//                row, 9 KB of LDS: seven workgroups per CU hide the staging latency that three 50 KB workgroups of the fused kernel could not.
//   k_fill_scan  exclusive prefix of the rows' 64-voxel batch counts + the row each range of `range` batches starts in.
//   k_fill_cave  persistent waves, no workgroup barrier after the tables are staged: a wave draws ranges of batches from work counters
//                (k_apply_features' scheme) and evaluates the cave biome of 64 listed voxels at a time, every lane busy whatever row the
//                voxels come from.  CRYSTAL / LUSH voxels (their rules start with one more simplex3) are set aside in a per-wave LDS
//                buffer and evaluated 64 at a time as soon as 64 have gathered; lush voxels close to a cave surface go from a second
//                per-wave buffer to the device-wide queue of k_fill_lush, 64 per reservation.
// k_fill_far (fill_body<false>) keeps the rows beyond the domain, the fused way.
// ---------------------------------------------------------------------------------------------------------
#ifndef FILL_RANGE
#define FILL_RANGE 4             // batches per draw at most: ~60 us of work, the longest a wave can still be busy after the others ran dry
                                 // (small launches draw smaller ranges: launch_fill)
#endif
#define FILL_COUNTERS 16         // work counters, 64 B apart (one serialises at ~11 ns per draw in L2)
#define FILL_CLEAR_BYTES (64 * FILL_COUNTERS + 16)      // ... and the 16 bytes behind them: the lush queue's count word (+ three entries that every launch rewrites)
#define FILLB_THREADS 256
#ifndef FILLC_THREADS
#define FILLC_THREADS 256       // waves per persistent workgroup x 64 (the waves of a workgroup share the noise tables and nothing else)
#endif
#define FILLC_DEF_CAP 128        // per wave: CRYSTAL / LUSH voxels waiting for their noise rule
#define FILLC_LUSH_CAP 128       // per wave: lush voxels waiting for a queue reservation

#ifndef MM_FILLB_WAVES
#define MM_FILLB_WAVES 8          // 64 VGPRs without scratch; 18.7 KB of LDS per workgroup allows eight (7 -> 8: 1.73 -> 1.66 ms, round 4)
#endif
__attribute__((amdgpu_waves_per_eu(MM_FILLB_WAVES, MM_FILLB_WAVES)))
__global__ void __launch_bounds__(FILLB_THREADS)
k_fill_base(const float* __restrict__ hf, const float* __restrict__ bw, const float* __restrict__ layers, const mmgen_cave_layer* __restrict__ caveLayers,
            const int2* __restrict__ chunkPos, uint8_t* __restrict__ blocks, const int* __restrict__ srcIdx, int row0,
            unsigned* __restrict__ rowLists /*[rows][FILL_VOX]*/, int* __restrict__ rowCounts)
{
    __shared__ float s_bw[FILL_ROW][MMGEN_NUM_BIOMES];
    __shared__ float s_lh[FILL_ROW][MMGEN_NUM_MATERIALS + 1];
    __shared__ unsigned long long s_cave[FILL_ROW][CAVE_MASK_WORDS];      // the columns' cave voxels, one bit each (cave_mask_add)
    __shared__ int s_count, s_needTables;
    __shared__ float s_nzW[FILL_ROW][FILL_NZ_CAP];
    __shared__ uint8_t s_nzIdx[FILL_ROW][FILL_NZ_CAP], s_nzN[FILL_ROW], s_ocean[FILL_ROW], s_drawWater[FILL_ROW];
    __shared__ short s_drawMinY[FILL_ROW];

    const int t = threadIdx.x;
    const int lrow = xcd_block(blockIdx.x, gridDim.x), bid = row0 + lrow;      // row of this launch / of the batch the pointers belong to
    const int outChunk = bid >> 4, row = bid & 15;                 // one workgroup = one 16-column row of a chunk (z = row)
    const int chunk = srcIdx ? srcIdx[outChunk] : outChunk;        // inputs are read at `chunk`, blocks are written densely at outChunk
    const int2 cp = chunkPos[chunk];
    if (!prune_domain(cp.x, cp.y + row)) {                         // k_fill_far's row (the row's x range is [cp.x, cp.x + 15], cp.x a multiple of 16)
        if (t == 0) rowCounts[lrow] = 0;
        return;
    }
    const int idxBase = FILL_ROW * row;
    static_assert(FILLB_THREADS == 16 * FILL_ROW && MMGEN_MAX_CAVE_LAYERS_PER_COLUMN == 32, "16 lanes per column, two cave layers per lane");

    // the row's cave layers straight into registers (lane g of a column's 16: slots g and g + 16; 16 columns x 32 slots x 12 bytes are
    // contiguous) while the plane attributes are on their way to LDS: 24 weights + 20 layer starts + height per column, a row of a plane
    // is one 64-byte line, read whole
    const mmgen_cave_layer* ccl = caveLayers + (size_t)MMGEN_MAX_CAVE_LAYERS_PER_COLUMN * (chunk * 256 + idxBase + (t >> 4));
    const int cs0 = ccl[t & 15].start, ce0 = ccl[t & 15].end;
    // a column's slots are used in order (the first unused one starts at 384): the upper 16 are only read where slot 15 is in use - a
    // handful of columns of a generated world, and half of the 96 KB per chunk that were this kernel's largest input
    int cs1 = 384, ce1 = 384;
    if (__shfl(cs0, (t & 48) | 15) != 384) { cs1 = ccl[(t & 15) + 16].start; ce1 = ccl[(t & 15) + 16].end; }
    for (int i = t; i < FILL_ROW * 45; i += FILLB_THREADS) {
        const int k = i / FILL_ROW, c = i % FILL_ROW;               // consecutive lanes = consecutive columns of one plane
        const int idx2d = idxBase + c;
        if (k < MMGEN_NUM_BIOMES) s_bw[c][k] = bw[(size_t)MMGEN_BIOME_WEIGHTS_SIZE * chunk + 256 * k + idx2d];
        else if (k < MMGEN_NUM_BIOMES + MMGEN_NUM_MATERIALS) s_lh[c][k - MMGEN_NUM_BIOMES] = layers[(size_t)MMGEN_LAYERS_SIZE * chunk + 256 * (k - MMGEN_NUM_BIOMES) + idx2d];
        else s_lh[c][MMGEN_NUM_MATERIALS] = hf[chunk * 256 + idx2d];
    }
    if (t < FILL_ROW * CAVE_MASK_WORDS) (&s_cave[0][0])[t] = 0ull;
    if (t == 0) { s_count = 0; s_needTables = 0; }
    __syncthreads();
    cave_mask_add(s_cave[t >> 4], cs0, ce0);
    cave_mask_add(s_cave[t >> 4], cs1, ce1);
    {
        // per column: the biomes of positive weight (plus biome 0) in ascending order, and what the voxel loop needs to know about them -
        // 16 lanes per column (lane g looks at biomes g and g + 16, at layer pairs g and g + 16), ordered by ballots
        static_assert(MMGEN_NUM_BIOMES <= 32 && MMGEN_NUM_MATERIALS <= 32, "16 lanes per column");
        const int c = t >> 4, g = t & 15, shift = 16 * ((t & 63) >> 4);
        auto group = [&](bool p) { return (unsigned)(__ballot(p) >> shift) & 0xffffu; };      // the predicate over this column's 16 lanes
        const float height = s_lh[c][MMGEN_NUM_MATERIALS];
        const bool has1 = g + 16 < MMGEN_NUM_BIOMES;
        const float w0 = s_bw[c][g], w1 = has1 ? s_bw[c][g + 16] : 0.f;
        const bool k0 = g == 0 || w0 > 0.f, k1 = has1 && w1 > 0.f;
        const unsigned m0 = group(k0), m1 = group(k1), lower = (1u << g) - 1u;
        const int n0 = __popc(m0), pos0 = __popc(m0 & lower), pos1 = n0 + __popc(m1 & lower);
        if (k0 && pos0 < FILL_NZ_CAP) { s_nzIdx[c][pos0] = (uint8_t)g; s_nzW[c][pos0] = w0; }
        if (k1 && pos1 < FILL_NZ_CAP) { s_nzIdx[c][pos1] = (uint8_t)(g + 16); s_nzW[c][pos1] = w1; }
        const bool ocean = group((g < MMGEN_NUM_OCEAN_BIOMES && w0 > 0.f) || (g + 16 < MMGEN_NUM_OCEAN_BIOMES && w1 > 0.f)) != 0u;
        // (biome 0 can be drawn at weight 0 and PLAINS is the walk's fall-through: neither has a rule)
        int minY = imin(w0 > 0.f ? biome_rule_min_y(g, height) : 384, w1 > 0.f ? biome_rule_min_y(g + 16, height) : 384);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) minY = imin(minY, __shfl_xor(minY, o));
        const bool water = group((g == MMBIO_FROZEN_WASTELAND && w0 > 0.f) || (g + 16 == MMBIO_FROZEN_WASTELAND && w1 > 0.f)) != 0u;
        const int l1 = g + 16;
        const bool ok0 = g == MMGEN_NUM_FORWARD_MATERIALS - 1 || s_lh[c][g] <= s_lh[c][g + 1];
        const bool ok1 = l1 >= MMGEN_NUM_MATERIALS || l1 == MMGEN_NUM_FORWARD_MATERIALS - 1 || s_lh[c][l1] <= s_lh[c][l1 + 1];
        const bool sorted = group(ok0 && ok1) == 0xffffu;
        if (g == 0) {
            s_nzN[c] = (uint8_t)(n0 + __popc(m1)); s_ocean[c] = ocean ? 1 : 0; s_drawMinY[c] = (short)minY; s_drawWater[c] = (water ? 1 : 0) | (sorted ? 2 : 0);
            if (minY < 384) s_needTables = 1;                       // every rule with a height threshold is a noise rule (biome_rule_min_y)
        }
    }
    __syncthreads();
    // simplex tables only for the rows in which a biome with a noise rule has weight (a biome is only drawn at positive weight)
    if (s_needTables) noise_tables_init();
    uint8_t* outBase = blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * outChunk + 384 * idxBase;     // the 16 columns are contiguous: 6 144 bytes
    unsigned* list = rowLists + (size_t)FILL_VOX * lrow;

    // A listed voxel: its position in the row (FILL_VBITS), its block, and the two cave-surface distances of the layer walk as 5-bit codes
    // (depth_codes), read off the column's mask.
    // Walk: a wave = 16 consecutive y of four neighbouring columns (16-byte pieces per column in a wave's store), the four column groups
    // of a 16-y block in four consecutive waves: the list comes out ordered by depth, and the exits of the cave-biome evaluation - which
    // go by depth zone - retire whole waves of k_fill_cave instead of idling lanes
    static_assert(FILLB_THREADS == 256, "the walk below: a thread keeps its column, y advances by 16 per iteration");
    const int c = 4 * (t >> 6) + (t & 3);
    // everything above the column's surface and the sea is air (place_block_base's first test): nothing else is evaluated there - two
    // thirds of a generated world's voxels
    const int yAir = imax((int)__builtin_floorf(s_lh[c][MMGEN_NUM_MATERIALS]), MMGEN_SEA_LEVEL);      // y > yAir  <=>  fy > height && y > SEA_LEVEL
    for (int u = t; u < FILL_VOX; u += FILLB_THREADS) {
        const int y = 16 * (u >> 8) + ((u >> 2) & 15);
        const int v = 384 * c + y;                                  // position in the row's 6 144 output bytes
        if (y > yAir) { outBase[v] = MMB_AIR; continue; }
        const int wx = cp.x + c;
        int wz = cp.y + row;
        // wz is the same for the whole workgroup (one row of a chunk): left visible, the compiler hoists (float)wz * scale for each of the six
        // block-rule noises out of the loop into six VGPRs
        asm volatile("" : "+v"(wz));
        const ColumnBiomes cbi = {s_nzN[c], s_ocean[c] != 0, s_nzIdx[c], s_nzW[c], s_bw[c], s_drawMinY[c], (s_drawWater[c] & 1) != 0, (s_drawWater[c] & 2) != 0};
        const BaseBlock r = place_block_base<true>(cbi, s_lh[c], s_cave[c], y, s_lh[c][MMGEN_NUM_MATERIALS], wx, wz);
        outBase[v] = r.block;                                       // k_fill_cave only writes the voxels it changes
        if (r.needCave) {
            unsigned bdc, tdc;
            depth_codes(s_cave[c], y, bdc, tdc);
            list[atomicAdd(&s_count, 1)] = (unsigned)v | ((unsigned)r.block << FILL_VBITS) | (bdc << (FILL_VBITS + 8)) | (tdc << (FILL_VBITS + 13));
        }
    }
    __syncthreads();
    if (t == 0) rowCounts[lrow] = s_count;
}

// batchStart[r] = number of 64-voxel batches in the lists of the rows before r (batchStart[nRows] = all of them); rangeRow[g] = the row
// batch range * g lies in.  One workgroup per 1 024 rows; each sums the counts before its rows itself (a few hundred KB out of L2).
// (In the stage DAG its 16-wave workgroups find no CU to start on until k_feature_placements has left the chip: 0.17 ms instead of 0.03.
// That wait is worth having: with four-wave workgroups it is 0.11 ms, k_fill_cave's persistent workgroups then start beside the placement
// kernel's, the gather takes the slots those free, and the cave fill runs 6.1 ms instead of 5.5 - step 21.5 -> 22.1 ms, profiles/LOG.md r05.)
__global__ void __launch_bounds__(1024)
k_fill_scan(const int* __restrict__ rowCounts, int nRows, int range, int* __restrict__ batchStart, int* __restrict__ rangeRow)
{
    __shared__ int s_w[16], s_offset;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    auto wave_sum = [&](int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; };
    int part = 0;
    for (int i = t; i < 1024 * (int)blockIdx.x; i += 1024) part += (rowCounts[i] + 63) >> 6;
    part = wave_sum(part);
    if (lane == 0) s_w[wave] = part;
    __syncthreads();
    if (t == 0) { int o = 0; for (int w = 0; w < 16; ++w) o += s_w[w]; s_offset = o; }
    __syncthreads();
    const int offset = s_offset;
    const int row = 1024 * blockIdx.x + t;
    const int nb = row < nRows ? (rowCounts[row] + 63) >> 6 : 0;
    int incl = nb;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    __syncthreads();                                               // s_w is re-used
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += s_w[w];
    const int a = offset + before + incl - nb, b = a + nb;
    if (row < nRows) {
        batchStart[row] = a;
        if (row == nRows - 1) batchStart[nRows] = b;
        for (int g = (a + range - 1) / range; g * range < b; ++g) rangeRow[g] = row;
    }
}

__attribute__((amdgpu_waves_per_eu(MM_FILL_WAVES, MM_FILL_WAVES)))
__global__ void __launch_bounds__(FILLC_THREADS)
k_fill_cave(const float* __restrict__ hf, const int2* __restrict__ chunkPos, uint8_t* __restrict__ blocks, const int* __restrict__ srcIdx, int row0,
            const unsigned* __restrict__ rowLists, const int* __restrict__ rowCounts, const int* __restrict__ batchStart, const int* __restrict__ rangeRow,
            int nRows, int range, unsigned* __restrict__ lushQueue /*[0] = count, entries from [1]; nullable*/, unsigned lushCap, unsigned* __restrict__ work)
{
    __shared__ uint2 s_def[FILLC_THREADS / 64][FILLC_DEF_CAP];     // .x = list entry | isLush << 31, .y = row of this launch
    __shared__ unsigned s_lushBuf[FILLC_THREADS / 64][FILLC_LUSH_CAP];      // k_fill_lush's entries: outChunk << 17 | column << 9 | y
    __shared__ uint2 s_s2[FILLC_THREADS / 64][FILLC_DEF_CAP];      // voxels whose warped height decides nothing: list entry, row of this launch ...
    __shared__ float s_s2py[FILLC_THREADS / 64][FILLC_DEF_CAP];    // ... and that height
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (threadIdx.x == 0) atomicAdd(&work[1], 1u);                 // started workgroups (a word of the first counter's line): launch_fill's startedCounter
    noise_tables_init();                                           // once per (persistent) workgroup; no workgroup barrier after this one
    uint2* def = s_def[wave];
    unsigned* lushBuf = s_lushBuf[wave];
    uint2* s2 = s_s2[wave];
    float* s2py = s_s2py[wave];
    int nDef = 0, nLush = 0, nS2 = 0;                              // wave-uniform
    const unsigned long long below = (1ull << lane) - 1ull;

    // the position of a listed voxel; the row's chunk position through L2 (a wave's voxels come from one or two rows)
    struct Voxel { int outChunk, c, y, wx, wz; };
    auto voxel_of = [&](int lrow, int v) {
        Voxel x;
        const int bid = row0 + lrow;
        x.outChunk = bid >> 4;
        const int2 cp = chunkPos[srcIdx ? srcIdx[x.outChunk] : x.outChunk];
        x.c = v / 384; x.y = v - 384 * x.c;
        x.wx = cp.x + x.c; x.wz = cp.y + (bid & 15);
        return x;
    };
    auto block_ptr = [&](int lrow, int v) { const int bid = row0 + lrow; return blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * (bid >> 4) + 384 * FILL_ROW * (bid & 15) + v; };

    const int total = batchStart[nRows], nRanges = (total + range - 1) / range;
    int part = (int)((FILLC_THREADS / 64) * blockIdx.x + wave) % FILL_COUNTERS, dry = 0;
    (void)dry;
    unsigned drawn = 0u;
    if (lane == 0) drawn = atomicAdd(&work[16 * part], 1u);
    bool more = true;                                              // ranges left to draw
    int b = 0, b1 = 0, lrow = 0, rowA = 0, rowB = 0, cnt = 0;      // the range being worked on: next batch, end, and the row batch b lies in
    // Per batch only the list entry comes from memory, and it is requested one batch ahead (a row has ~28 batches in a row); the row's
    // 16 heights sit in the lanes of one register
    unsigned ePre = 0u;
    int preB = -1;

    // One loop, three kinds of step (each heavy body exists once in the code): a full reservation of lush voxels, a full batch of deferred
    // CRYSTAL / LUSH voxels, the next batch of listed voxels; when the ranges have run dry the two buffers are drained.
    for (;;) {
        if (nLush >= 64 || (!more && nDef == 0 && nLush > 0)) {
            // lush voxels from the top of the wave's buffer to the device-wide queue; a reservation that does not fit is evaluated here
            const int n = imin(nLush, 64);
            wave_lds_sync();
            unsigned qbase = 0xffffffffu;
            if (lushQueue) { if (lane == 0) qbase = atomicAdd(lushQueue, (unsigned)n); qbase = (unsigned)__builtin_amdgcn_readfirstlane((int)qbase); }
            const bool queued = lushQueue && qbase <= lushCap && (unsigned)n <= lushCap - qbase;
            if (lane < n) {
                const unsigned e = lushBuf[nLush - n + lane];
                if (queued) lushQueue[1 + qbase + lane] = e;
                else {
                    // a reservation that straddles the capacity marks its in-range slots as holes (they hold the previous launch's entries)
                    if (lushQueue && qbase < lushCap && (unsigned)lane < lushCap - qbase) lushQueue[1 + qbase + lane] = 0xffffffffu;
                    const int outChunk = e >> 17, idx2d = (e >> 9) & 255, y = e & 511;
                    const int2 cp = chunkPos[srcIdx ? srcIdx[outChunk] : outChunk];
                    blocks[(size_t)MMGEN_BLOCKS_PER_CHUNK * outChunk + 384 * idx2d + y] = lush_clay_or_moss(cp.x + (idx2d & 15), y, cp.y + (idx2d >> 4), CellDirect());
                }
            }
            nLush -= n;
            continue;
        }
        if (nDef >= 64 || (!more && nDef > 0)) {
            // CRYSTAL / LUSH voxels from the top of the wave's buffer: the one simplex3 both noise rules start with
            const int n = imin(nDef, 64);
            wave_lds_sync();
            bool lush = false;
            unsigned word = 0u;
            if (lane < n) {
                const uint2 d = def[nDef - n + lane];
                const int dr = (int)d.y, v = d.x & FILL_VMASK;
                const int cb = (d.x >> 31) ? MMCB_LUSH_CAVES : MMCB_CRYSTAL_CAVES;
                const uint8_t base = (uint8_t)((d.x >> FILL_VBITS) & 255);
                uint8_t block = base;
                const int bdc = (d.x >> (FILL_VBITS + 8)) & 31, tdc = (d.x >> (FILL_VBITS + 13)) & 31;
                const Voxel x = voxel_of(dr, v);
                float ax, ay, az;
                cave_post_noise_pos(cb, x.wx, x.y, x.wz, ax, ay, az);
                const float nz = simplex3_inl(ax, ay, az);
                lush = cave_post_apply(block, cb, nz, x.wx, x.y, x.wz, bdc == 31 ? -1 : bdc, tdc == 31 ? -1 : tdc);
                if (lush) word = ((unsigned)x.outChunk << 17) | ((unsigned)(FILL_ROW * ((row0 + dr) & 15) + x.c) << 9) | (unsigned)x.y;
                else if (block != base) *block_ptr(dr, v) = block;
            }
            nDef -= n;
            const unsigned long long lm = __ballot(lush);
            if (lush) lushBuf[nLush + __popcll(lm & below)] = word;
            nLush += __popcll(lm);
            continue;
        }
        if (nS2 >= 64 || (!more && nS2 > 0)) {
            // stage 2 of the cave biome (the x and z components of the warp, rocky, one depth band, the draw) for voxels whose warped
            // height did not settle it
            const int n = imin(nS2, 64);
            wave_lds_sync();
            bool defer = false;
            uint2 d = make_uint2(0u, 0u);
            if (lane < n) {
                d = s2[nS2 - n + lane];
                const float py = s2py[nS2 - n + lane];
                const unsigned e = d.x;
                const int dr = (int)d.y, v = e & FILL_VMASK;
                const uint8_t base = (uint8_t)((e >> FILL_VBITS) & 255);
                const int bdc = (e >> (FILL_VBITS + 8)) & 31, tdc = (e >> (FILL_VBITS + 13)) & 31;
                const Voxel x = voxel_of(dr, v);
                const bool wantDeep = bdc == 0 && (base == MMB_DEEPSLATE || base == MMB_BLACKSTONE);
                // LUSH_CAVES only converts within 1.5 + 4.5 simplex3 <= 1.5 + 4.5 * 1.37 = 7.67 blocks of a cave surface: further away only
                // CRYSTAL_CAVES can change the block (depth codes: 31 = no such surface; MM_SIMPLEX3_BOUND holds inside the pruning domain)
                static_assert(1.5f + 4.5f * MM_SIMPLEX3_BOUND < 8.f, "depth beyond which LUSH_CAVES cannot convert");
                const bool crystalOnly = !wantDeep && bdc > 7 && tdc > 7;
                const float maxHeight = hf[(srcIdx ? srcIdx[x.outChunk] : x.outChunk) * 256 + FILL_ROW * ((row0 + dr) & 15) + x.c];
                const int cb = cave_biome_rest<true>(x.wx, x.y, x.wz, maxHeight, 190249401, wantDeep, crystalOnly, py);
                if (cb == MMCB_CRYSTAL_CAVES || cb == MMCB_LUSH_CAVES) {
                    defer = true;
                    d.x = e | (cb == MMCB_LUSH_CAVES ? 0x80000000u : 0u);
                } else if (wantDeep && cb != MMCB_NONE) {
                    uint8_t block = base;
                    cave_biome_block_post(block, cb, x.wx, x.y, x.wz, 0, -1);      // WARPED / AMBER re-skin, no noise
                    if (block != base) *block_ptr(dr, v) = block;
                }
            }
            nS2 -= n;
            const unsigned long long dm = __ballot(defer);
            if (defer) def[nDef + __popcll(dm & below)] = d;
            nDef += __popcll(dm);
            continue;
        }
        if (!more) break;
        if (b >= b1) {                                             // next range
            const int g = __builtin_amdgcn_readfirstlane((int)drawn) * FILL_COUNTERS + part;
            if (g >= nRanges) {
#if MM_COUNTER_PROBE_LOADS
                const int nx = next_live_counter(work, FILL_COUNTERS, part, nRanges);
                if (nx < 0) { more = false; continue; }
                part = nx;
#else
                if (++dry == FILL_COUNTERS) { more = false; continue; }
                part = (part + 1) % FILL_COUNTERS;
#endif
            }
            if (lane == 0) drawn = atomicAdd(&work[16 * part], 1u);        // the next draw is in flight while this range is worked on
            if (g >= nRanges) continue;
            b = range * g; b1 = imin(b + range, total);
            lrow = rangeRow[g];
            rowA = batchStart[lrow]; rowB = batchStart[lrow + 1]; cnt = rowCounts[lrow];
        }
        while (b >= rowB) { ++lrow; rowA = rowB; rowB = batchStart[lrow + 1]; cnt = rowCounts[lrow]; }      // rows without stone voxels have no batch
        const int k = 64 * (b - rowA) + lane;
        const unsigned* list = rowLists + (size_t)FILL_VOX * lrow;
        unsigned e = ePre;
        if (preB != b) e = k < cnt ? list[k] : 0u;
        ++b;
        if (b < b1 && b < rowB) { ePre = k + 64 < cnt ? list[k + 64] : 0u; preB = b; }      // same row, same range: the next step's entry
        // stage 1: the warped height (3 of the 9 simplex3 of the warp, the same work in every lane); above / below the depth bands it
        // settles the biome (NONE), the other voxels go on to stage 2, 64 at a time
        bool on = false;
        float py = 0.f;
        if (k < cnt) {
            const int v = e & FILL_VMASK;
            const uint8_t base = (uint8_t)((e >> FILL_VBITS) & 255);
            const int bdc = (e >> (FILL_VBITS + 8)) & 31;
            const Voxel x = voxel_of(lrow, v);
            // WARPED / AMBER only act on the top DEEPSLATE / BLACKSTONE block of a cave floor (caveBottomDepth == 0)
            const bool wantDeep = bdc == 0 && (base == MMB_DEEPSLATE || base == MMB_BLACKSTONE);
            const float maxHeight = hf[(srcIdx ? srcIdx[x.outChunk] : x.outChunk) * 256 + FILL_ROW * ((row0 + lrow) & 15) + x.c];
            on = !cave_biome_py<true>(x.wx, x.y, x.wz, maxHeight, wantDeep, py);
        }
        const unsigned long long om = __ballot(on);
        if (on) { const int at = nS2 + __popcll(om & below); s2[at] = make_uint2(e, (unsigned)lrow); s2py[at] = py; }
        nS2 += __popcll(om);
    }
}

__global__ void __launch_bounds__(FILL_THREADS)
k_fill_far(const float* __restrict__ hf, const float* __restrict__ bw, const float* __restrict__ layers, const mmgen_cave_layer* __restrict__ caveLayers,
           const int2* __restrict__ chunkPos, uint8_t* __restrict__ blocks, const int* __restrict__ srcIdx, unsigned* __restrict__ lushQueue, unsigned lushCap)
{
    fill_body<false>(hf, bw, layers, caveLayers, chunkPos, blocks, srcIdx, lushQueue, lushCap);
}

// The queued lush voxels of a whole k_fill launch, 64 to a wave (entry = outChunk << 17 | column << 9 | y).
__global__ void __launch_bounds__(256)
k_fill_lush(const unsigned* __restrict__ lushQueue, unsigned lushCap, const int2* __restrict__ chunkPos, const int* __restrict__ srcIdx,
            uint8_t* __restrict__ blocks)
{
    noise_tables_init<false>();
    const unsigned reserved = lushQueue[0];
    // reservations beyond the capacity were evaluated by k_fill itself; every reservation that fits lies below lushCap
    const unsigned n = reserved < lushCap ? reserved : lushCap;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned e = lushQueue[1 + i];
        if (e == 0xffffffffu) continue;                       // hole left by a reservation that straddled the capacity
        const int outChunk = e >> 17, idx2d = (e >> 9) & 255, y = e & 511;
        const int2 cp = chunkPos[srcIdx ? srcIdx[outChunk] : outChunk];
        blocks[(size_t)MMGEN_BLOCKS_PER_CHUNK * outChunk + 384 * idx2d + y] = lush_clay_or_moss(cp.x + (idx2d & 15), y, cp.y + (idx2d >> 4), CellDirect());
    }
}

// =========================================================================================================
// Debug probes: evaluate one device function per item so that tests can pin the device math against the golden
// vectors (real glm simplex, frozen KATs) through the C ABI.  Inputs/outputs are packed fp32 (ints bit-cast).
// =========================================================================================================
__global__ void __launch_bounds__(256) k_probe(int fn, const float* __restrict__ in, int n, float* __restrict__ out)
{
    noise_tables_init();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    switch (fn) {
    case MMGEN_PROBE_SIN: out[i] = sinf_(in[i]); break;
    case MMGEN_PROBE_COS: out[i] = cosf_(in[i]); break;
    case MMGEN_PROBE_POW: out[i] = powf_(in[2 * i], in[2 * i + 1]); break;
    case MMGEN_PROBE_ATAN2: out[i] = atan2f_(in[2 * i], in[2 * i + 1]); break;
    case MMGEN_PROBE_ACOS: out[i] = acosf_(in[i]); break;
    case MMGEN_PROBE_SIMPLEX2: out[i] = simplex2(in[2 * i], in[2 * i + 1]); break;
    case MMGEN_PROBE_SIMPLEX3: out[i] = simplex3(in[3 * i], in[3 * i + 1], in[3 * i + 2]); break;
    case MMGEN_PROBE_FBM2_5: out[i] = fbm2<5>(in[2 * i], in[2 * i + 1]); break;
    case MMGEN_PROBE_FBM3_4: out[i] = fbm3<4>(in[3 * i], in[3 * i + 1], in[3 * i + 2]); break;
    case MMGEN_PROBE_RAND3FROM3: { const f3 r = rand3from3(in[3 * i], in[3 * i + 1], in[3 * i + 2]); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
    case MMGEN_PROBE_WORLEY2: {
        const Worley2 w = worley2(in[2 * i], in[2 * i + 1]); const f3 c = rand3from2(w.closest.x, w.closest.y);
        float* o = out + 5 * i; o[0] = w.d1; o[1] = c.x; o[2] = c.y; o[3] = c.z; o[4] = (w.d2 - w.d1) * 0.5f; break; }
    case MMGEN_PROBE_WORLEY3: {
        const Worley3 w = worley3(in[3 * i], in[3 * i + 1], in[3 * i + 2], CellDirect()); const f3 c = rand3from3(w.closest.x, w.closest.y, w.closest.z);
        float* o = out + 5 * i; o[0] = w.d1; o[1] = c.x; o[2] = c.y; o[3] = c.z; o[4] = (w.d2 - w.d1) * 0.5f; break; }
    case MMGEN_PROBE_SPECIAL_CAVE_NOISE: out[i] = special_cave_noise(in[3 * i], in[3 * i + 1], in[3 * i + 2], CellDirect()); break;
    case MMGEN_PROBE_BIOME_HEIGHT: out[i] = biome_height(__float_as_int(in[3 * i]), in[3 * i + 1], in[3 * i + 2]); break;
    case MMGEN_PROBE_CAVE_BIOME:
        out[i] = (float)cave_biome(__float_as_int(in[5 * i]), __float_as_int(in[5 * i + 1]), __float_as_int(in[5 * i + 2]), in[5 * i + 3], __float_as_int(in[5 * i + 4]));
        break;
    case MMGEN_PROBE_HASH: out[i] = __int_as_float((int)hash32((uint32_t)__float_as_int(in[i]))); break;
    case MMGEN_PROBE_RNG4_U01: {   // in: x y z w (ints), out: 4 draws; w == INT_MIN selects the 3-argument seeding
        const int x = __float_as_int(in[4 * i]), y = __float_as_int(in[4 * i + 1]), z = __float_as_int(in[4 * i + 2]), w = __float_as_int(in[4 * i + 3]);
        MinStd r = (w == (int)0x80000000) ? rng3(x, y, z) : rng4(x, y, z, w);
        for (int k = 0; k < 4; ++k) out[4 * i + k] = r.u01();
        break; }
    case MMGEN_PROBE_SIMPLEX3_SPLIT: {
        const Sx3Cell c = simplex3_part1(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
        float q[12];
        simplex3_gradients(c, q);
        out[i] = simplex3_part3(c, q);
        break; }
    default: break;
    }
}

}  // namespace mm

// =========================================================================================================
// launchers (host)
// =========================================================================================================
namespace mmk {

using mmk::KID_FILL_FAR; using mmk::KID_HEIGHTFIELD; using mmk::KID_LAYERS; using mmk::KID_FIX_BACKWARD; using mmk::KID_CAVE_COLUMNS; using mmk::KID_CAVE_VOXELS;
using mmk::KID_CAVE_BIOMES; using mmk::KID_FILL; using mmk::KID_PROBE; using mmk::KID_FILL_LUSH;

// every kernel of this translation unit can reach simplex noise: make sure the per-device table image exists, then launch (timed when
// profiling is on, mmgen_prof.h)
#define LAUNCH(KID, KERNEL, GRID, BLOCK, STREAM, ...)                                        \
    do {                                                                                     \
        { const int ne_ = mm::noise_tables_ensure(STREAM); if (ne_) return ne_; }            \
        MMK_LAUNCH(KID, KERNEL, GRID, BLOCK, STREAM, __VA_ARGS__);                           \
    } while (0)

int prepare_kernels() { return mm::noise_tables_ensure(nullptr); }

int launch_heightfield(const int32_t* pos, int n, float* hf, float* bw, float* gathered, hipStream_t s)
{
    if (n <= 0) return 0;
    if (gathered) LAUNCH(KID_HEIGHTFIELD, mm::k_heightfield<true>, dim3(n), dim3(384), s, (const int2*)pos, hf, bw, gathered);
    else LAUNCH(KID_HEIGHTFIELD, mm::k_heightfield<false>, dim3(n), dim3(256), s, (const int2*)pos, hf, bw, (float*)nullptr);
    return 0;
}

int launch_layers(const float* gathered, const float* bw, const int32_t* pos, int n, float* layers, hipStream_t s, float* stratifiedCopy, int nCopy)
{
    if (n <= 0) return 0;
    LAUNCH(KID_LAYERS, mm::k_layers, dim3(n), dim3(256), s, gathered, bw, (const int2*)pos, layers, stratifiedCopy, nCopy);
    return 0;
}

int launch_fix_backward(float* layers, int n, hipStream_t s)
{
    if (n <= 0) return 0;
    LAUNCH(KID_FIX_BACKWARD, mm::k_fix_backward, dim3(n), dim3(256), s, layers);
    return 0;
}

int launch_caves(const float* hf, const float* bw, const int32_t* pos, int n, mmgen_cave_layer* caveLayers, float* colInfoScratch, int colInfoChunks,
                 const int* chunkList, const uint8_t* colNeed, hipStream_t s, hipEvent_t afterVoxels, int biomeWorkgroupsPerCu, hipEvent_t beforeVoxels,
                 const unsigned* waitCounter, unsigned waitTarget)
{
    if (n <= 0) return 0;
    // (scratch layout: [n chunks][256] float2 per-column info, then k_cave_biomes' work counters - cave_scratch_bytes)
    unsigned* cbWork = (unsigned*)(colInfoScratch + 2 * 256 * (size_t)colInfoChunks);
    static_assert(16 * CB_COUNTERS == 256, "k_cave_columns' first workgroup clears one word per thread");
    LAUNCH(KID_CAVE_COLUMNS, mm::k_cave_columns, dim3(n), dim3(256), s, bw, (const int2*)pos, (float2*)colInfoScratch, chunkList, cbWork);
    // (the per-column pass is small and runs beside whatever the event stands for; the voxel launch is the one that takes the chip)
    if (beforeVoxels) { const hipError_t ew = hipStreamWaitEvent(s, beforeVoxels, 0); if (ew != hipSuccess) return (int)ew; }
    // ... and, where that something is a persistent launch that must be ON THE CHIP first (the relaxation: a few hundred workgroups that
    // spin on each other), one lane watches its started-workgroups counter for a moment: an event only orders the two launches' eligibility,
    // and whichever dispatcher is faster then wins the slots (measured: without this the order flips with a 90 us change upstream)
    if (waitCounter && waitTarget) LAUNCH(KID_CAVE_COLUMNS, mm::k_wait_counter, dim3(1), dim3(64), s, waitCounter, waitTarget);
    static const int wideMax = [] { const char* e = getenv("MMGEN_CAVE_WIDE_MAX_ROWS"); return e ? atoi(e) : CAVE_WIDE_MAX_ROWS; }();      // (A/B: 0 = never)
    if (n * 16 <= wideMax)
        LAUNCH(KID_CAVE_VOXELS, mm::k_cave_voxels_wide, dim3(n * 16), dim3(CAVE_THREADS_WIDE), s, hf, (const float2*)colInfoScratch, (const int2*)pos, caveLayers, chunkList, colNeed);
    else
        LAUNCH(KID_CAVE_VOXELS, mm::k_cave_voxels, dim3(n * 16), dim3(CAVE_THREADS), s, hf, (const float2*)colInfoScratch, (const int2*)pos, caveLayers, chunkList, colNeed);
    // the layers' extents are final here (what the base fill reads); their biomes follow
    if (afterVoxels) { const hipError_t ee = hipEventRecord(afterVoxels, s); if (ee != hipSuccess) return (int)ee; }
    const int cus = device_cus();
    if (!cus) return (int)hipErrorInvalidDevice;
    static const int smallMax = [] { const char* e = getenv("MMGEN_CB_SMALL_MAX_CHUNKS"); return e ? atoi(e) : CB_SMALL_MAX_CHUNKS; }();      // (A/B: 0 = never)
    const bool small = n <= smallMax;
    const int unitCols = small ? CB_UNIT_COLS_SMALL : CB_UNIT_COLS;
    const long long units = (long long)n * (256 / unitCols), fit = (long long)cus * (biomeWorkgroupsPerCu > 0 && biomeWorkgroupsPerCu < MM_CB_WAVES ? biomeWorkgroupsPerCu : MM_CB_WAVES);   // persistent: MM_CB_WAVES 4-wave workgroups per CU at most
    if (n >= (1 << 18)) return (int)hipErrorInvalidValue;                  // item ids carry the list index in 18 bits
    const dim3 grid((unsigned)(units / 4 + 1 < fit ? units / 4 + 1 : fit));
    if (small)
        LAUNCH(KID_CAVE_BIOMES, mm::k_cave_biomes_small, grid, dim3(CB_THREADS), s, hf, (const int2*)pos, caveLayers, chunkList, (int)units, cbWork);
    else
        LAUNCH(KID_CAVE_BIOMES, mm::k_cave_biomes, grid, dim3(CB_THREADS), s, hf, (const int2*)pos, caveLayers, chunkList, (int)units, cbWork);
    return 0;
}

size_t cave_scratch_bytes(int chunks) { return (size_t)chunks * 256 * sizeof(float2) + 64 * CB_COUNTERS; }

// Scratch of one launch_fill call (caller-owned, fill_scratch layout below): the lush queue, the work counters of k_fill_cave, and per
// sub-batch of kFillSub chunks the row lists with their counts / batch prefix / range index.
namespace {
struct FillScratch { unsigned* lush; unsigned lushCap; unsigned* work; int* counts; int* batchStart; int* rangeRow; unsigned* lists; size_t bytes; };
constexpr int kFillBatch = 1 << 14;           // queue entries carry the (batch-relative) chunk index in 15 bits
constexpr int kFillSub = 1 << 13;             // chunks per k_fill_base / k_fill_cave launch: bounds the row lists (393 KB per chunk)
inline size_t align256(size_t b) { return (b + 255) / 256 * 256; }
std::atomic<unsigned> g_lushCapOverride{0u};        // mmgen_debug_set_lush_queue_cap
FillScratch fill_scratch(char* base, int n)      // base may be null: only `bytes` is meaningful then
{
    const size_t nb = (size_t)(n < kFillBatch ? n : kFillBatch), rows = 16 * (size_t)(n < kFillSub ? n : kFillSub);
    FillScratch f;
    size_t o = 0;
    // the work counters directly in front of the queue's count word: one memset clears both (FILL_CLEAR_BYTES)
    f.work = (unsigned*)(base + o); o += 64 * FILL_COUNTERS;
    f.lush = (unsigned*)(base + o); f.lushCap = (unsigned)(2048 * nb); o += align256(4 * (2048 * nb + 1));
    f.counts = (int*)(base + o); o += align256(4 * rows);
    f.batchStart = (int*)(base + o); o += align256(4 * (rows + 1));
    f.rangeRow = (int*)(base + o); o += align256(4 * (rows * (FILL_VOX / 64) + 1));      // ranges of one batch, every voxel listed: the most there can be
    f.lists = (unsigned*)(base + o); o += 4 * (size_t)FILL_VOX * rows;
    f.bytes = o;
    const unsigned cap = g_lushCapOverride.load(std::memory_order_relaxed);
    if (cap && cap < f.lushCap) f.lushCap = cap;
    return f;
}
}  // namespace

size_t fill_queue_bytes(int n) { return n <= 0 ? 0 : fill_scratch((char*)0x1000, n).bytes; }
void debug_set_lush_queue_cap(int entries) { g_lushCapOverride.store(entries > 0 ? (unsigned)entries : 0u, std::memory_order_relaxed); }

int launch_fill(const float* hf, const float* bw, const float* layers, const mmgen_cave_layer* caveLayers, const int32_t* pos, int n,
                uint8_t* blocks, const int* srcIdx, unsigned* scratch, size_t scratchBytes, bool allInPruneDomain, hipStream_t s, bool countersCleared,
                const unsigned** startedCounter, unsigned* startedTarget, hipEvent_t beforeCave)
{
    if (startedCounter) *startedCounter = nullptr;
    if (startedTarget) *startedTarget = 0u;
    if (n <= 0) return 0;
    if (!scratch || scratchBytes < fill_queue_bytes(n)) return (int)hipErrorInvalidValue;
    const FillScratch f = fill_scratch((char*)scratch, n);
    const int cus = device_cus();
    if (!cus) return (int)hipErrorInvalidDevice;
    for (int b0 = 0; b0 < n; b0 += kFillBatch) {
        const int nb = n - b0 < kFillBatch ? n - b0 : kFillBatch;
        // without an index list inputs and outputs are both dense: shift every per-chunk pointer; with one only the list and the output move
        const size_t in0 = srcIdx ? 0 : (size_t)b0;
        const int* idx = srcIdx ? srcIdx + b0 : nullptr;
        uint8_t* out = blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * b0;
        const int2* p = (const int2*)pos + in0;
        const float* hfB = hf + 256 * in0;
        const float* bwB = bw + (size_t)MMGEN_BIOME_WEIGHTS_SIZE * in0;
        const float* layB = layers + (size_t)MMGEN_LAYERS_SIZE * in0;
        const mmgen_cave_layer* clB = caveLayers + (size_t)MMGEN_CAVE_LAYERS_SIZE * in0;
        hipError_t e = hipSuccess;
        // the work counters and the queue's count in one memset (the queue's entries are (re)written by every launch)
        if (!(countersCleared && b0 == 0)) e = hipMemsetAsync(f.work, 0, FILL_CLEAR_BYTES, s);
        if (e != hipSuccess) return (int)e;
        for (int c0 = 0; c0 < nb; c0 += kFillSub) {
            const int nc = nb - c0 < kFillSub ? nb - c0 : kFillSub, nRows = nc * (256 / FILL_ROW), row0 = c0 * (256 / FILL_ROW);
            if (c0 > 0) e = hipMemsetAsync(f.work, 0, 64 * FILL_COUNTERS, s);          // (a second sub-batch of the same batch: the counters only)
            if (e != hipSuccess) return (int)e;
            LAUNCH(KID_FILL_BASE, mm::k_fill_base, dim3(nRows), dim3(FILLB_THREADS), s, hfB, bwB, layB, clB, p, out, idx, row0, f.lists, f.counts);
            // a row lists ~28 batches; enough ranges for every wave to draw a few (a 256-chunk call would otherwise hand 2 ranges to each)
            const long long perWave = 28LL * nRows / ((long long)cus * 4 * MM_FILL_WAVES * 6);
            const int range = perWave < 1 ? 1 : (perWave > FILL_RANGE ? FILL_RANGE : (int)perWave);
            if (beforeCave && b0 == 0 && c0 == 0) { e = hipStreamWaitEvent(s, beforeCave, 0); if (e != hipSuccess) return (int)e; }
            LAUNCH(KID_FILL_SCAN, mm::k_fill_scan, dim3((nRows + 1023) / 1024), dim3(1024), s, (const int*)f.counts, nRows, range, f.batchStart, f.rangeRow);
            // persistent: MM_FILL_WAVES waves per SIMD
            const unsigned cgrid = (unsigned)(cus * (4 * MM_FILL_WAVES / (FILLC_THREADS / 64)));
            LAUNCH(KID_FILL, mm::k_fill_cave, dim3(cgrid), dim3(FILLC_THREADS), s, hfB, p, out, idx, row0, (const unsigned*)f.lists, (const int*)f.counts,
                   (const int*)f.batchStart, (const int*)f.rangeRow, nRows, range, f.lush, f.lushCap, f.work);
            // a caller that wants the cave fill's persistent workgroups on the chip before it starts something beside them watches this
            // word reach the grid size (one sub-batch only: the next one re-uses the counters)
            // (all but a few: the grid is the chip's exact capacity, and a watcher's own wave can keep the last workgroup of one CU waiting -
            // measured, 1535 of 1536 for the whole launch in most steps of a C++ host)
            if (n <= kFillSub) { if (startedCounter) *startedCounter = f.work + 1; if (startedTarget) *startedTarget = cgrid - cgrid / 64; }
        }
        if (!allInPruneDomain)             // rows beyond the pruning domain (k_fill_base leaves them alone)
            LAUNCH(KID_FILL_FAR, mm::k_fill_far, dim3(nb * (256 / FILL_ROW)), dim3(FILL_THREADS), s, hfB, bwB, layB, clB, p, out, idx, f.lush, f.lushCap);
        const unsigned grid = (unsigned)nb * 4 < 2048u ? (unsigned)nb * 4 : 2048u;
        LAUNCH(KID_FILL_LUSH, mm::k_fill_lush, dim3(grid), dim3(256), s, (const unsigned*)f.lush, f.lushCap, p, idx, out);
    }
    return 0;
}

// the two memsets launch_fill starts with, for a caller that has the scratch long before the fill's inputs exist (countersCleared = true)
int launch_fill_clear(int n, unsigned* scratch, size_t scratchBytes, hipStream_t s)
{
    if (n <= 0) return 0;
    if (!scratch || scratchBytes < fill_queue_bytes(n)) return (int)hipErrorInvalidValue;
    const FillScratch f = fill_scratch((char*)scratch, n);
    return (int)hipMemsetAsync(f.work, 0, FILL_CLEAR_BYTES, s);
}

int launch_wait_counter(const unsigned* counter, unsigned target, hipStream_t s)
{
    LAUNCH(KID_CAVE_COLUMNS, mm::k_wait_counter, dim3(1), dim3(64), s, counter, target);
    return 0;
}

int launch_probe(int fn, const float* in, int n, float* out, hipStream_t s)
{
    if (n <= 0) return 0;
    LAUNCH(KID_PROBE, mm::k_probe, dim3((n + 255) / 256), dim3(256), s, fn, in, n, out);
    return 0;
}

}  // namespace mmk
