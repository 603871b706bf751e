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
#include <vector>

namespace mm {

// =========================================================================================================
// K1 — heightfield + 24 biome weights.  GATHERED variant also produces the 18x18 ring the layer stage needs
// (the ring heights are a pure function of position, so no neighbour chunk is read).
// =========================================================================================================
template <bool GATHERED>
__global__ void __launch_bounds__(GATHERED ? 384 : 256)
k_heightfield(const int2* __restrict__ chunkPos, float* __restrict__ hf, float* __restrict__ bw, float* __restrict__ gathered)
{
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
    float height = 0.f;
    float* wout = bw + (size_t)MMGEN_BIOME_WEIGHTS_SIZE * chunk + idx;
    for (int b = 0; b < MMGEN_NUM_BIOMES; ++b) {
        const float w = biome_weight(b, bn);
        if (w > 0.f) height += w * biome_height(b, wx, wz);
        if (interior) wout[256 * b] = w;
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
k_layers(const float* __restrict__ gathered, const float* __restrict__ bw, const int2* __restrict__ chunkPos, float* __restrict__ layers)
{
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

    // forward stratified layers 0..9: start = running height; stop accumulating once above the surface.  Layers after
    // the stop carry the running height (the reference leaves them unwritten; they never influence a block).
    float height = 0.f;
    bool stopped = false;
#pragma unroll
    for (int l = 0; l < MMGEN_NUM_FORWARD_MATERIALS; ++l) {
        out[256 * l] = height;
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
//   k_cave_voxels  : one workgroup (6 waves) per column, lane = y; Worley cell points of the column's reachable
//                    7x8x7 cell box staged in LDS; solid/air bits → wave ballots → (start,end) runs
//   k_cave_biomes  : per (layer slot, column) cave-biome pair, lanes = columns so that occupied slots pack densely
// =========================================================================================================
__global__ void __launch_bounds__(256)
k_cave_columns(const float* __restrict__ bw, const int2* __restrict__ chunkPos, float2* __restrict__ colInfo, const int* __restrict__ chunkList)
{
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

#define CELL_NX 8          // 4 adjacent columns share one tile: +1 cell in x over the single-column reach
#define CELL_NY 8
#define CELL_NZ 7
#define CELL_N (CELL_NX * CELL_NY * CELL_NZ)
#define CAVE_COLS 4        // columns per workgroup (same z row of the chunk, x = 4g .. 4g+3)
#define CAVE_YEVAL 144     // voxels y < 144 may need the noise (threshold is 0 once y + 50*obw >= 142); 144 = 2.25 waves
#define CAVE_THREADS (CAVE_COLS * CAVE_YEVAL)     // 576 = 9 full waves: no partially filled wave

struct CellTile {
    const float* pts;     // LDS, 3 floats per cell
    int ox, oy, oz;
    MM_DEV f3 operator()(int cx, int cy, int cz) const
    {
        const int ix = cx - ox, iy = cy - oy, iz = cz - oz;
        if ((unsigned)ix < CELL_NX && (unsigned)iy < CELL_NY && (unsigned)iz < CELL_NZ) {
            const float* p = pts + 3 * ((ix * CELL_NY + iy) * CELL_NZ + iz);
            return mk3(p[0], p[1], p[2]);
        }
        return rand3from3((float)cx, (float)cy, (float)cz);   // outside the staged box: same value, computed directly
    }
};

#ifndef MM_CAVE_LATTICE
#define MM_CAVE_LATTICE 1
#endif
#define LAT_CAP 64
// per-wave LDS staging of simplex lattice gradients: keys of the distinct (cell, corner ordering) pairs met by the wave's 64 voxels
// and the 4 corner gradients (12 floats) of each
typedef float f4v __attribute__((ext_vector_type(4)));
struct alignas(16) LatticeWave { f4v key[LAT_CAP]; f4v q[LAT_CAP][3]; };
typedef __attribute__((address_space(3))) LatticeWave* LatticePtr;      // LDS pointer: ds_read/ds_write instead of flat_*

// Wave-local LDS hand-off: LDS operations of one wave are executed in issue order, so lanes of the same wave see each other's
// writes without a workgroup barrier; only the compiler has to be kept from reordering across the hand-off.
MM_DEV void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// value of the previous lane (lane 0 gets its own): one DPP move (wave_shr:1)
#ifndef MM_PREV_LANE
#define MM_PREV_LANE 1
#endif
#if MM_PREV_LANE == 0
MM_DEV int prev_lane_i(int v) { return __shfl_up(v, 1); }
#elif MM_PREV_LANE == 1
MM_DEV int prev_lane_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
#else
MM_DEV int prev_lane_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false); }   // row_shr:1, rows of 16
#endif
MM_DEV float prev_lane_f(float v) { return __int_as_float(prev_lane_i(__float_as_int(v))); }

// one lane per queued (cell, ordering) pair computes the 4 corner gradients
static __device__ __attribute__((noinline)) void lattice_build(LatticePtr Lp, int n, int lane)
{
    wave_lds_sync();
    if (lane < n) {
        float q[12];
        const f4v k = Lp->key[lane];
        simplex3_gradients(k.x, k.y, k.z, __float_as_int(k.w), q);
#pragma unroll
        for (int i = 0; i < 3; ++i) Lp->q[lane][i] = f4v{q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]};
    }
    wave_lds_sync();
}

static __device__ __attribute__((noinline)) float lattice_eval(LatticePtr Lp, int slot, float ix, float iy, float iz, float x0x, float x0y,
                                                                 float x0z, int order)
{
    Sx3Cell c; c.ix = ix; c.iy = iy; c.iz = iz; c.x0x = x0x; c.x0y = x0y; c.x0z = x0z; c.order = order;
    float q[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f4v v = Lp->q[slot][i];
        q[4 * i] = v.x; q[4 * i + 1] = v.y; q[4 * i + 2] = v.z; q[4 * i + 3] = v.w;
    }
    return simplex3_part3(c, q);
}

// Evaluates NS simplex3 samples per lane.  Along a column the samples of one site fall into few lattice cells, so instead of
// every lane recomputing the 12 permutation hashes + gradient decode of its cell (≈ 2/3 of simplex3), the wave (1) finds the
// change points of (cell, ordering) along its lanes, (2) lets ONE lane per distinct pair compute the gradients — pairs of
// SEVERAL sites in different lanes of the same pass, flushed whenever the 64-entry table would overflow — and (3) every lane
// fetches its cell's gradients from LDS.  Bit-exact: the gradients are the same function of the same arguments.
// Wave-local (no workgroup barrier): must be called wave-uniformly; waves may take different numbers of flushes.
template <int NS>
MM_DEV void simplex3_sites(const float* sx, const float* sy, const float* sz, float* out, LatticeWave& L, int lane)
{
    Sx3Cell c[NS];
    int slot[NS];
    int base = 0, first = 0;
    const LatticePtr Lp = (LatticePtr)&L;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        c[s] = simplex3_part1(sx[s], sy[s], sz[s]);
        // cross-lane reads first, unconditionally (a short-circuited || would run them under a partial EXEC mask)
        const float pix = prev_lane_f(c[s].ix), piy = prev_lane_f(c[s].iy), piz = prev_lane_f(c[s].iz);
        const int po = prev_lane_i(c[s].order);
        const bool leader = (int)(MM_PREV_LANE == 2 ? (lane & 15) == 0 : lane == 0) | (int)(pix != c[s].ix) | (int)(piy != c[s].iy) |
                            (int)(piz != c[s].iz) | (int)(po != c[s].order);
        const unsigned long long m = __ballot(leader);
        const int cnt = __popcll(m);
        if (base + cnt > LAT_CAP) {            // wave-uniform: flush the queued sites first
            lattice_build(Lp, base, lane);
#pragma unroll
            for (int u = 0; u < NS; ++u)
                if (u >= first && u < s) out[u] = lattice_eval(Lp, slot[u], c[u].ix, c[u].iy, c[u].iz, c[u].x0x, c[u].x0y, c[u].x0z, c[u].order);
            wave_lds_sync();
            first = s;
            base = 0;
        }
        slot[s] = base + __popcll(m & ((2ull << lane) - 1ull)) - 1;
        if (leader) {
            Lp->key[slot[s]] = f4v{c[s].ix, c[s].iy, c[s].iz, __int_as_float(c[s].order)};
        }
        base += cnt;
    }
    lattice_build(Lp, base, lane);
#pragma unroll
    for (int u = 0; u < NS; ++u)
        if (u >= first) out[u] = lattice_eval(Lp, slot[u], c[u].ix, c[u].iy, c[u].iz, c[u].x0x, c[u].x0y, c[u].x0z, c[u].order);
    wave_lds_sync();      // L is reused by the next call
}

// One workgroup = 4 neighbouring columns.  Lane e evaluates voxel (column e / 144, y = e % 144): 576 lanes = 9 FULL waves (a
// 384-lane-per-column mapping pays 3 waves for 142 useful lanes).  Voxels y >= 144 never need noise: solid iff
// y <= min(max((int)h, 128), ravine cut), so their bits are built analytically.  The air/solid bits of all 4 x 384 voxels go to
// LDS as 64-bit words; runs are extracted with popcount prefixes over those words.
#ifndef MM_CAVE_WAVES
#define MM_CAVE_WAVES 6          // 80 VGPRs: 2 workgroups of 9 waves per CU (the default allocation, 98 VGPRs, fits only one)
#endif
__attribute__((amdgpu_waves_per_eu(MM_CAVE_WAVES, MM_CAVE_WAVES)))
__global__ void __launch_bounds__(CAVE_THREADS)
k_cave_voxels(const float* __restrict__ hf, const float2* __restrict__ colInfo, const int2* __restrict__ chunkPos,
              mmgen_cave_layer* __restrict__ caveLayers, const int* __restrict__ chunkList)
{
    __shared__ float s_cells[3 * CELL_N];
    __shared__ unsigned long long s_solid[CAVE_COLS][6];      // solid bit of voxel y at word y / 64, bit y % 64
#if MM_CAVE_LATTICE
    __shared__ LatticeWave s_lat[CAVE_THREADS / 64];
#endif
    __shared__ int s_layers[CAVE_COLS][3 * MMGEN_MAX_CAVE_LAYERS_PER_COLUMN];

    const int t = threadIdx.x;
    const int chunk = chunkList ? chunkList[blockIdx.x >> 6] : (int)(blockIdx.x >> 6);
    const int group = blockIdx.x & 63;                         // 64 groups of 4 columns per chunk
    const int c = t / CAVE_YEVAL, y = t - c * CAVE_YEVAL;
    const int idx2d = 4 * group + c;                           // x = 4 (group % 4) + c, z = group / 4
    const int col = chunk * 256 + idx2d;
    const int2 cp = chunkPos[chunk];
    const int wx = cp.x + (idx2d & 15), wz = cp.y + (idx2d >> 4);
    const float maxHeight = hf[col];
    const float2 ci = colInfo[col];
    const float obw = ci.x, ravineY = ci.y;

    const float npx = (float)wx * 0.0050f, npz = (float)wz * 0.0050f;
    // cell tile: sample position = noisePos * (1, 1.6, 1) + offset with |offset| < 1.8; origin from the group's first column
    CellTile tile;
    tile.pts = s_cells;
    tile.ox = (int)__builtin_floorf(((float)(cp.x + ((4 * group) & 15)) * 0.0050f) * 1.f) - 3;
    tile.oy = -3;
    tile.oz = (int)__builtin_floorf(npz * 1.f) - 3;
    if (t < CELL_N) {
        const int iz = t % CELL_NZ, iy = (t / CELL_NZ) % CELL_NY, ix = t / (CELL_NZ * CELL_NY);
        const f3 p = rand3from3((float)(tile.ox + ix), (float)(tile.oy + iy), (float)(tile.oz + iz));
        s_cells[3 * t] = p.x; s_cells[3 * t + 1] = p.y; s_cells[3 * t + 2] = p.z;
    }
    if (t < CAVE_COLS * 6) s_solid[t / 6][t % 6] = 0ull;
    if (t < CAVE_COLS * 3 * MMGEN_MAX_CAVE_LAYERS_PER_COLUMN) (&s_layers[0][0])[t] = ((t % 3) == 2) ? 0 : 384;   // {384, 384, biomes = 0}
    __syncthreads();

    const int topSolid = imax((int)maxHeight, MMGEN_SEA_LEVEL);
    const float fy = (float)y;
    const float npy = fy * 0.0050f;
    const bool inBand = (y != 0) && (y <= topSolid);
    const float topRatio = smoothstep(142.f, 95.f, fy + obw * 50.f);
    const float bottomRatio = smoothstep(5.f, 20.f, fy);
    const bool needThr = inBand && topRatio > 0.f;      // threshold is a product with topRatio: 0 → "threshold > 0.04" is false
    bool cave = (y != 0) && !inBand;                    // y == 0 solid, y > topSolid air
#if MM_CAVE_LATTICE
    // All 23 simplex3 evaluations of a voxel go through simplex3_sites: every wave shares the lattice gradients of the cells its
    // 64 voxels fall into (stage structure is uniform over the workgroup; lanes that do not need a value simply ignore it).
    LatticeWave& L = s_lat[t >> 6];
    const int lane = t & 63;
    float thr = 0.f;
    if (__ballot(needThr) != 0ull) {       // wave-uniform
        float sx[4], sy[4], sz[4], v[4];
        float ax = npx * 4.f, ay = npy * 4.f, az = npz * 4.f;
#pragma unroll
        for (int o = 0; o < 4; ++o) { sx[o] = ax; sy[o] = ay; sz[o] = az; ax *= 2.f; ay *= 2.f; az *= 2.f; }
        simplex3_sites<4>(sx, sy, sz, v, L, lane);
        const float fa = (((0.f + 0.5f * v[0]) + 0.25f * v[1]) + 0.125f * v[2]) + 0.0625f * v[3];
        float bx = npx * 0.0700f, by = npy * 0.0700f, bz = npz * 0.0700f;
#pragma unroll
        for (int o = 0; o < 4; ++o) { sx[o] = bx; sy[o] = by; sz[o] = bz; bx *= 2.f; by *= 2.f; bz *= 2.f; }
        simplex3_sites<4>(sx, sy, sz, v, L, lane);
        const float fb = (((0.f + 0.5f * v[0]) + 0.25f * v[1]) + 0.125f * v[2]) + 0.0625f * v[3];
        thr = 0.24f + 0.12f * fa;
        const float huge = smoothstep(0.2f, 0.4f, fb);
        thr *= (1.f + 1.4f * huge);
        thr *= topRatio * (0.3f + 0.7f * bottomRatio);
    }
    const bool needWarp = needThr && thr > 0.04f;
    float warp[3] = {0.f, 0.f, 0.f};
    if (__ballot(needWarp) != 0ull) {      // wave-uniform
        const float offx[3] = {0.f, 5923.45f, 1765.68f}, offy[3] = {0.f, 4129.42f, 4704.36f}, offz[3] = {0.f, 5790.48f, 5692.12f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float sx[5], sy[5], sz[5], v[5];
            float wx0 = npx * 0.8000f, wy0 = npy * 0.8000f, wz0 = npz * 0.8000f;
            if (k > 0) { wx0 += offx[k]; wy0 += offy[k]; wz0 += offz[k]; }
#pragma unroll
            for (int o = 0; o < 5; ++o) { sx[o] = wx0; sy[o] = wy0; sz[o] = wz0; wx0 *= 2.f; wy0 *= 2.f; wz0 *= 2.f; }
            simplex3_sites<5>(sx, sy, sz, v, L, lane);
            warp[k] = ((((0.f + 0.5f * v[0]) + 0.25f * v[1]) + 0.125f * v[2]) + 0.0625f * v[3]) + 0.03125f * v[4];
        }
    }
    if (needWarp) {
        const float n = special_cave_noise(npx * 1.f + warp[0] * 1.8f, npy * 1.6f + warp[1] * 1.8f, npz * 1.f + warp[2] * 1.8f, tile);
        cave = n < thr;
    }
    if (inBand && !cave) cave = fy > ravineY;
#else
    if (needThr) {
        float thr = 0.24f + 0.12f * fbm3<4>(npx * 4.f, npy * 4.f, npz * 4.f);
        const float huge = smoothstep(0.2f, 0.4f, fbm3<4>(npx * 0.0700f, npy * 0.0700f, npz * 0.0700f));
        thr *= (1.f + 1.4f * huge);
        thr *= topRatio * (0.3f + 0.7f * bottomRatio);
        if (thr > 0.04f) {
            const f3 o = fbm3from3<5>(npx * 0.8000f, npy * 0.8000f, npz * 0.8000f);
            const float n = special_cave_noise(npx * 1.f + o.x * 1.8f, npy * 1.6f + o.y * 1.8f, npz * 1.f + o.z * 1.8f, tile);
            cave = n < thr;
        }
    }
    if (inBand && !cave) cave = fy > ravineY;
#endif
    // the wave's 64 lanes may straddle two columns / two 64-bit words: OR each lane's bit into its word
    if (!cave) atomicOr(&s_solid[c][y >> 6], 1ull << (y & 63));
    // analytic part, y in [144, 384): solid iff y <= topSolid and not (y > ravineY)   (topRatio == 0 there)
    if (y < 4) {               // 4 lanes per column fill words 2..5 (word 2 holds y 128..191: bits >= 16 only)
        const int w = 2 + y;
        unsigned long long m = 0ull;
        for (int b = 0; b < 64; ++b) {
            const int yy = 64 * w + b;
            if (yy >= CAVE_YEVAL && yy <= topSolid && !((float)yy > ravineY)) m |= 1ull << b;
        }
        if (m) atomicOr(&s_solid[c][w], m);
    }
    __syncthreads();

    // flips: solid(y) != solid(y+1), y = 383 compares with "not solid"; rank by popcount prefix; 4 x 384 voxels over 576 lanes
    for (int v = t; v < CAVE_COLS * 384; v += CAVE_THREADS) {
        const int cc = v / 384, yy = v - cc * 384;
        const int w = yy >> 6, b = yy & 63;
        int before = 0;
        unsigned long long mine = 0ull;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const unsigned long long m = s_solid[cc][k];
            const unsigned long long nl = (k < 5) ? (s_solid[cc][k + 1] & 1ull) : 0ull;
            const unsigned long long f = m ^ ((m >> 1) | (nl << 63));
            if (k < w) before += __popcll(f);
            if (k == w) mine = f;
        }
        if ((mine >> b) & 1ull) {
            const int rank = before + __popcll(mine & ((1ull << b) - 1ull));
            if (rank < 2 * MMGEN_MAX_CAVE_LAYERS_PER_COLUMN)       // canonical: runs beyond 32 layers are dropped
                s_layers[cc][3 * (rank >> 1) + (rank & 1)] = yy;
        }
    }
    __syncthreads();
    if (t < CAVE_COLS * 3 * MMGEN_MAX_CAVE_LAYERS_PER_COLUMN) {
        const int cc = t / (3 * MMGEN_MAX_CAVE_LAYERS_PER_COLUMN), k = t % (3 * MMGEN_MAX_CAVE_LAYERS_PER_COLUMN);
        ((int*)(caveLayers + (size_t)MMGEN_MAX_CAVE_LAYERS_PER_COLUMN * (chunk * 256 + 4 * group + cc)))[k] = s_layers[cc][k];
    }
}

__global__ void __launch_bounds__(256)
k_cave_biomes(const float* __restrict__ hf, const int2* __restrict__ chunkPos, mmgen_cave_layer* __restrict__ caveLayers,
              const int* __restrict__ chunkList)
{
    const int chunk = chunkList ? chunkList[blockIdx.x >> 5] : (int)(blockIdx.x >> 5), k = blockIdx.x & 31;
    const int t = threadIdx.x;
    mmgen_cave_layer* L = caveLayers + ((size_t)256 * chunk + t) * MMGEN_MAX_CAVE_LAYERS_PER_COLUMN + k;
    const int start = L->start;
    if (start == 384) return;           // unused slot: biomes stay NONE (0)
    const int end = L->end;
    const int2 cp = chunkPos[chunk];
    const int wx = cp.x + (t & 15), wz = cp.y + (t >> 4);
    const float maxHeight = hf[(size_t)256 * chunk + t];
    const int bottom = cave_biome(wx, start, wz, maxHeight, 329271348);
    const int top = (end == 384) ? MMCB_NONE : cave_biome(wx, end + 1, wz, maxHeight, 4982921);
    L->bottom_biome = (uint8_t)bottom;
    L->top_biome = (uint8_t)top;
}

// =========================================================================================================
// K6 — fill.  One workgroup = 4 neighbouring columns (12 waves).  Two phases inside the workgroup:
//   1. every voxel gets its base block (bedrock / air / water / cave air / layer material + surface-biome rules): cheap, lane = y;
//   2. the voxels whose block a cave biome could still alter (STONE / DEEPSLATE / BLACKSTONE below ground — the only blocks
//      caveBiomeBlockPostProcess touches) are compacted into an LDS list and processed densely: the cave-biome evaluation
//      (9 simplex3D + 12 simplex2D) is the expensive part of fill, and a dense list keeps every lane of every wave on it instead
//      of interleaving it with lanes that returned "air" long ago.  Order inside the list is irrelevant (voxels are independent).
// =========================================================================================================
struct BaseBlock { uint8_t block; bool needCave; int bottomDepth, topDepth; };

MM_DEV BaseBlock place_block_base(const float* s_bw, const float* s_lh, const mmgen_cave_layer* s_cl, int y, float height, int wx, int wz)
{
    BaseBlock r; r.needCave = false; r.bottomDepth = -384; r.topDepth = -384;
    if (y == 0) { r.block = MMB_BEDROCK; return r; }
    const float fy = (float)y;
    if (fy > height && y > MMGEN_SEA_LEVEL) { r.block = MMB_AIR; return r; }

    bool isOcean = false;
#pragma unroll
    for (int b = 0; b < MMGEN_NUM_OCEAN_BIOMES; ++b) isOcean = isOcean || (s_bw[b] > 0.f);

    MinStd rng = rng3(wx, y, wz);
    const int randBiome = random_biome(s_bw, 1, rng.u01());
    const bool isTop = fy >= height - 1.f;

    uint8_t block = MMB_AIR;
    if (fy > height && y <= MMGEN_SEA_LEVEL) {
        block = MMB_WATER;
        biome_block_post(block, randBiome, wx, y, wz, isTop);
        if (isOcean) { r.block = block; return r; }
    }

    int bottomDepth = -384, topDepth = -384;
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
    for (int l = l0; l < MMGEN_NUM_MATERIALS; ++l) {
        if (s_lh[l] <= fy && fy < s_lh[l + 1]) { layer = l; break; }
    }
    block = (layer < 0) ? (uint8_t)MMB_STONE : kMaterialBlock[layer];   // canonical: no layer (y == height exactly) → STONE
    if (isTop && block == MMB_DIRT) block = kGrassBlock[randBiome];

    biome_block_post(block, randBiome, wx, y, wz, isTop);
    r.block = block;
    r.needCave = cave_post_can_apply(block);
    r.bottomDepth = bottomDepth;
    r.topDepth = topDepth;
    return r;
}

#define FILL_COLS 4
#define FILL_THREADS 768

__global__ void __launch_bounds__(FILL_THREADS)
k_fill(const float* __restrict__ hf, const float* __restrict__ bw, const float* __restrict__ layers,
       const mmgen_cave_layer* __restrict__ caveLayers, const int2* __restrict__ chunkPos, uint8_t* __restrict__ blocks,
       const int* __restrict__ srcIdx)
{
    __shared__ float s_bw[FILL_COLS][MMGEN_NUM_BIOMES];
    __shared__ float s_lh[FILL_COLS][MMGEN_NUM_MATERIALS + 1];
    __shared__ mmgen_cave_layer s_cl[FILL_COLS][MMGEN_MAX_CAVE_LAYERS_PER_COLUMN];
    __shared__ unsigned int s_list[FILL_COLS * 384];          // voxel (11 bits) | block (8) | 13 spare
    __shared__ int s_depth[FILL_COLS * 384];                  // bottomDepth (low 16, signed) | topDepth (high 16, signed)
    __shared__ int s_count;

    const int t = threadIdx.x;
    const int outChunk = blockIdx.x >> 6, group = blockIdx.x & 63;
    const int chunk = srcIdx ? srcIdx[outChunk] : outChunk;        // inputs are read at `chunk`, blocks are written densely at outChunk
    const int idxBase = 4 * group;

    // stage the 4 columns: 24 weights + 20 layer starts + height + 96 cave-layer words each = 141 words per column
    if (t < FILL_COLS * 141) {
        const int c = t / 141, k = t % 141;
        const int idx2d = idxBase + c;
        if (k < MMGEN_NUM_BIOMES) s_bw[c][k] = bw[(size_t)MMGEN_BIOME_WEIGHTS_SIZE * chunk + 256 * k + idx2d];
        else if (k < MMGEN_NUM_BIOMES + MMGEN_NUM_MATERIALS) s_lh[c][k - MMGEN_NUM_BIOMES] = layers[(size_t)MMGEN_LAYERS_SIZE * chunk + 256 * (k - MMGEN_NUM_BIOMES) + idx2d];
        else if (k == MMGEN_NUM_BIOMES + MMGEN_NUM_MATERIALS) s_lh[c][MMGEN_NUM_MATERIALS] = hf[chunk * 256 + idx2d];
        else ((int*)s_cl[c])[k - 45] = ((const int*)(caveLayers + (size_t)MMGEN_MAX_CAVE_LAYERS_PER_COLUMN * (chunk * 256 + idx2d)))[k - 45];
    }
    if (t == 0) s_count = 0;
    __syncthreads();

    const int2 cp = chunkPos[chunk];
    uint8_t* outBase = blocks + (size_t)MMGEN_BLOCKS_PER_CHUNK * outChunk + 384 * idxBase;     // the 4 columns are contiguous: 1536 bytes

    // phase 1: base blocks
    for (int v = t; v < FILL_COLS * 384; v += FILL_THREADS) {
        const int c = v / 384, y = v - 384 * c;
        const int idx2d = idxBase + c;
        const int wx = cp.x + (idx2d & 15), wz = cp.y + (idx2d >> 4);
        const BaseBlock r = place_block_base(s_bw[c], s_lh[c], s_cl[c], y, s_lh[c][MMGEN_NUM_MATERIALS], wx, wz);
        if (r.needCave) {
            const int slot = atomicAdd(&s_count, 1);
            s_list[slot] = (unsigned)v | ((unsigned)r.block << 11);
            s_depth[slot] = (r.bottomDepth & 0xffff) | (r.topDepth << 16);
        } else {
            outBase[v] = r.block;
        }
    }
    __syncthreads();

    // phase 2: cave-biome rules on the compacted stone voxels
    const int count = s_count;
    for (int i = t; i < count; i += FILL_THREADS) {
        const unsigned e = s_list[i];
        const int v = e & 2047;
        uint8_t block = (uint8_t)(e >> 11);
        const int d = s_depth[i];
        const int bottomDepth = (int)(short)(d & 0xffff), topDepth = d >> 16;
        const int c = v / 384, y = v - 384 * c;
        const int idx2d = idxBase + c;
        const int wx = cp.x + (idx2d & 15), wz = cp.y + (idx2d >> 4);
        const int cb = cave_biome(wx, y, wz, s_lh[c][MMGEN_NUM_MATERIALS], 190249401);
        cave_biome_block_post(block, cb, wx, y, wz, bottomDepth, topDepth);
        outBase[v] = block;
    }
}

// =========================================================================================================
// Debug probes: evaluate one device function per item so that tests can pin the device math against the golden
// vectors (real glm simplex, frozen KATs) through the C ABI.  Inputs/outputs are packed fp32 (ints bit-cast).
// =========================================================================================================
__global__ void __launch_bounds__(256) k_probe(int fn, const float* __restrict__ in, int n, float* __restrict__ out)
{
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
        simplex3_gradients(c.ix, c.iy, c.iz, c.order, q);
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

// ---------------------------------------------------------------------------------------------------------
// Optional per-kernel timing with HIP events on the launch stream (used by bench.py for the roofline line; off by default).
// ---------------------------------------------------------------------------------------------------------
struct ProfRec { int id; hipEvent_t a, b; };
static bool g_prof = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static const char* const kKernelNames[] = {"k_heightfield", "k_layers", "k_fix_backward", "k_cave_columns", "k_cave_voxels", "k_cave_biomes", "k_fill", "k_probe"};
enum { KID_HEIGHTFIELD, KID_LAYERS, KID_FIX_BACKWARD, KID_CAVE_COLUMNS, KID_CAVE_VOXELS, KID_CAVE_BIOMES, KID_FILL, KID_PROBE, KID_COUNT };

static hipEvent_t get_event()
{
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}

void profile_enable(bool on) { g_prof = on; }
int profile_num_kernels() { return KID_COUNT; }
const char* profile_kernel_name(int id) { return (id >= 0 && id < KID_COUNT) ? kKernelNames[id] : ""; }
// Synchronises the recorded events, accumulates total milliseconds and launch counts per kernel id, and clears the records.
int profile_collect(double* total_ms, long long* counts)
{
    for (int i = 0; i < KID_COUNT; ++i) { total_ms[i] = 0; counts[i] = 0; }
    for (auto& r : g_recs) {
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e != hipSuccess) return (int)e;
        total_ms[r.id] += ms; counts[r.id] += 1;
        g_pool.push_back(r.a); g_pool.push_back(r.b);
    }
    g_recs.clear();
    return 0;
}

#define LAUNCH(KID, KERNEL, GRID, BLOCK, STREAM, ...)                                        \
    do {                                                                                     \
        ProfRec rec_; rec_.id = (KID);                                                       \
        if (g_prof) { rec_.a = get_event(); rec_.b = get_event(); (void)hipEventRecord(rec_.a, (STREAM)); } \
        hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, (STREAM), __VA_ARGS__);                   \
        if (g_prof) { (void)hipEventRecord(rec_.b, (STREAM)); g_recs.push_back(rec_); }      \
        hipError_t e_ = hipGetLastError();                                                   \
        if (e_ != hipSuccess) return (int)e_;                                                \
    } while (0)

int launch_heightfield(const int32_t* pos, int n, float* hf, float* bw, float* gathered, hipStream_t s)
{
    if (n <= 0) return 0;
    if (gathered) LAUNCH(KID_HEIGHTFIELD, mm::k_heightfield<true>, dim3(n), dim3(384), s, (const int2*)pos, hf, bw, gathered);
    else LAUNCH(KID_HEIGHTFIELD, mm::k_heightfield<false>, dim3(n), dim3(256), s, (const int2*)pos, hf, bw, (float*)nullptr);
    return 0;
}

int launch_layers(const float* gathered, const float* bw, const int32_t* pos, int n, float* layers, hipStream_t s)
{
    if (n <= 0) return 0;
    LAUNCH(KID_LAYERS, mm::k_layers, dim3(n), dim3(256), s, gathered, bw, (const int2*)pos, layers);
    return 0;
}

int launch_fix_backward(float* layers, int n, hipStream_t s)
{
    if (n <= 0) return 0;
    LAUNCH(KID_FIX_BACKWARD, mm::k_fix_backward, dim3(n), dim3(256), s, layers);
    return 0;
}

int launch_caves(const float* hf, const float* bw, const int32_t* pos, int n, mmgen_cave_layer* caveLayers, float* colInfoScratch,
                 const int* chunkList, hipStream_t s)
{
    if (n <= 0) return 0;
    LAUNCH(KID_CAVE_COLUMNS, mm::k_cave_columns, dim3(n), dim3(256), s, bw, (const int2*)pos, (float2*)colInfoScratch, chunkList);
    LAUNCH(KID_CAVE_VOXELS, mm::k_cave_voxels, dim3(n * 64), dim3(CAVE_THREADS), s, hf, (const float2*)colInfoScratch, (const int2*)pos, caveLayers, chunkList);
    LAUNCH(KID_CAVE_BIOMES, mm::k_cave_biomes, dim3(n * 32), dim3(256), s, hf, (const int2*)pos, caveLayers, chunkList);
    return 0;
}

int launch_fill(const float* hf, const float* bw, const float* layers, const mmgen_cave_layer* caveLayers, const int32_t* pos, int n,
                uint8_t* blocks, const int* srcIdx, hipStream_t s)
{
    if (n <= 0) return 0;
    LAUNCH(KID_FILL, mm::k_fill, dim3(n * 64), dim3(FILL_THREADS), s, hf, bw, layers, caveLayers, (const int2*)pos, blocks, srcIdx);
    return 0;
}

int launch_probe(int fn, const float* in, int n, float* out, hipStream_t s)
{
    if (n <= 0) return 0;
    LAUNCH(KID_PROBE, mm::k_probe, dim3((n + 255) / 256), dim3(256), s, fn, in, n, out);
    return 0;
}

}  // namespace mmk
