// mmgen device noise library for gfx950: permutation-polynomial simplex 2D/3D with glm 0.9.9.8's exact fp32
// operation order (reference math spec: external/include/glm/gtc/noise.inl:591-721, detail/_noise.hpp:8-60),
// fbm stacks, sin-hash cell points and Worley distance searches (reference: src/util/rng.hpp:102-320).
// Scalar formulation: every lane carries one sample; no vector temporaries.
//
// LDS tables.  The path is VALU-issue bound, so the integer-valued part of simplex noise is served from LDS: glm's permute() only ever
// sees small integer-valued floats (lattice coordinates mod 289 plus earlier permute results), and the gradient of a lattice corner is a
// function of the final permute value alone.  Every workgroup loads
//   e[v]      = {simplex3 corner gradient * taylorInvSqrt of permute(v), 16 * permute(v)},  v in [-2, 580): every value the chain feeds to permute
//   grad2[j]  = {simplex2 (a0, h, norm factor) of permute(j - 1), 16 * permute(j - 1)},     j - 1 in [-1, 294]
// at kernel entry (noise_tables_init) from a per-device image that k_noise_tables_build computed with the SAME fp32 instruction
// sequences the direct evaluation uses, so a lookup returns bit for bit what the arithmetic would have produced.  A simplex3 then costs
// 9 LDS reads + 142 VALU instead of 325 VALU, a simplex2 5 reads + ~75 instead of 151.  Domain: mod289 of an integer-valued |x| < 2^24
// lies in [-1, 289] and permute of [-16, 700) in [0, 288] (tests/test_oracle_math.py::test_noise_table_domains); lattice coordinates
// beyond the table domains (2^23 simplex3, 2^21 simplex2; reached only far out in the int32 world, tests "far coordinates") take the
// direct arithmetic path.  EVERY kernel that can reach simplex2/simplex3 calls noise_tables_init() before its first use.
#pragma once
#include <atomic>
#include <mutex>
#include "mm_math.cuh"

namespace mm {

// Code-size control: the voxel kernels evaluate simplex noise at 20-40 call sites; fully inlined that is 80-110 KB of straight-line
// code per kernel, larger than the instruction cache.  MM_SIMPLEX_ATTR selects one shared (non-inlined) body per translation unit.
#ifndef MM_SIMPLEX_ATTR
#define MM_SIMPLEX_ATTR static __device__ __attribute__((noinline))
#endif

struct f2 { float x, y; };
struct f3 { float x, y, z; };
typedef float f4v __attribute__((ext_vector_type(4)));

MM_DEV f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
MM_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }

// ---------------------------------------------------------------------------------------------------------
// simplex: arithmetic pieces (glm order) and the LDS tables built from them
// ---------------------------------------------------------------------------------------------------------
MM_DEV float mod289(float x) { return x - __builtin_floorf(x * (1.f / 289.f)) * 289.f; }
MM_DEV float permute(float x) { return mod289(((x * 34.f) + 1.f) * x); }

// simplex2: gradient terms of a corner from its permute value p: (a0, h, 1.79284291400159 - 0.85373472095314 * (a0^2 + h^2))
MM_DEV f3 simplex2_corner(float p)
{
    const float C3 = (float)0.024390243902439;
    const float K1 = (float)1.79284291400159, K2 = (float)0.85373472095314;
    const float X = 2.f * fract(p * C3) - 1.f;
    const float h = __builtin_fabsf(X) - 0.5f;
    const float a = X - __builtin_floorf(X + 0.5f);
    return mk3(a, h, K1 - K2 * (a * a + h * h));
}

// simplex3: gradient of a corner from its permute value p, already scaled by taylorInvSqrt
MM_DEV f3 simplex3_corner(float p)
{
    const float n_ = (float)0.142857142857;
    const float nsx = n_ * 2.f - 0.f, nsy = n_ * 0.5f - 1.f, nsz = n_ * 1.f - 0.f;
    const float K1 = (float)1.79284291400159, K2 = (float)0.85373472095314;
    const float j = p - 49.f * __builtin_floorf((p * nsz) * nsz);
    const float x_ = __builtin_floorf(j * nsz);
    const float y_ = __builtin_floorf(j - 7.f * x_);
    const float px = x_ * nsx + nsy;
    const float py = y_ * nsx + nsy;
    const float hh = (1.f - __builtin_fabsf(px)) - __builtin_fabsf(py);
    // b0 = (x.x, x.y, y.x, y.y), b1 = (x.z, x.w, y.z, y.w); s = floor(b)*2+1; sh = -step(h, 0); a = b + s * sh
    const float sh = -((0.f < hh) ? 0.f : 1.f);
    const float sx = __builtin_floorf(px) * 2.f + 1.f;
    const float sy = __builtin_floorf(py) * 2.f + 1.f;
    const float g0 = px + sx * sh;
    const float g1 = py + sy * sh;
    const float nrm = K1 - K2 * ((g0 * g0 + g1 * g1) + hh * hh);
    return mk3(g0 * nrm, g1 * nrm, hh * nrm);
}

#define MM_GRAD2_N 296
// Fused last level: the gradient of a corner is looked up directly with the value that used to go into the last permute, (b + x + o)
// with b in [0, 288], x in [-1, 289], o in {0, 1}.
// simplex3 (4 corners, two perm levels): ONE table of 16-byte entries indexed by the value v fed to permute - entry v = {gradient of
//   permute(v) (3 floats), 16 * permute(v)}.  Every index of the chain is such a value: z and z + 1 (>= -1), p + y + o and b + x + o (both
//   in [-1, 578]), and the perm word is the byte offset of the next entry as it stands: a corner is add3, read, add3, read.  The table
//   covers the whole range, every entry computed from its own argument with the functions above - no reduction of an index mod 289, no
//   assumption about permute's period.  (History: 296 gradient entries + a wrap of the index cost 10 VALU more per simplex3; 12-byte
//   gradient entries read as one ds_read_b96 at a 4-byte boundary are legal and took k_cave_voxels from 11.5 to 24.9 ms.)
// simplex2 (3 corners, one perm level): 296 entries of 16 bytes, indices of 289 and beyond wrap, j' = min_u32(j, j - 289) in [0, 290]
//   (permute has period 289 exactly in fp32 on this range, tests/test_oracle_math.py::test_noise_table_domains).  The fourth word of entry
//   j is 16 * permute(j - 1), the one perm level simplex2 needs (j - 1 in [0, 289]).
// Two objects (known LDS bases, immediate offsets in the non-inlined callees): a kernel that never reaches simplex2 does not reference
// s_noise2 and does not pay its 4.6 KB.
#define MM_T3_LO 2
#define MM_T3_N 582
struct alignas(16) NoiseTables3 { f4v e[MM_T3_N]; };
struct alignas(16) NoiseTables2 { f4v grad2[MM_GRAD2_N]; };
static_assert(sizeof(NoiseTables3) % 16 == 0 && sizeof(NoiseTables2) % 16 == 0, "copied as 16-byte words");
static __shared__ NoiseTables3 s_noise;
static __shared__ NoiseTables2 s_noise2;

typedef __attribute__((address_space(3))) const char* lds_bytes;
// off16 = 16 * v: 16 * permute(v) resp. the gradient of permute(v)
MM_DEV int perm16u(int off16) { return *(__attribute__((address_space(3))) const int*)((lds_bytes)s_noise.e + 16 * MM_T3_LO + 12 + off16); }
MM_DEV f4v grad3u(int off16) { return *(__attribute__((address_space(3))) const f4v*)((lds_bytes)s_noise.e + 16 * MM_T3_LO + off16); }
// simplex2: off16 = 16 * index, index in [0, 289]; returns 16 * permute(index)
MM_DEV int perm16_of2(int off16) { return *(__attribute__((address_space(3))) const int*)((lds_bytes)s_noise2.grad2 + 16 + 12 + off16); }
// s16 = 16 * (b + x + o + 1): 16 x the table index of the gradient of permute(b + x + o), before the wrap at 289 (j' = min_u32(j, j - 289));
// the callers fold the "+ 1" into the corner constant, so a corner costs add3 + add + min
MM_DEV int grad_wrap16(int s16) { const unsigned j = (unsigned)s16; const unsigned w = j - 16u * 289u; return (int)(w < j ? w : j); }
MM_DEV f4v grad2_at16(int s16) { return *(__attribute__((address_space(3))) const f4v*)((lds_bytes)s_noise2.grad2 + grad_wrap16(s16)); }

// The tables are built ONCE per device and translation unit by k_noise_tables_build (below, with the arithmetic functions above)
// into this global image; every workgroup then just copies the 14 KB image (9.3 + 4.7) into LDS at kernel entry (16-byte words, L2 resident)
// instead of recomputing 878 entries - 2 - 9 % of the noise kernels' time went into that.
static __device__ NoiseTables3 g_noise;
static __device__ NoiseTables2 g_noise2;

static __global__ void __launch_bounds__(256) k_noise_tables_build()
{
    const int t = threadIdx.x;
    for (int i = t; i < MM_T3_N; i += 256) {
        const float p = permute((float)(i - MM_T3_LO));
        const f3 g3 = simplex3_corner(p);
        g_noise.e[i] = f4v{g3.x, g3.y, g3.z, __int_as_float(16 * (int)p)};
    }
    for (int i = t; i < MM_GRAD2_N; i += 256) {
        const float p = permute((float)(i - 1));
        const f3 g2 = simplex2_corner(p);
        g_noise2.grad2[i] = f4v{g2.x, g2.y, g2.z, __int_as_float(16 * (int)p)};
    }
}

// Host side: mmgen_init() builds the image of every translation unit on the selected device (mmk::prepare_*), so launches after
// init only test a flag here (safe under stream capture).  A launch on a device that was never initialised builds the image on
// the caller's stream and waits for it, so that launches on other streams can never see a half-built image.
static inline int noise_tables_ensure(hipStream_t s)
{
    static std::atomic<bool> built[64];
    static std::mutex mu;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    if (built[dev].load(std::memory_order_acquire)) return 0;
    std::lock_guard<std::mutex> lk(mu);
    if (built[dev].load(std::memory_order_relaxed)) return 0;
    hipLaunchKernelGGL(k_noise_tables_build, dim3(1), dim3(256), 0, s);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return (int)e;
    built[dev].store(true, std::memory_order_release);
    return 0;
}

// Called by every thread of the workgroup at kernel entry (ends with a workgroup barrier).  NEED2 = false: the kernel never
// evaluates simplex2 (it must not: the table is not loaded, and its LDS is not allocated).
template <bool NEED2 = true>
MM_DEV void noise_tables_init()
{
    const int nt = blockDim.x * blockDim.y * blockDim.z;
    const int t = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    constexpr int n3 = (int)(sizeof(NoiseTables3) / 16), n2 = (int)(sizeof(NoiseTables2) / 16);
    const uint4* src3 = (const uint4*)&g_noise;
    uint4* dst3 = (uint4*)&s_noise;
    for (int i = t; i < n3; i += nt) dst3[i] = src3[i];
    if (NEED2) {
        const uint4* src2 = (const uint4*)&g_noise2;
        uint4* dst2 = (uint4*)&s_noise2;
        for (int i = t; i < n2; i += nt) dst2[i] = src2[i];
    }
    __syncthreads();
}

MM_DEV float falloff_max0(float x) { return __builtin_fmaxf(x, 0.f); }

template <bool KNOWN_IN = false>
MM_DEV float simplex2_inl(float vx, float vy)
{
    const float C0 = (float)0.211324865405187, C1 = (float)0.366025403784439;
    const float C2 = (float)-0.577350269189626;

    const float d = vx * C1 + vy * C1;
    float ix = __builtin_floorf(vx + d), iy = __builtin_floorf(vy + d);
    const float e = ix * C0 + iy * C0;
    const float x0x = (vx - ix) + e, x0y = (vy - iy) + e;

    const bool gt = x0x > x0y;
    const float i1x = gt ? 1.f : 0.f, i1y = gt ? 0.f : 1.f;
    const float ax = (x0x + C0) - i1x, ay = (x0y + C0) - i1y;     // x12.xy
    const float bx = x0x + C2, by = x0y + C2;                     // x12.zw

    // table domain: lattice coordinates below 2^21 in magnitude (see the remainder below); NaN fails the test
    const bool inDomain = KNOWN_IN || (__builtin_fabsf(ix) < 2097152.f && __builtin_fabsf(iy) < 2097152.f);

    // glm::max(x, 0) = (x < 0) ? 0 : x keeps a -0 that v_max_f32 turns into +0; the value is squared next, (-0)^2 = (+0)^2 = +0, and NaN
    // positions are outside every caller's domain: one instruction instead of compare + select
    float m0 = falloff_max0(0.5f - (x0x * x0x + x0y * x0y));
    float m1 = falloff_max0(0.5f - (ax * ax + ay * ay));
    float m2 = falloff_max0(0.5f - (bx * bx + by * by));
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;

    f3 c0, c1, c2;
    if (inDomain) {
        // glm::mod(i, 289) = i - 289 floor(i / 289) with an IEEE division.  For an integer |i| < 2^24 - 512 that is the integer remainder r in
        // [0, 288] (the correctly rounded quotient has the true quotient's floor, 289 * floor and the subtraction are exact;
        // test_device_simplex_lattice_hash_shortcuts).  The remainder without a division or an integer multiply (quarter rate): for an integer
        // |i| <= 2^21, t = i + 0.5 is exact and lies at least 0.5 from every multiple of 289, i.e. t / 289 at least 0.5 / 289 = 1.73e-3 from
        // every integer, while fl(t * fl(1 / 289)) is within |t / 289| * 2^-23 <= 7257 * 1.2e-7 = 8.7e-4 of t / 289: its floor is
        // floor(t / 289) = floor(i / 289), and one fused multiply-add returns i - 289 floor(i / 289) exactly (a small integer).
        const float qx = __builtin_floorf((ix + 0.5f) * (1.f / 289.f)), qy = __builtin_floorf((iy + 0.5f) * (1.f / 289.f));
        const int xi = (int)__builtin_fmaf(-289.f, qx, ix), yi = (int)__builtin_fmaf(-289.f, qy, iy);
        const int x16 = 16 * xi, y16 = 16 * yi;
        const int py0 = perm16_of2(y16), py1 = perm16_of2(y16 + 16);  // i1.y is 0 or 1: the middle corner re-uses one of the two
        const f4v t0 = grad2_at16((py0 + x16) + 16);
        const f4v t1 = grad2_at16(gt ? (py0 + x16) + 32 : (py1 + x16) + 16);
        const f4v t2 = grad2_at16((py1 + x16) + 32);
        c0 = mk3(t0.x, t0.y, t0.z); c1 = mk3(t1.x, t1.y, t1.z); c2 = mk3(t2.x, t2.y, t2.z);
    } else {
        ix = gmod(ix, 289.f);
        iy = gmod(iy, 289.f);
        c0 = simplex2_corner(permute((permute(iy + 0.f) + ix) + 0.f));
        c1 = simplex2_corner(permute((permute(iy + i1y) + ix) + i1x));
        c2 = simplex2_corner(permute((permute(iy + 1.f) + ix) + 1.f));
    }
    m0 = m0 * c0.z;
    m1 = m1 * c1.z;
    m2 = m2 * c2.z;
    const float g0 = c0.x * x0x + c0.y * x0y;
    const float g1 = c1.x * ax + c1.y * ay;
    const float g2 = c2.x * bx + c2.y * by;
    return 130.f * ((m0 * g0 + m1 * g1) + m2 * g2);
}

// ---------------------------------------------------------------------------------------------------------
// Bounds the exact prunings rely on (cave_biome's depth zones, the cave threshold, the rasterisers' extents).  Adversarially aligned
// gradients give sup |simplex2| <= 1.0348 and sup |simplex3| <= 1.2259 over a lattice cell (tests/test_oracle_math.py::test_simplex_bounds,
// test_simplex3_bound).  In floating point the skew sum that picks the cell is rounded, so the evaluated point can lie outside its cell by
// a few ulp of the ARGUMENT: the prunings are therefore only applied where the arguments are small - columns within MM_PRUNE_DOMAIN
// blocks of the origin (a 4 096 x 4 096-chunk world; the largest simplex2 argument there is 2.6e4, the largest simplex3 argument 1e4, the
// point at most 0.006 outside its cell, which can add at most 0.006 * 3 * 0.0625 * 130 * 0.79 = 0.115 resp. 0.006 * 4 * 0.1296 * 42 =
// 0.13) - and with the slack folded into the constants.  Beyond the domain every voxel takes the unpruned path: same values, more work.
// ---------------------------------------------------------------------------------------------------------
#define MM_SIMPLEX2_BOUND 1.16f
#define MM_SIMPLEX3_BOUND 1.37f
#define MM_PRUNE_DOMAIN 32768
MM_DEV bool prune_domain(int wx, int wz) { return wx > -MM_PRUNE_DOMAIN && wx < MM_PRUNE_DOMAIN && wz > -MM_PRUNE_DOMAIN && wz < MM_PRUNE_DOMAIN; }

// ---------------------------------------------------------------------------------------------------------
// simplex3 split at the lattice: part1 (skewed cell + offsets inside it) -> gradients of the cell's 4 simplex corners
// -> part3 (falloff * gradient . offset).  The gradients are a function of (cell, corner ordering) ONLY and come from the
// LDS tables.  simplex3 = part3(part1(v), gradients(part1(v))): glm's operations in glm's order, merely regrouped (checked by
// the probe tests against real glm).
// ---------------------------------------------------------------------------------------------------------
// The corner ordering: g = step(x0.yzx, x0), l = 1 - g, i1 = min(g, l.zxy), i2 = max(g, l.zxy) are all in {0, 1}, so min / max are AND /
// OR of the three order predicates and their complements: i1x = gx AND NOT gz, i2x = gx OR NOT gz, ...  The three comparisons are taken
// as wave masks (ballot: NOT g, i.e. "x0.x < x0.y" - false for NaN exactly like step), the six combinations are s_andn2 / s_orn2 on the
// scalar unit for the whole wave, and each use is one v_cndmask on that mask (inverse ballot): 3 VALU compares per simplex3.  Written
// with bools the compiler emits both polarities of every comparison (6 compares), and `a && !b` becomes a nested per-lane select.
struct Sx3Cell {
    float ix, iy, iz; float x0x, x0y, x0z;
    unsigned long long lx, ly, lz;          // wave masks of NOT gx, NOT gy, NOT gz
    MM_DEV bool i1x() const { return __builtin_amdgcn_inverse_ballot_w64(~lx & lz); }
    MM_DEV bool i1y() const { return __builtin_amdgcn_inverse_ballot_w64(~ly & lx); }
    MM_DEV bool i1z() const { return __builtin_amdgcn_inverse_ballot_w64(~lz & ly); }
    MM_DEV bool i2x() const { return __builtin_amdgcn_inverse_ballot_w64(~lx | lz); }
    MM_DEV bool i2y() const { return __builtin_amdgcn_inverse_ballot_w64(~ly | lx); }
    MM_DEV bool i2z() const { return __builtin_amdgcn_inverse_ballot_w64(~lz | ly); }
};

MM_DEV float sel_f(bool m, float ifSet, float ifClear) { return m ? ifSet : ifClear; }
MM_DEV int sel_i(bool m, int ifSet, int ifClear) { return m ? ifSet : ifClear; }
MM_DEV int sel_16_0(bool m) { return m ? 16 : 0; }

MM_DEV Sx3Cell simplex3_part1(float vx, float vy, float vz)
{
    const float Cx = (float)(1.0 / 6.0), Cy = (float)(1.0 / 3.0);
    Sx3Cell c;
    const float d = (vx * Cy + vy * Cy) + vz * Cy;
    c.ix = __builtin_floorf(vx + d); c.iy = __builtin_floorf(vy + d); c.iz = __builtin_floorf(vz + d);
    const float e = (c.ix * Cx + c.iy * Cx) + c.iz * Cx;
    c.x0x = (vx - c.ix) + e; c.x0y = (vy - c.iy) + e; c.x0z = (vz - c.iz) + e;
    // g = step(x0.yzx, x0)
    c.lx = __builtin_amdgcn_ballot_w64(c.x0x < c.x0y); c.ly = __builtin_amdgcn_ballot_w64(c.x0y < c.x0z); c.lz = __builtin_amdgcn_ballot_w64(c.x0z < c.x0x);
    return c;
}

// 12 floats: (qx, qy, qz) of the 4 corners, already scaled by taylorInvSqrt.  Arithmetic form; ix, iy, iz already mod289'd.
MM_DEV void simplex3_gradients_direct(float ix, float iy, float iz, const Sx3Cell& c, float* __restrict__ q)
{
    const float oz[4] = {0.f, sel_f(c.i1z(), 1.f, 0.f), sel_f(c.i2z(), 1.f, 0.f), 1.f}, oy[4] = {0.f, sel_f(c.i1y(), 1.f, 0.f), sel_f(c.i2y(), 1.f, 0.f), 1.f},
                ox[4] = {0.f, sel_f(c.i1x(), 1.f, 0.f), sel_f(c.i2x(), 1.f, 0.f), 1.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float a = permute(iz + oz[k]);
        const float b = permute((a + iy) + oy[k]);
        const f3 g = simplex3_corner(permute((b + ix) + ox[k]));
        q[3 * k] = g.x; q[3 * k + 1] = g.y; q[3 * k + 2] = g.z;
    }
}

// table domain of the gradients: lattice coordinates below 2^23 in magnitude (then mod289 is an integer in [-1, 289] and 289 * floor(i /
// 289) is exact, see below); NaN fails the test
MM_DEV bool simplex3_in_domain(float ix, float iy, float iz) { return __builtin_fabsf(ix) < 8388608.f && __builtin_fabsf(iy) < 8388608.f && __builtin_fabsf(iz) < 8388608.f; }

// Same values through the LDS tables (see the header); falls back to the arithmetic outside the tables' domain.  KNOWN_IN = the caller
// has established the domain for this evaluation (the fbm stacks test their last octave once instead of every octave).
template <bool KNOWN_IN = false>
MM_DEV void simplex3_gradients(const Sx3Cell& c, float* __restrict__ q)
{
    float ix = c.ix, iy = c.iy, iz = c.iz;
    if (KNOWN_IN || simplex3_in_domain(ix, iy, iz)) {
        // mod289(i) = i - floor(i * (1 / 289)) * 289: for |i| < 2^23 the product floor * 289 is an integer below 2^24, hence exact, and so is the
        // difference: one fused multiply-add returns the same value as the multiply and the subtraction
        ix = __builtin_fmaf(-289.f, __builtin_floorf(ix * (1.f / 289.f)), ix);
        iy = __builtin_fmaf(-289.f, __builtin_floorf(iy * (1.f / 289.f)), iy);
        iz = __builtin_fmaf(-289.f, __builtin_floorf(iz * (1.f / 289.f)), iz);
        const int x16 = 16 * (int)ix, y16 = 16 * (int)iy, z16 = 16 * (int)iz;
        // level z: the four corners only ever need permute(z) and permute(z + 1), as byte offsets of the next level's entries
        const int pz0 = perm16u(z16), pz1 = perm16u(z16 + 16);
        const int a[4] = {pz0, sel_i(c.i1z(), pz1, pz0), sel_i(c.i2z(), pz1, pz0), pz1};
        const int oy[4] = {0, sel_16_0(c.i1y()), sel_16_0(c.i2y()), 16}, ox[4] = {0, sel_16_0(c.i1x()), sel_16_0(c.i2x()), 16};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int b = perm16u((a[k] + y16) + oy[k]);
            const f4v g = grad3u((b + x16) + ox[k]);           // fused: gradient of permute(b + x + o)
            q[3 * k] = g.x; q[3 * k + 1] = g.y; q[3 * k + 2] = g.z;
        }
    } else {
        ix = mod289(ix); iy = mod289(iy); iz = mod289(iz);
        simplex3_gradients_direct(ix, iy, iz, c, q);
    }
}

MM_DEV float simplex3_part3(const Sx3Cell& c, const float* __restrict__ q)
{
    const float Cx = (float)(1.0 / 6.0), Cy = (float)(1.0 / 3.0);
    // x0 - i1 with i1 in {0, 1}: x0 - 0 is x0 itself (exact, sign of zero included), so both middle corners select between x0 and the ONE
    // difference x0 - 1 per axis
    const float dx = c.x0x - 1.f, dy = c.x0y - 1.f, dz = c.x0z - 1.f;
    const float cx[4] = {c.x0x, sel_f(c.i1x(), dx, c.x0x) + Cx, sel_f(c.i2x(), dx, c.x0x) + Cy, c.x0x - 0.5f};
    const float cy[4] = {c.x0y, sel_f(c.i1y(), dy, c.x0y) + Cx, sel_f(c.i2y(), dy, c.x0y) + Cy, c.x0y - 0.5f};
    const float cz[4] = {c.x0z, sel_f(c.i1z(), dz, c.x0z) + Cx, sel_f(c.i2z(), dz, c.x0z) + Cy, c.x0z - 0.5f};
    float mm4[4], pd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float m = falloff_max0(0.6f - ((cx[k] * cx[k] + cy[k] * cy[k]) + cz[k] * cz[k]));      // see simplex2_inl
        m = m * m;
        mm4[k] = m * m;
        pd[k] = (q[3 * k] * cx[k] + q[3 * k + 1] * cy[k]) + q[3 * k + 2] * cz[k];
    }
    return 42.f * ((mm4[0] * pd[0] + mm4[1] * pd[1]) + (mm4[2] * pd[2] + mm4[3] * pd[3]));
}

template <bool KNOWN_IN = false>
MM_DEV float simplex3_inl(float vx, float vy, float vz)
{
    const Sx3Cell c = simplex3_part1(vx, vy, vz);
    float q[12];
    simplex3_gradients<KNOWN_IN>(c, q);
    return simplex3_part3(c, q);
}

// One shared (non-inlined) body per translation unit for the scattered call sites (block rules, rasterisers); the fbm stacks
// below inline the body ONCE inside a rolled octave loop instead: no call, no callee-saved register pressure around it.
MM_SIMPLEX_ATTR float simplex2(float vx, float vy) { return simplex2_inl<false>(vx, vy); }
MM_SIMPLEX_ATTR float simplex3(float vx, float vy, float vz) { return simplex3_inl<false>(vx, vy, vz); }

// ---------------------------------------------------------------------------------------------------------
// fbm stacks (rng.hpp:166-191): amplitude halves, frequency doubles, octaves summed in order
// ---------------------------------------------------------------------------------------------------------
template <int OCT, bool KNOWN_IN>
MM_DEV float fbm2_loop(float x, float y)
{
    float acc = 0.f, amp = 1.f;
#pragma unroll 1
    for (int i = 0; i < OCT; ++i) {
        amp *= 0.5f;
        acc += amp * simplex2_inl<KNOWN_IN>(x, y);
        x *= 2.f; y *= 2.f;
    }
    return acc;
}

// domain test once per stack (see fbm3 below): |floor(v + (vx + vy) * 0.366)| < 2 max|v| + 1, the last octave evaluates at 2^(OCT - 1) v,
// so max|v| < 2^(20 - OCT) keeps every lattice coordinate below 2^20 + 1 (simplex2_inl's table domain is 2^21)
// HOIST is opt-in: it pays in the cave-biome bands (k_fill -1.2 %), and it costs k_heightfield 50 % of its time (its two dozen inlined
// stacks sit in one divergent switch; measured, profiles/README.md r03), so the surface functions keep the per-octave test
template <int OCT, bool HOIST = false>
MM_DEV float fbm2(float x, float y)
{
    if (!HOIST) return fbm2_loop<OCT, false>(x, y);
    const float m = __builtin_fmaxf(__builtin_fabsf(x), __builtin_fabsf(y));
    if (__builtin_expect(m < (float)(1 << (20 - OCT)), 1)) return fbm2_loop<OCT, true>(x, y);
    // beyond the table domain (block coordinates of hundreds of millions): the shared out-of-line simplex2 with its per-call test, so that
    // the call sites carry ONE inlined loop body (k_heightfield inlines dozens of stacks: a second body each cost it 50 % of its time)
    float acc = 0.f, amp = 1.f;
#pragma unroll 1
    for (int i = 0; i < OCT; ++i) {
        amp *= 0.5f;
        acc += amp * simplex2(x, y);
        x *= 2.f; y *= 2.f;
    }
    return acc;
}

template <int OCT, bool KNOWN_IN>
MM_DEV float fbm3_loop(float x, float y, float z)
{
    float acc = 0.f, amp = 1.f;
#pragma unroll 1
    for (int i = 0; i < OCT; ++i) {
        amp *= 0.5f;
        acc += amp * simplex3_inl<KNOWN_IN>(x, y, z);
        x *= 2.f; y *= 2.f; z *= 2.f;
    }
    return acc;
}

// The table-domain test of the gradients (|lattice coordinate| < 2^23) once per stack instead of once per octave: a lattice coordinate is
// floor(v + (vx + vy + vz) / 3), at most 2 max|v| + 1 in magnitude, and the last octave evaluates at 2^(OCT - 1) times the argument - so
// max|v| < 2^(22 - OCT) puts every octave inside the domain (NaN fails the test and takes the per-octave check).  Same values either way.
template <int OCT>
MM_DEV float fbm3(float x, float y, float z)
{
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(x), __builtin_fabsf(y)), __builtin_fabsf(z));
    if (__builtin_expect(m < (float)(1 << (22 - OCT)), 1)) return fbm3_loop<OCT, true>(x, y, z);
    float acc = 0.f, amp = 1.f;                 // beyond the domain: the shared out-of-line simplex3 (see fbm2)
#pragma unroll 1
    for (int i = 0; i < OCT; ++i) {
        amp *= 0.5f;
        acc += amp * simplex3(x, y, z);
        x *= 2.f; y *= 2.f; z *= 2.f;
    }
    return acc;
}

template <int OCT>
MM_DEV f2 fbm2from2(float x, float y) { return mk2(fbm2<OCT>(x, y), fbm2<OCT>(x + 5923.45f, y + 4129.42f)); }

template <int OCT>
MM_DEV f3 fbm3from3(float x, float y, float z)
{
    // rolled over the three components as well: one inlined simplex body for all 3 * OCT samples
    float r[3];
#pragma unroll 1
    for (int k = 0; k < 3; ++k) {
        const float ox = k == 0 ? 0.f : (k == 1 ? 5923.45f : 1765.68f), oy = k == 0 ? 0.f : (k == 1 ? 4129.42f : 4704.36f),
                    oz = k == 0 ? 0.f : (k == 1 ? 5790.48f : 5692.12f);
        r[k] = k == 0 ? fbm3<OCT>(x, y, z) : fbm3<OCT>(x + ox, y + oy, z + oz);
    }
    return mk3(r[0], r[1], r[2]);
}

MM_DEV f2 simplex2from2(float x, float y) { return mk2(simplex2(x, y), simplex2(x + 5923.45f, y + 4129.42f)); }

// ---------------------------------------------------------------------------------------------------------
// sin hashes (rng.hpp:102-155)
// ---------------------------------------------------------------------------------------------------------
MM_SIMPLEX_ATTR float hash_unit(float t) { return fract(sinf_(t) * 39021.426f); }
MM_DEV float rand1from2(float x, float y) { return hash_unit(x * 238.68f + y * 491.28f); }
MM_DEV float rand1from3(float x, float y, float z) { return hash_unit((x * 238.68f + y * 491.28f) + z * 640.88f); }
MM_DEV f2 rand2from2(float x, float y) { return mk2(hash_unit(x * 238.68f + y * 491.28f), hash_unit(x * 654.37f + y * 560.45f)); }
MM_DEV f2 rand2from3(float x, float y, float z)
{
    return mk2(hash_unit((x * 238.68f + y * 491.28f) + z * 640.88f), hash_unit((x * 654.37f + y * 560.45f) + z * 151.81f));
}
MM_DEV f3 rand3from2(float x, float y)
{
    return mk3(hash_unit(x * 238.68f + y * 491.28f), hash_unit(x * 654.37f + y * 560.45f), hash_unit(x * 640.88f + y * 151.81f));
}
MM_DEV f3 rand3from3(float x, float y, float z)
{
    return mk3(hash_unit((x * 238.68f + y * 491.28f) + z * 402.98f), hash_unit((x * 654.37f + y * 560.45f) + z * 747.42f),
               hash_unit((x * 640.88f + y * 151.81f) + z * 674.81f));
}

// ---------------------------------------------------------------------------------------------------------
// Worley (rng.hpp:193-320).  Neighbour iteration order x, y, (z) from -1 to 1 is part of the contract (ties).
// ---------------------------------------------------------------------------------------------------------
struct Worley2 { float d1, d2; f2 closest; };

MM_DEV Worley2 worley2(float px, float py)
{
    const float flx = __builtin_floorf(px), fly = __builtin_floorf(py);
    const int ux = (int)flx, uy = (int)fly;
    const float fx = px - flx, fy = py - fly;
    Worley2 w;
    w.d1 = 3.402823466e+38f; w.d2 = 3.402823466e+38f; w.closest = mk2(0.f, 0.f);
    for (int x = -1; x <= 1; ++x) {
        for (int y = -1; y <= 1; ++y) {
            const f2 pt = rand2from2((float)(ux + x), (float)(uy + y));
            const float dx = ((float)x + pt.x) - fx, dy = ((float)y + pt.y) - fy;
            const float dist = __builtin_sqrtf(dx * dx + dy * dy);
            if (dist < w.d1) { w.d2 = w.d1; w.d1 = dist; w.closest = pt; }
            else if (dist < w.d2) { w.d2 = dist; }
        }
    }
    return w;
}

// Direct (no table) provider of 3D cell points.
struct CellDirect {
    static constexpr int kBoxStrideX = 0, kBoxStrideY = 0;
    MM_DEV const float* box27(int, int, int) const { return nullptr; }     // nothing staged: special_cave_noise takes its generic loop
    MM_DEV f3 operator()(int cx, int cy, int cz) const { return rand3from3((float)cx, (float)cy, (float)cz); }
    MM_DEV void row3(int cx, int cy, int cz, f3 (&out)[3]) const
    {
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k] = (*this)(cx, cy, cz - 1 + k);
    }
};

struct Worley3 { float d1, d2, d3; f3 closest; };

template <class Cells>
MM_DEV Worley3 worley3(float px, float py, float pz, const Cells& cells)
{
    const float flx = __builtin_floorf(px), fly = __builtin_floorf(py), flz = __builtin_floorf(pz);
    const int ux = (int)flx, uy = (int)fly, uz = (int)flz;
    const float fx = px - flx, fy = py - fly, fz = pz - flz;
    Worley3 w;
    w.d1 = 3.402823466e+38f; w.d2 = 3.402823466e+38f; w.d3 = 3.402823466e+38f; w.closest = mk3(0.f, 0.f, 0.f);
    for (int x = -1; x <= 1; ++x) {
        for (int y = -1; y <= 1; ++y) {
            for (int z = -1; z <= 1; ++z) {
                const f3 pt = cells(ux + x, uy + y, uz + z);
                const float dx = ((float)x + pt.x) - fx, dy = ((float)y + pt.y) - fy, dz = ((float)z + pt.z) - fz;
                const float dist = __builtin_sqrtf((dx * dx + dy * dy) + dz * dz);
                // three-smallest tracking (specialCaveNoise, rng.hpp:300-314); d1/d2/closest coincide with worley(vec3)
                if (dist < w.d1) { w.d3 = w.d2; w.d2 = w.d1; w.d1 = dist; w.closest = pt; }
                else if (dist < w.d2) { w.d3 = w.d2; w.d2 = dist; }
                else if (dist < w.d3) { w.d3 = dist; }
            }
        }
    }
    return w;
}

// specialCaveNoise (rng.hpp:300-320) only uses the VALUES of the first and third smallest distance.  sqrt is monotonic and
// correctly rounded, so the three smallest sqrt(d2) are the sqrt of the three smallest d2 whatever the tie order: the search
// runs on squared distances and takes 2 square roots instead of 27.
// Fast path of the search when the 27 cells are staged in LDS (Cells::box27): same values, fewer instructions.
//  * The three smallest of a multiset do not depend on the order the cells are visited in, and the update of (s1 <= s2 <= s3) by a
//    value u is s3' = med3(s2, s3, u), s2' = med3(s1, s2, u), s1' = min(s1, u).  Squared distances are finite, >= +0 and never NaN for
//    a finite position, so their order as floats is their order as unsigned integers: one v_med3_u32 each, no canonicalisation.
//  * (float)0 + pt.x is pt.x (a cell point is never -0), so the centre slabs skip that addition.
//  * Exact skip of whole (x, y) columns of cells.  A cell point lies in [0, 1]^3 of its cell, so in the cells x = -1 the computed
//    dx = fl(fl(-1 + pt.x) - fx) <= -fx and in x = +1 dx >= fl(1 - fx) >= 0 (rounding is monotone), hence fl(dx dx) >= fl(fx fx) resp.
//    fl(gx gx) with gx = fl(1 - fx); likewise y; and d2 = fl(fl(dx dx + dy dy) + dz dz) >= fl(bx + by) with those bounds (0 in the centre
//    slab).  A cell with d2 >= s3 leaves (s1, s2, s3) as they are, so a column whose bound is >= s3 in every lane of the wave is not
//    evaluated (a NaN bound never skips).  Centre column first, then the four face columns, then the four corner columns: s3 is small by
//    the time the far ones are tested.  The lanes of a wave are neighbours in space, so they mostly agree.
#ifndef MM_WORLEY_FAST
#define MM_WORLEY_FAST 1
#endif
MM_DEV unsigned med3_u32(unsigned a, unsigned b, unsigned c)
{
    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    const unsigned m = lo > c ? lo : c;          // max(min(a, b), c)
    return hi < m ? hi : m;                      // min(max(a, b), max(min(a, b), c))
}

template <class Cells>
MM_DEV float special_cave_noise(float px, float py, float pz, const Cells& cells)
{
    const float flx = __builtin_floorf(px), fly = __builtin_floorf(py), flz = __builtin_floorf(pz);
    const int ux = (int)flx, uy = (int)fly, uz = (int)flz;
    const float fx = px - flx, fy = py - fly, fz = pz - flz;
#if MM_WORLEY_FAST
    if (const float* box = cells.box27(ux, uy, uz)) {          // the point of cell (ux - 1, uy - 1, uz - 1); x stride Cells::kBoxStrideX, y stride kBoxStrideY, z stride 3
        unsigned u1 = 0x7f7fffffu, u2 = 0x7f7fffffu, u3 = 0x7f7fffffu;
        auto column = [&](int x, int y) {
            const float* p = box + (x + 1) * Cells::kBoxStrideX + (y + 1) * Cells::kBoxStrideY;
#pragma unroll
            for (int z = -1; z <= 1; ++z) {
                const float qx = p[3 * (z + 1)], qy = p[3 * (z + 1) + 1], qz = p[3 * (z + 1) + 2];
                const float dx = (x == 0 ? qx : (float)x + qx) - fx, dy = (y == 0 ? qy : (float)y + qy) - fy, dz = (z == 0 ? qz : (float)z + qz) - fz;
                const unsigned u = __float_as_uint((dx * dx + dy * dy) + dz * dz);
                u3 = med3_u32(u2, u3, u); u2 = med3_u32(u1, u2, u); u1 = u1 < u ? u1 : u;
            }
        };
        const float gx = 1.f - fx, gy = 1.f - fy;
        const float bxm = fx * fx, bxp = gx * gx, bym = fy * fy, byp = gy * gy;
        column(0, 0);
        if (!(bxm >= __uint_as_float(u3))) column(-1, 0);
        if (!(bxp >= __uint_as_float(u3))) column(1, 0);
        if (!(bym >= __uint_as_float(u3))) column(0, -1);
        if (!(byp >= __uint_as_float(u3))) column(0, 1);
        if (!((bxm + bym) >= __uint_as_float(u3))) column(-1, -1);
        if (!((bxm + byp) >= __uint_as_float(u3))) column(-1, 1);
        if (!((bxp + bym) >= __uint_as_float(u3))) column(1, -1);
        if (!((bxp + byp) >= __uint_as_float(u3))) column(1, 1);
        return __builtin_sqrtf(__uint_as_float(u3)) / __builtin_sqrtf(__uint_as_float(u1)) - 1.f;
    }
#endif
    float s1 = 3.402823466e+38f, s2 = 3.402823466e+38f, s3 = 3.402823466e+38f;
    // 9 rolled (x, y) steps of 3 unrolled z cells: a fully unrolled search hoists all 81 cell-point loads and spills
#pragma unroll 1
    for (int xy = 0; xy < 9; ++xy) {
        const int x = xy / 3 - 1, y = xy % 3 - 1;
        {
            f3 row[3];
            cells.row3(ux + x, uy + y, uz, row);
#pragma unroll
            for (int z = -1; z <= 1; ++z) {
                const f3 pt = row[z + 1];
                const float dx = ((float)x + pt.x) - fx, dy = ((float)y + pt.y) - fy, dz = ((float)z + pt.z) - fz;
                const float d2 = (dx * dx + dy * dy) + dz * dz;
                // keep the three smallest: s1 <= s2 <= s3.  Squared distances are sums of squares: never NaN for a finite position, never -0,
                // so the hardware min / max (one instruction each) return what glm's compare-and-select would
                const float a = __builtin_fminf(s1, d2), b = __builtin_fmaxf(s1, d2);
                const float c = __builtin_fminf(s2, b), d = __builtin_fmaxf(s2, b);
                s1 = a; s2 = c; s3 = __builtin_fminf(s3, d);
            }
        }
    }
    return __builtin_sqrtf(s3) / __builtin_sqrtf(s1) - 1.f;
}

}  // namespace mm
