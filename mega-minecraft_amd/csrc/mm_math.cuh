// mmgen device math: deterministic transcendental functions + integer hash / minstd RNG for gfx950.
//
// Every function here is a fixed sequence of IEEE-754 operations; the translation unit must be compiled with
// -ffp-contract=off (no implicit FMA) and without fast-math.  The only fused operations are the explicit
// __builtin_fmaf / __builtin_fma calls below.  The contract (constants, operation order) is documented in
// DESIGN.md "Math contract"; it replaces CUDA libm's sinf/cosf/powf/atan2f/acosf used by the reference
// (src/util/rng.hpp:102-155, src/terrain/biomeFuncs.hpp:235,311,375, src/terrain/featurePlacement.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MM_DEV __device__ __forceinline__

namespace mm {

// ---------------------------------------------------------------------------------------------------------
// sin / cos: fp64 two-term Cody-Waite reduction by pi/2 (valid to |x| ~ 2^40), fp32 minimax polynomials.
// ---------------------------------------------------------------------------------------------------------
struct Reduced { float r; int q; };

MM_DEV Reduced reduce_pio2(float x)
{
    const double xd = (double)x;
    const double k = __builtin_rint(xd * 0.63661977236758138243);
    double rd = __builtin_fma(-k, 1.57079632679489655800e+00, xd);
    rd = __builtin_fma(-k, 6.12323399573676603587e-17, rd);
    const double kq = k - 4.0 * __builtin_floor(k * 0.25);
    Reduced out;
    out.r = (float)rd;
    out.q = (int)kq;
    return out;
}

MM_DEV float sin_poly(float r)
{
    const float z = r * r;
    float p = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    p = __builtin_fmaf(z, p, -1.6666654611e-1f);
    return __builtin_fmaf(r * z, p, r);
}

MM_DEV float cos_poly(float r)
{
    const float z = r * r;
    float p = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    p = __builtin_fmaf(z, p, 4.166664568298827e-2f);
    return __builtin_fmaf(z * z, p, __builtin_fmaf(-0.5f, z, 1.0f));
}

MM_DEV float sinf_(float x)
{
    const Reduced a = reduce_pio2(x);
    const float v = (a.q & 1) ? cos_poly(a.r) : sin_poly(a.r);
    return (a.q & 2) ? -v : v;
}

MM_DEV float cosf_(float x)
{
    const Reduced a = reduce_pio2(x);
    const float v = (a.q & 1) ? sin_poly(a.r) : cos_poly(a.r);
    return ((a.q + 1) & 2) ? -v : v;
}

// ---------------------------------------------------------------------------------------------------------
// powf via fp64 log/exp (fdlibm polynomial sets), atan2f / acosf via fp64 atan.
// ---------------------------------------------------------------------------------------------------------
MM_DEV double log_(double x)
{
    uint64_t bits = (uint64_t)__double_as_longlong(x);
    int k = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)bits);
    if (m > 1.41421356237309514547) { m = m * 0.5; k += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
    const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

MM_DEV double exp_(double x)
{
    const double k = __builtin_rint(x * 1.44269504088896338700e+00);
    const double hi = x - k * 6.93147180369123816490e-01;
    const double lo = k * 1.90821492927058770002e-10;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (1.66666666666666019037e-01 + t * (-2.77777777770155933842e-03 + t * (6.61375632143793436117e-05
                     + t * (-1.65339022054652515390e-06 + t * 4.13813679705723846039e-08))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    uint64_t bits = (uint64_t)__double_as_longlong(y);
    bits += (uint64_t)((int64_t)k) << 52;
    return __longlong_as_double((long long)bits);
}

// one shared body per translation unit: inlined at every call site its two dozen fp64 constants are hoisted to the top of the kernel,
// where they are the registers the rasterisers run out of
static __device__ __attribute__((noinline)) float powf_general(float x, float y) { return (float)exp_((double)y * log_((double)x)); }

MM_DEV float powf_(float x, float y)
{
    if (y == 2.f) return x * x;
    if (x == 0.f) return (y > 0.f) ? 0.f : 1.f;
    if (x == 1.f || y == 0.f) return 1.f;
    if (x < 0.f) return __builtin_nanf("");
    return powf_general(x, y);
}

MM_DEV double atan_(double x)
{
    const bool neg = x < 0.0;
    double ax = neg ? -x : x;
    if (ax >= 7.3786976294838206e+19) return neg ? -1.57079632679489655800e+00 : 1.57079632679489655800e+00;
    double hi = 0.0, lo = 0.0;
    bool direct = false;
    if (ax < 0.4375) {
        direct = true;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) { hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; ax = (2.0 * ax - 1.0) / (2.0 + ax); }
        else             { hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; ax = (ax - 1.0) / (ax + 1.0); }
    } else {
        if (ax < 2.4375) { hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; ax = (ax - 1.5) / (1.0 + 1.5 * ax); }
        else             { hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; ax = -1.0 / ax; }
    }
    const double z = ax * ax;
    const double w = z * z;
    const double s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02
                      + w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02
                      + w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
    double res;
    if (direct) res = ax - ax * (s1 + s2);
    else        res = hi - ((ax * (s1 + s2) - lo) - ax);
    return neg ? -res : res;
}

MM_DEV float atan2f_(float y, float x)
{
    const double PI_D = 3.14159265358979311600e+00;
    const double yd = (double)y, xd = (double)x;
    if (xd == 0.0 && yd == 0.0) return 0.f;
    if (xd == 0.0) return (float)(yd > 0.0 ? 0.5 * PI_D : -0.5 * PI_D);
    const double a = atan_(yd / xd);
    if (xd > 0.0) return (float)a;
    return (float)(yd >= 0.0 ? a + PI_D : a - PI_D);
}

MM_DEV float acosf_(float x)
{
    const double xd = (double)x;
    const double s = __builtin_sqrt((1.0 - xd) * (1.0 + xd));
    if (xd == 0.0) return (float)1.57079632679489655800e+00;
    const double a = atan_(s / xd);
    return (float)(xd > 0.0 ? a : a + 3.14159265358979311600e+00);
}

MM_DEV float fmodf_(float x, float y) { return __builtin_fmodf(x, y); }   // exact in IEEE arithmetic

// ---------------------------------------------------------------------------------------------------------
// glm scalar helpers (external/include/glm/detail/func_common.inl): evaluation order is part of the contract.
// ---------------------------------------------------------------------------------------------------------
MM_DEV float gmin(float a, float b) { return (b < a) ? b : a; }
MM_DEV float gmax(float a, float b) { return (a < b) ? b : a; }
MM_DEV int imin(int a, int b) { return (b < a) ? b : a; }
MM_DEV int imax(int a, int b) { return (a < b) ? b : a; }
MM_DEV float clampf(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
MM_DEV float fract(float x) { return x - __builtin_floorf(x); }
MM_DEV float gmod(float a, float b) { return a - b * __builtin_floorf(a / b); }
MM_DEV float mixf(float x, float y, float a) { return x * (1.f - a) + y * a; }
// glm::clamp = min(max(x, 0), 1) with compare-and-select; the hardware max / min differ from it only for x = -0 (they return +0) and
// for NaN.  t = -0 and t = +0 give the same product t * t * (3 - 2 t) = +0, and no call site can produce a NaN (finite positions,
// e1 != e0 everywhere: the edges are constants or differ by 5 - 3 |f|, 10 - 7 |f| with |f| < 1.02): two instructions instead of four
MM_DEV float smoothstep(float e0, float e1, float x)
{
    const float t = __builtin_fminf(__builtin_fmaxf((x - e0) / (e1 - e0), 0.f), 1.f);
    return t * t * (3.f - 2.f * t);
}

// ---------------------------------------------------------------------------------------------------------
// integer hash + thrust::minstd_rand + uniform_real_distribution<float>(0,1)   (src/util/rng.hpp:69-96)
// ---------------------------------------------------------------------------------------------------------
MM_DEV uint32_t hash32(uint32_t a)
{
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

struct MinStd {
    uint32_t x;
    // m = 2^31 - 1 is a Mersenne prime: 2^31 = 1 (mod m), so a residue is "low 31 bits + the rest", folded once or twice - the
    // same value as the % of thrust's static_mod, without a division
    MM_DEV void seed(uint32_t s)
    {
        s = (s >= 4294967294u) ? s - 4294967294u : (s >= 2147483647u ? s - 2147483647u : s);
        x = s ? s : 1u;
    }
    MM_DEV uint32_t next()
    {
        const uint64_t p = (uint64_t)x * 48271ull;                       // < 2^47: exact
        uint32_t r = (uint32_t)(p & 0x7fffffffull) + (uint32_t)(p >> 31);  // < 2^31 + 2^16
        x = (r >= 2147483647u) ? r - 2147483647u : r;
        return x;
    }
    MM_DEV float u01() { return (float)(next() - 1u) / 2147483648.f; }
};

MM_DEV MinStd rng3(int x, int y, int z)
{
    MinStd r;
    r.seed(hash32(0x80000000u | ((uint32_t)x << 22) | (uint32_t)y) ^ hash32((uint32_t)z));
    return r;
}
MM_DEV MinStd rng4(int x, int y, int z, int w)
{
    MinStd r;
    r.seed(hash32(0x80000000u | ((uint32_t)x << 22) | ((uint32_t)y << 11) | (uint32_t)w) ^ hash32((uint32_t)z));
    return r;
}

// Work counters of the persistent kernels (k_fill_cave, k_cave_biomes, k_apply_features): counter p - 64 B apart, `work[16 * p]` - hands out
// the items p, p + N, p + 2 N, ...; a wave draws from its own and moves on when that has run dry.  It used to try the others one
// read-modify-write at a time, so at the end of a launch EVERY wave touched EVERY counter: waves x N atomics, serialised per counter at
// ~11 ns in L2 - 68 us of tail for the cave fill's 6 144 waves (1 % of it in bulk, 40 % of it in a 35-chunk streaming tick), 45 us for
// the rasterisers' 4 096.  Now a wave whose counter is dry looks at all N with ONE round trip of plain loads (lane p reads counter p)
// and goes straight to the next one that still has items - or learns that none has.  (A counter seen live can be dry by the time the
// draw lands: the caller's loop then simply asks again; counters only grow, so it ends.)  Wave-uniform call.
#ifndef MM_COUNTER_PROBE_LOADS
#define MM_COUNTER_PROBE_LOADS 1      // 0: the old walk, one atomic per counter (A/B)
#endif
MM_DEV int next_live_counter(const unsigned* work, int nCounters, int after, int nItems)      // -1: every counter is dry
{
    const int lane = threadIdx.x & 63;
    unsigned v = 0xffffffffu;
    if (lane < nCounters) v = __hip_atomic_load(&work[16 * lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool live = lane < nCounters && (long long)v * nCounters + lane < (long long)nItems;
    const unsigned long long m = __ballot(live);
    if (!m) return -1;
    const unsigned long long later = m & ~((2ull << after) - 1ull);      // the first live one after `after`, cyclically
    return later ? (int)__builtin_ctzll(later) : (int)__builtin_ctzll(m);
}

// LDS hand-off inside one wave: its LDS operations execute in issue order, only the compiler must not reorder across the hand-off
MM_DEV void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


}  // namespace mm
