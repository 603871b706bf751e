// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Deterministic transcendental functions ("mm libm contract").
//
// The reference calls CUDA's device libm (sinf, cosf, sincosf, powf, acosf, atan2f, fmodf) and MSVC's
// host libm (sinf in isFeaturePos chunk.cu:999-1008, tanf in biomeFuncs.hpp:843-847).  Neither is
// reproducible off NVIDIA/Windows and the reference pins none of them with a test, so the build
// defines its own: every function below is a fixed sequence of IEEE-754 operations (+ - * / sqrt
// fma rint floor, int<->fp conversions) that gives the same bits under g++ -ffp-contract=off on x86
// and under hipcc -ffp-contract=off on gfx950.  The HIP product carries its own copy of the
// same contract (mega-minecraft_amd/csrc/mm_math.cuh); tests/test_math.py compares the two bit for bit
// and checks both against glibc to <= 2 ulp.
//
// parity unpinned at this boundary: results follow this contract, not CUDA's libm.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace mmo {

static inline float mm_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static inline double mm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// Argument reduction shared by sin/cos: x = k*(pi/2) + r, |r| <= pi/4 (+tiny), q = k mod 4.
// Done in fp64 with a two-term pi/2 so that |x| up to ~2^40 (the rand*From* hashes feed
// dot products up to ~1e11 into sin, rng.hpp:102-155) reduces to better than fp32 accuracy.
static inline void mm_reduce_pio2(float x, float* r, int* q)
{
    const double xd = (double)x;
    const double k = rint(xd * 0.63661977236758138243);              // 2/pi
    double rd = mm_fma(-k, 1.57079632679489655800e+00, xd);           // pi/2 hi
    rd = mm_fma(-k, 6.12323399573676603587e-17, rd);                  // pi/2 lo
    const double kq = k - 4.0 * floor(k * 0.25);                      // k mod 4 in {0,1,2,3}
    *r = (float)rd;
    *q = (int)kq;
}

static inline float mm_sin_poly(float r)
{
    const float z = r * r;
    float p = mm_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    p = mm_fmaf(z, p, -1.6666654611e-1f);
    return mm_fmaf(r * z, p, r);
}

static inline float mm_cos_poly(float r)
{
    const float z = r * r;
    float p = mm_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    p = mm_fmaf(z, p, 4.166664568298827e-2f);
    return mm_fmaf(z * z, p, mm_fmaf(-0.5f, z, 1.0f));
}

static inline float mm_sinf(float x)
{
    float r; int q;
    mm_reduce_pio2(x, &r, &q);
    const float s = mm_sin_poly(r);
    const float c = mm_cos_poly(r);
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}

static inline float mm_cosf(float x)
{
    float r; int q;
    mm_reduce_pio2(x, &r, &q);
    const float s = mm_sin_poly(r);
    const float c = mm_cos_poly(r);
    float v = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -v : v;
}

static inline void mm_sincosf(float x, float* sn, float* cs)
{
    *sn = mm_sinf(x);
    *cs = mm_cosf(x);
}

// ---- fp64 log / exp (fdlibm e_log.c / e_exp.c published algorithm, restated), used by powf.
static inline double mm_log(double x)   // x > 0, finite, normal
{
    uint64_t bits; std::memcpy(&bits, &x, 8);
    int k = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;   // m in [1,2)
    double m; std::memcpy(&m, &bits, 8);
    if (m > 1.41421356237309514547) { m = m * 0.5; k += 1; }          // m in (sqrt2/2, sqrt2]
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

static inline double mm_exp(double x)   // |x| < 700
{
    const double k = rint(x * 1.44269504088896338700e+00);
    const double hi = x - k * 6.93147180369123816490e-01;
    const double lo = k * 1.90821492927058770002e-10;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (1.66666666666666019037e-01 + t * (-2.77777777770155933842e-03 + t * (6.61375632143793436117e-05
                     + t * (-1.65339022054652515390e-06 + t * 4.13813679705723846039e-08))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    uint64_t bits; std::memcpy(&bits, &y, 8);
    bits += (uint64_t)((int64_t)k) << 52;                             // y * 2^k (no over/underflow in range)
    double out; std::memcpy(&out, &bits, 8);
    return out;
}

// powf for the reference's call sites (biomeFuncs.hpp:235,311,375; featurePlacement.hpp): x >= 0.
static inline float mm_powf(float x, float y)
{
    if (y == 2.f) return x * x;                 // correctly rounded x^2
    if (x == 0.f) return (y > 0.f) ? 0.f : 1.f;
    if (x == 1.f || y == 0.f) return 1.f;
    if (x < 0.f) return NAN;
    return (float)mm_exp((double)y * mm_log((double)x));
}

// atan(x) in fp64 (fdlibm s_atan.c algorithm restated), used by atan2f / acosf.
static inline double mm_atan(double x)
{
    static const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00};
    static const double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17};
    static const double aT[11] = {3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
                                  9.09088713343650656196e-02, -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02,
                                  4.97687799461593236017e-02, -3.65315727442169155270e-02, 1.62858201153657823623e-02};
    const bool neg = x < 0.0;
    double ax = neg ? -x : x;
    int id;
    if (ax >= 7.3786976294838206e+19) {           // 2^66
        return neg ? -1.57079632679489655800e+00 : 1.57079632679489655800e+00;
    }
    if (ax < 0.4375) {
        id = -1;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) { id = 0; ax = (2.0 * ax - 1.0) / (2.0 + ax); }
        else             { id = 1; ax = (ax - 1.0) / (ax + 1.0); }
    } else {
        if (ax < 2.4375) { id = 2; ax = (ax - 1.5) / (1.0 + 1.5 * ax); }
        else             { id = 3; ax = -1.0 / ax; }
    }
    const double z = ax * ax;
    const double w = z * z;
    const double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    double res;
    if (id < 0) res = ax - ax * (s1 + s2);
    else        res = atanhi[id] - ((ax * (s1 + s2) - atanlo[id]) - ax);
    return neg ? -res : res;
}

static inline float mm_atan2f(float y, float x)
{
    const double PI = 3.14159265358979311600e+00;
    const double yd = (double)y, xd = (double)x;
    if (xd == 0.0 && yd == 0.0) return 0.f;
    if (xd == 0.0) return (float)(yd > 0.0 ? 0.5 * PI : -0.5 * PI);
    const double a = mm_atan(yd / xd);
    if (xd > 0.0) return (float)a;
    return (float)(yd >= 0.0 ? a + PI : a - PI);
}

static inline float mm_acosf(float x)   // |x| <= 1
{
    const double xd = (double)x;
    const double s = sqrt((1.0 - xd) * (1.0 + xd));
    if (xd == 0.0) return (float)1.57079632679489655800e+00;
    const double PI = 3.14159265358979311600e+00;
    const double a = mm_atan(s / xd);
    return (float)(xd > 0.0 ? a : a + PI);
}

// fmodf is exact in IEEE arithmetic (the result is always representable), so glibc's fmodf and
// ocml's fmodf agree bit for bit; the oracle uses libm's.
static inline float mm_fmodf(float x, float y) { return fmodf(x, y); }

}  // namespace mmo
