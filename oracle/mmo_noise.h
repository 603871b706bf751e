// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Restatement of the reference math layer:
//   src/util/rng.hpp:69-96    hash, makeSeededRandomEngine x3
//   thrust minstd_rand + uniform_real_distribution<float> (CUDA toolkit thrust, not vendored; algorithm
//     confirmed against rocThrust 7.2: thrust/random/detail/linear_congruential_engine.inl:43-61,
//     detail/mod.h static_mod, detail/uniform_real_distribution.inl operator())
//   src/util/rng.hpp:102-155  rand{1,2,3}From{1,2,3}
//   external/include/glm/gtc/noise.inl:591-645 simplex(vec2), :648-721 simplex(vec3);
//     detail/_noise.hpp:8-60 mod289 / permute / taylorInvSqrt
//   src/util/rng.hpp:161-191  simplex2From2, fbm, fbm2From2, fbm3From3
//   src/util/rng.hpp:193-320  worley(vec2), worley(vec3), specialCaveNoise
#pragma once
#include <cfloat>
#include "mmo_vec.h"
#include "mmo_math.h"

namespace mmo {

// ---------------------------------------------------------------- utility functions (rng.hpp:9-63)
template <class T> static inline int manhattanLength(T v) { return compAdd(g_abs(v)); }
template <class T> static inline int manhattanDistance(T a, T b) { return compAdd(g_abs(a - b)); }
template <class T> static inline bool isInRange(T v, T min, T max) { return v >= min && v <= max; }
template <class T> static inline bool isPosInRange(T pos, T corner1, T corner2)
{
    T minPos = g_min(corner1, corner2);
    T maxPos = g_max(corner1, corner2);
    return pos.x >= minPos.x && pos.x <= maxPos.x
        && pos.y >= minPos.y && pos.y <= maxPos.y
        && pos.z >= minPos.z && pos.z <= maxPos.z;
}
static inline float getRatio(float v, float minVal, float maxVal) { return (v - minVal) / (maxVal - minVal); }
static inline float saturate(float v) { return g_clamp(v, 0.f, 1.f); }
static inline bool isSaturated(float v) { return v >= 0.f && v <= 1.f; }

static bool calculateLineParams(const vec3 pos, const vec3 linePos1, const vec3 linePos2, float* ratio, float* distFromLine)
{
    vec3 vecLine = linePos2 - linePos1;
    vec3 pointPos = pos - linePos1;
    *ratio = g_dot(pointPos, vecLine) / g_dot(vecLine, vecLine);
    vec3 pointLine = vecLine * (*ratio);
    *distFromLine = g_distance(pointPos, pointLine);
    return isSaturated(*ratio);
}


// ---------------------------------------------------------------- integer hash + minstd (rng.hpp:69-96)
static inline uint32_t hash_u32(uint32_t a)
{
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

struct Rng {           // thrust::minstd_rand: x <- 48271 x mod (2^31-1), min 1, max 2^31-2
    uint32_t x;
    explicit Rng(uint32_t seed)
    {
        uint32_t s = seed % 2147483647u;
        x = (s == 0u) ? 1u : s;
    }
    uint32_t next()
    {
        // static_mod<uint32, 48271, 0, 2147483647>: Schrage, q = m / a, r = m % a
        const uint32_t q = 2147483647u / 48271u, r = 2147483647u % 48271u;
        uint32_t t1 = 48271u * (x % q);
        uint32_t t2 = r * (x / q);
        x = (t1 >= t2) ? (t1 - t2) : (2147483647u - t2 + t1);
        return x;
    }
    // thrust::uniform_real_distribution<float>(0,1)
    float u01()
    {
        float result = (float)(next() - 1u);
        result /= (1.f + (float)(2147483646u - 1u));
        return (result * (1.f - 0.f)) + 0.f;
    }
};

// thrust::uniform_real_distribution<T>(a, b), thrust/random/detail/uniform_real_distribution.inl operator(): the engine's draw minus
// its minimum, over 1 + (max - min) in T, scaled to [a, b).  Declared the way the reference declares them
// (`thrust::uniform_real_distribution<float> u01(0, 1);`, featurePlacement.hpp:155-156) and called the same way (`u01(featureRng)`).
template <class T>
struct uniform_real_distribution {
    T a, b;
    uniform_real_distribution(T a_, T b_) : a(a_), b(b_) {}
    T operator()(Rng& urng) const
    {
        T result = (T)(urng.next() - 1u);
        result /= ((T)1 + (T)(2147483646u - 1u));
        return (result * (b - a)) + a;
    }
};

// The reference seeds in `int` arithmetic: (1 << 31) | (x << 22) | y shifts bits out of and into the sign bit, which is modular two's
// complement arithmetic since C++20 (the oracle is built with -std=c++20; CUDA's nvcc computes the same bits); the int converts to the
// unsigned parameter of hash() and the int result to the engine's unsigned seed modulo 2^32.
static inline Rng makeSeededRandomEngine(int x)
{
    int h = hash_u32(x);
    return Rng(h);
}
static inline Rng makeSeededRandomEngine(int x, int y, int z)
{
    int h = hash_u32((1 << 31) | (x << 22) | y) ^ hash_u32(z);
    return Rng(h);
}
static inline Rng makeSeededRandomEngine(int x, int y, int z, int w)
{
    int h = hash_u32((1 << 31) | (x << 22) | (y << 11) | w) ^ hash_u32(z);
    return Rng(h);
}

// ---------------------------------------------------------------- sin hashes (rng.hpp:102-155)
static inline float rand1From1(float v) { return g_fract(mm_sinf(v * 238.68f) * 39021.426f); }
static inline float rand1From2(vec2 v) { return g_fract(mm_sinf(g_dot(v, vec2(238.68f, 491.28f))) * 39021.426f); }
static inline float rand1From3(vec3 v) { return g_fract(mm_sinf(g_dot(v, vec3(238.68f, 491.28f, 640.88f))) * 39021.426f); }
static inline vec2 mm_sinf(vec2 v) { return vec2(mm_sinf(v.x), mm_sinf(v.y)); }             // glm::sin(vecN): component-wise
static inline vec3 mm_sinf(vec3 v) { return vec3(mm_sinf(v.x), mm_sinf(v.y), mm_sinf(v.z)); }
static inline vec2 rand2From2(vec2 v)
{
    return g_fract(mm_sinf(vec2(
        g_dot(v, vec2(238.68f, 491.28f)),
        g_dot(v, vec2(654.37f, 560.45f))
    )) * 39021.426f);
}
static inline vec2 rand2From3(vec3 v)
{
    return g_fract(mm_sinf(vec2(
        g_dot(v, vec3(238.68f, 491.28f, 640.88f)),
        g_dot(v, vec3(654.37f, 560.45f, 151.81f))
    )) * 39021.426f);
}
static inline vec3 rand3From2(vec2 v)
{
    return g_fract(mm_sinf(vec3(
        g_dot(v, vec2(238.68f, 491.28f)),
        g_dot(v, vec2(654.37f, 560.45f)),
        g_dot(v, vec2(640.88f, 151.81f))
    )) * 39021.426f);
}
static inline vec3 rand3From3(vec3 v)
{
    return g_fract(mm_sinf(vec3(
        g_dot(v, vec3(238.68f, 491.28f, 402.98f)),
        g_dot(v, vec3(654.37f, 560.45f, 747.42f)),
        g_dot(v, vec3(640.88f, 151.81f, 674.81f))
    )) * 39021.426f);
}

// ---------------------------------------------------------------- glm simplex (noise.inl:591-721)
static inline float mod289(float x) { return x - floorf(x * (1.f / 289.f)) * 289.f; }
static inline float permute(float x) { return mod289(((x * 34.f) + 1.f) * x); }
static inline vec3 permute(vec3 v) { return vec3(permute(v.x), permute(v.y), permute(v.z)); }
static inline vec4 permute(vec4 v) { return vec4(permute(v.x), permute(v.y), permute(v.z), permute(v.w)); }

// helper for "scalar + vec3" (glm: vec3(scalar + v.x, ...))
static inline vec3 operator+(float s, vec3 a) { return vec3(s + a.x, s + a.y, s + a.z); }
static inline vec4 operator+(float s, vec4 a) { return vec4(s + a.x, s + a.y, s + a.z, s + a.w); }

static inline float simplex(vec2 v)
{
    const float Cx = (float)0.211324865405187, Cy = (float)0.366025403784439,
                Cz = (float)-0.577350269189626, Cw = (float)0.024390243902439;
    vec2 i = g_floor(v + g_dot(v, vec2(Cy)));
    vec2 x0 = v - i + g_dot(i, vec2(Cx));
    vec2 i1 = (x0.x > x0.y) ? vec2(1.f, 0.f) : vec2(0.f, 1.f);
    vec4 x12 = vec4(x0.x, x0.y, x0.x, x0.y) + vec4(Cx, Cx, Cz, Cz);
    x12 = vec4(x12.x - i1.x, x12.y - i1.y, x12.z, x12.w);

    i = g_mod(i, 289.f);
    vec3 p = permute(permute(i.y + vec3(0.f, i1.y, 1.f)) + i.x + vec3(0.f, i1.x, 1.f));

    vec3 m = g_max(vec3(0.5f) - vec3(g_dot(x0, x0), g_dot(vec2(x12.x, x12.y), vec2(x12.x, x12.y)),
                                     g_dot(vec2(x12.z, x12.w), vec2(x12.z, x12.w))), vec3(0.f));
    m = m * m;
    m = m * m;

    vec3 x = 2.f * g_fract(p * Cw) - 1.f;
    vec3 h = g_abs(x) - 0.5f;
    vec3 ox = g_floor(x + 0.5f);
    vec3 a0 = x - ox;

    m *= (float)1.79284291400159 - (float)0.85373472095314 * (a0 * a0 + h * h);

    vec3 g;
    g.x = a0.x * x0.x + h.x * x0.y;
    g.y = a0.y * x12.x + h.y * x12.y;
    g.z = a0.z * x12.z + h.z * x12.w;
    return 130.f * g_dot(m, g);
}

static inline float simplex(vec3 v)
{
    const float Cx = (float)(1.0 / 6.0), Cy = (float)(1.0 / 3.0);
    const float Dx = 0.f, Dy = 0.5f, Dz = 1.f, Dw = 2.f;

    vec3 i = g_floor(v + g_dot(v, vec3(Cy)));
    vec3 x0 = v - i + g_dot(i, vec3(Cx));

    vec3 g = g_step(vec3(x0.y, x0.z, x0.x), x0);
    vec3 l = 1.f - g;
    vec3 i1 = g_min(g, vec3(l.z, l.x, l.y));
    vec3 i2 = g_max(g, vec3(l.z, l.x, l.y));

    vec3 x1 = x0 - i1 + Cx;
    vec3 x2 = x0 - i2 + Cy;
    vec3 x3 = x0 - Dy;

    i = vec3(mod289(i.x), mod289(i.y), mod289(i.z));
    vec4 p = permute(permute(permute(
        i.z + vec4(0.f, i1.z, i2.z, 1.f)) +
        i.y + vec4(0.f, i1.y, i2.y, 1.f)) +
        i.x + vec4(0.f, i1.x, i2.x, 1.f));

    const float n_ = (float)0.142857142857;
    vec3 ns = n_ * vec3(Dw, Dy, Dz) - vec3(Dx, Dz, Dx);

    vec4 j = p - 49.f * g_floor(p * ns.z * ns.z);

    vec4 x_ = g_floor(j * ns.z);
    vec4 y_ = g_floor(j - 7.f * x_);

    vec4 x = x_ * ns.x + ns.y;
    vec4 y = y_ * ns.x + ns.y;
    vec4 h = 1.f - g_abs(x) - g_abs(y);

    vec4 b0(x.x, x.y, y.x, y.y);
    vec4 b1(x.z, x.w, y.z, y.w);

    vec4 s0 = g_floor(b0) * 2.f + 1.f;
    vec4 s1 = g_floor(b1) * 2.f + 1.f;
    vec4 sh = -g_step(h, vec4(0.f, 0.f, 0.f, 0.f));

    vec4 a0 = vec4(b0.x, b0.z, b0.y, b0.w) + vec4(s0.x, s0.z, s0.y, s0.w) * vec4(sh.x, sh.x, sh.y, sh.y);
    vec4 a1 = vec4(b1.x, b1.z, b1.y, b1.w) + vec4(s1.x, s1.z, s1.y, s1.w) * vec4(sh.z, sh.z, sh.w, sh.w);

    vec3 p0(a0.x, a0.y, h.x);
    vec3 p1(a0.z, a0.w, h.y);
    vec3 p2(a1.x, a1.y, h.z);
    vec3 p3(a1.z, a1.w, h.w);

    vec4 norm = (float)1.79284291400159 - (float)0.85373472095314 * vec4(g_dot(p0, p0), g_dot(p1, p1), g_dot(p2, p2), g_dot(p3, p3));
    p0 *= norm.x;
    p1 *= norm.y;
    p2 *= norm.z;
    p3 *= norm.w;

    vec4 m = g_max(0.6f - vec4(g_dot(x0, x0), g_dot(x1, x1), g_dot(x2, x2), g_dot(x3, x3)), vec4(0.f, 0.f, 0.f, 0.f));
    m = m * m;
    return 42.f * g_dot(m * m, vec4(g_dot(p0, x0), g_dot(p1, x1), g_dot(p2, x2), g_dot(p3, x3)));
}

// ---------------------------------------------------------------- fbm family (rng.hpp:161-191)
static inline vec2 simplex2From2(vec2 pos) { return vec2(simplex(pos), simplex(pos + vec2(5923.45f, 4129.42f))); }

template <int octaves = 5, class T>
static inline float fbm(T pos)
{
    float fbm = 0.f;
    float amplitude = 1.f;
    for (int i = 0; i < octaves; ++i) {
        amplitude *= 0.5f;
        fbm += amplitude * simplex(pos);
        pos *= 2.f;
    }
    return fbm;
}

template <int octaves = 5>
static inline vec2 fbm2From2(vec2 pos)
{
    return vec2(fbm<octaves>(pos), fbm<octaves>(pos + vec2(5923.45f, 4129.42f)));
}

template <int octaves = 5>
static inline vec3 fbm3From3(vec3 pos)
{
    return vec3(fbm<octaves>(pos), fbm<octaves>(pos + vec3(5923.45f, 4129.42f, 5790.48f)),
                fbm<octaves>(pos + vec3(1765.68f, 4704.36f, 5692.12f)));
}

// ---------------------------------------------------------------- worley (rng.hpp:193-320)
static inline float worley(vec2 pos, vec3* colorPtr = nullptr, float* edgeDistPtr = nullptr)
{
    ivec2 uvInt = ivec2(g_floor(pos));
    vec2 uvFract = g_fract(pos);

    float minDist1 = FLT_MAX;
    float minDist2 = FLT_MAX;
    vec2 closestPoint;
    for (int x = -1; x <= 1; ++x) {
        for (int y = -1; y <= 1; ++y) {
            ivec2 neighbor = ivec2(x, y);
            vec2 point = rand2From2(uvInt + neighbor);
            vec2 diff = vec2(neighbor) + point - uvFract;
            float dist = g_length(diff);
            if (dist < minDist1) {
                minDist2 = minDist1;
                minDist1 = dist;
                closestPoint = point;
            } else if (dist < minDist2) {
                minDist2 = dist;
            }
        }
    }
    if (colorPtr != nullptr) *colorPtr = rand3From2(closestPoint);
    if (edgeDistPtr != nullptr) *edgeDistPtr = (minDist2 - minDist1) * 0.5f;
    return minDist1;
}

static inline float worley(vec3 pos, vec3* colorPtr = nullptr, float* edgeDistPtr = nullptr)
{
    ivec3 uvInt = ivec3(g_floor(pos));
    vec3 uvFract = g_fract(pos);

    float minDist1 = FLT_MAX;
    float minDist2 = FLT_MAX;
    vec3 closestPoint;
    for (int x = -1; x <= 1; ++x) {
        for (int y = -1; y <= 1; ++y) {
            for (int z = -1; z <= 1; ++z) {
                ivec3 neighbor = ivec3(x, y, z);
                vec3 point = rand3From3(uvInt + neighbor);
                vec3 diff = vec3(neighbor) + point - uvFract;
                float dist = g_length(diff);
                if (dist < minDist1) {
                    minDist2 = minDist1;
                    minDist1 = dist;
                    closestPoint = point;
                } else if (dist < minDist2) {
                    minDist2 = dist;
                }
            }
        }
    }
    if (colorPtr != nullptr) *colorPtr = rand3From3(closestPoint);
    if (edgeDistPtr != nullptr) *edgeDistPtr = (minDist2 - minDist1) * 0.5f;
    return minDist1;
}

static inline float specialCaveNoise(vec3 pos)
{
    ivec3 uvInt = ivec3(g_floor(pos));
    vec3 uvFract = g_fract(pos);

    float minDist1 = FLT_MAX;
    float minDist2 = FLT_MAX;
    float minDist3 = FLT_MAX;
    for (int x = -1; x <= 1; ++x) {
        for (int y = -1; y <= 1; ++y) {
            for (int z = -1; z <= 1; ++z) {
                ivec3 neighbor = ivec3(x, y, z);
                vec3 point = rand3From3(uvInt + neighbor);
                vec3 diff = vec3(neighbor) + point - uvFract;
                float dist = g_length(diff);
                if (dist < minDist1) {
                    minDist3 = minDist2;
                    minDist2 = minDist1;
                    minDist1 = dist;
                } else if (dist < minDist2) {
                    minDist3 = minDist2;
                    minDist2 = dist;
                } else if (dist < minDist3) {
                    minDist3 = dist;
                }
            }
        }
    }
    return minDist3 / minDist1 - 1.f;
}

}  // namespace mmo
