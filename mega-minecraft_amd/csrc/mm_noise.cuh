// mmgen device noise library for gfx950: permutation-polynomial simplex 2D/3D with glm 0.9.9.8's exact fp32
// operation order (reference math spec: external/include/glm/gtc/noise.inl:591-721, detail/_noise.hpp:8-60),
// fbm stacks, sin-hash cell points and Worley distance searches (reference: src/util/rng.hpp:102-320).
// Scalar formulation: every lane carries one sample; no vector temporaries, no lookup tables (the simplex
// variant used by the reference has none), Worley cell points optionally served from an LDS tile.
#pragma once
#include "mm_math.cuh"

namespace mm {

// Code-size control: the voxel kernels evaluate simplex noise at 20-40 call sites; fully inlined that is 80-110 KB of straight-line
// code per kernel, larger than the instruction cache.  MM_SIMPLEX_ATTR selects one shared (non-inlined) body per translation unit.
#ifndef MM_SIMPLEX_ATTR
#define MM_SIMPLEX_ATTR static __device__ __attribute__((noinline))
#endif

struct f2 { float x, y; };
struct f3 { float x, y, z; };

MM_DEV f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
MM_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }

// ---------------------------------------------------------------------------------------------------------
// simplex
// ---------------------------------------------------------------------------------------------------------
MM_DEV float mod289(float x) { return x - __builtin_floorf(x * (1.f / 289.f)) * 289.f; }
MM_DEV float permute(float x) { return mod289(((x * 34.f) + 1.f) * x); }

MM_SIMPLEX_ATTR float simplex2(float vx, float vy)
{
    const float C0 = (float)0.211324865405187, C1 = (float)0.366025403784439;
    const float C2 = (float)-0.577350269189626, C3 = (float)0.024390243902439;

    const float d = vx * C1 + vy * C1;
    float ix = __builtin_floorf(vx + d), iy = __builtin_floorf(vy + d);
    const float e = ix * C0 + iy * C0;
    const float x0x = (vx - ix) + e, x0y = (vy - iy) + e;

    const bool gt = x0x > x0y;
    const float i1x = gt ? 1.f : 0.f, i1y = gt ? 0.f : 1.f;
    const float ax = (x0x + C0) - i1x, ay = (x0y + C0) - i1y;     // x12.xy
    const float bx = x0x + C2, by = x0y + C2;                     // x12.zw

    ix = gmod(ix, 289.f);
    iy = gmod(iy, 289.f);
    const float p0 = permute((permute(iy + 0.f) + ix) + 0.f);
    const float p1 = permute((permute(iy + i1y) + ix) + i1x);
    const float p2 = permute((permute(iy + 1.f) + ix) + 1.f);

    float m0 = gmax(0.5f - (x0x * x0x + x0y * x0y), 0.f);
    float m1 = gmax(0.5f - (ax * ax + ay * ay), 0.f);
    float m2 = gmax(0.5f - (bx * bx + by * by), 0.f);
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;

    const float X0 = 2.f * fract(p0 * C3) - 1.f, X1 = 2.f * fract(p1 * C3) - 1.f, X2 = 2.f * fract(p2 * C3) - 1.f;
    const float h0 = __builtin_fabsf(X0) - 0.5f, h1 = __builtin_fabsf(X1) - 0.5f, h2 = __builtin_fabsf(X2) - 0.5f;
    const float a0 = X0 - __builtin_floorf(X0 + 0.5f), a1 = X1 - __builtin_floorf(X1 + 0.5f), a2 = X2 - __builtin_floorf(X2 + 0.5f);

    const float K1 = (float)1.79284291400159, K2 = (float)0.85373472095314;
    m0 = m0 * (K1 - K2 * (a0 * a0 + h0 * h0));
    m1 = m1 * (K1 - K2 * (a1 * a1 + h1 * h1));
    m2 = m2 * (K1 - K2 * (a2 * a2 + h2 * h2));

    const float g0 = a0 * x0x + h0 * x0y;
    const float g1 = a1 * ax + h1 * ay;
    const float g2 = a2 * bx + h2 * by;
    return 130.f * ((m0 * g0 + m1 * g1) + m2 * g2);
}

MM_SIMPLEX_ATTR float simplex3(float vx, float vy, float vz)
{
    const float Cx = (float)(1.0 / 6.0), Cy = (float)(1.0 / 3.0);

    const float d = (vx * Cy + vy * Cy) + vz * Cy;
    float ix = __builtin_floorf(vx + d), iy = __builtin_floorf(vy + d), iz = __builtin_floorf(vz + d);
    const float e = (ix * Cx + iy * Cx) + iz * Cx;
    const float x0x = (vx - ix) + e, x0y = (vy - iy) + e, x0z = (vz - iz) + e;

    // g = step(x0.yzx, x0); l = 1 - g; i1 = min(g, l.zxy); i2 = max(g, l.zxy)
    const float gx = (x0x < x0y) ? 0.f : 1.f, gy = (x0y < x0z) ? 0.f : 1.f, gz = (x0z < x0x) ? 0.f : 1.f;
    const float lx = 1.f - gx, ly = 1.f - gy, lz = 1.f - gz;
    const float i1x = gmin(gx, lz), i1y = gmin(gy, lx), i1z = gmin(gz, ly);
    const float i2x = gmax(gx, lz), i2y = gmax(gy, lx), i2z = gmax(gz, ly);

    const float x1x = (x0x - i1x) + Cx, x1y = (x0y - i1y) + Cx, x1z = (x0z - i1z) + Cx;
    const float x2x = (x0x - i2x) + Cy, x2y = (x0y - i2y) + Cy, x2z = (x0z - i2z) + Cy;
    const float x3x = x0x - 0.5f, x3y = x0y - 0.5f, x3z = x0z - 0.5f;

    ix = mod289(ix); iy = mod289(iy); iz = mod289(iz);

    float p[4];
    {
        const float oz[4] = {0.f, i1z, i2z, 1.f}, oy[4] = {0.f, i1y, i2y, 1.f}, ox[4] = {0.f, i1x, i2x, 1.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = permute(iz + oz[k]);
            const float b = permute((a + iy) + oy[k]);
            p[k] = permute((b + ix) + ox[k]);
        }
    }

    const float n_ = (float)0.142857142857;
    const float nsx = n_ * 2.f - 0.f, nsy = n_ * 0.5f - 1.f, nsz = n_ * 1.f - 0.f;

    float px[4], py[4], hh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float j = p[k] - 49.f * __builtin_floorf((p[k] * nsz) * nsz);
        const float x_ = __builtin_floorf(j * nsz);
        const float y_ = __builtin_floorf(j - 7.f * x_);
        px[k] = x_ * nsx + nsy;
        py[k] = y_ * nsx + nsy;
        hh[k] = (1.f - __builtin_fabsf(px[k])) - __builtin_fabsf(py[k]);
    }
    // b0 = (x.x, x.y, y.x, y.y), b1 = (x.z, x.w, y.z, y.w); s = floor(b)*2+1; sh = -step(h, 0)
    // a0 = b0.xzyw + s0.xzyw * sh.xxyy ; a1 = b1.xzyw + s1.xzyw * sh.zzww
    float gx_[4], gy_[4];   // gradient xy per corner: corner k uses (x[k], y[k]) with sh[k]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float sh = -((0.f < hh[k]) ? 0.f : 1.f);
        const float sx = __builtin_floorf(px[k]) * 2.f + 1.f;
        const float sy = __builtin_floorf(py[k]) * 2.f + 1.f;
        gx_[k] = px[k] + sx * sh;
        gy_[k] = py[k] + sy * sh;
    }

    const float K1 = (float)1.79284291400159, K2 = (float)0.85373472095314;
    float cx[4] = {x0x, x1x, x2x, x3x}, cy[4] = {x0y, x1y, x2y, x3y}, cz[4] = {x0z, x1z, x2z, x3z};
    float mm4[4], pd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float nrm = K1 - K2 * ((gx_[k] * gx_[k] + gy_[k] * gy_[k]) + hh[k] * hh[k]);
        const float qx = gx_[k] * nrm, qy = gy_[k] * nrm, qz = hh[k] * nrm;
        float m = gmax(0.6f - ((cx[k] * cx[k] + cy[k] * cy[k]) + cz[k] * cz[k]), 0.f);
        m = m * m;
        mm4[k] = m * m;
        pd[k] = (qx * cx[k] + qy * cy[k]) + qz * cz[k];
    }
    return 42.f * ((mm4[0] * pd[0] + mm4[1] * pd[1]) + (mm4[2] * pd[2] + mm4[3] * pd[3]));
}

// ---------------------------------------------------------------------------------------------------------
// simplex3 split at the lattice: part1 (skewed cell + offsets inside it) -> gradients of the cell's 4 simplex corners
// -> part3 (falloff * gradient . offset).  The gradients are a function of (cell, corner ordering) ONLY, so voxels that
// share a cell can share them through LDS (k_cave_voxels).  simplex3_part3(part1(v), grad(part1(v))) == simplex3(v)
// bit for bit: the same operations in the same order, merely regrouped (checked by the probe tests against real glm).
// ---------------------------------------------------------------------------------------------------------
struct Sx3Cell { float ix, iy, iz; float x0x, x0y, x0z; int order; };      // order bits: gx | gy << 1 | gz << 2

MM_DEV Sx3Cell simplex3_part1(float vx, float vy, float vz)
{
    const float Cx = (float)(1.0 / 6.0), Cy = (float)(1.0 / 3.0);
    Sx3Cell c;
    const float d = (vx * Cy + vy * Cy) + vz * Cy;
    c.ix = __builtin_floorf(vx + d); c.iy = __builtin_floorf(vy + d); c.iz = __builtin_floorf(vz + d);
    const float e = (c.ix * Cx + c.iy * Cx) + c.iz * Cx;
    c.x0x = (vx - c.ix) + e; c.x0y = (vy - c.iy) + e; c.x0z = (vz - c.iz) + e;
    c.order = ((c.x0x < c.x0y) ? 0 : 1) | ((c.x0y < c.x0z) ? 0 : 2) | ((c.x0z < c.x0x) ? 0 : 4);
    return c;
}

// 12 floats: (qx, qy, qz) of the 4 corners, already scaled by taylorInvSqrt
MM_DEV void simplex3_gradients(float ix, float iy, float iz, int order, float* __restrict__ q)
{
    const float gx = (order & 1) ? 1.f : 0.f, gy = (order & 2) ? 1.f : 0.f, gz = (order & 4) ? 1.f : 0.f;
    const float lx = 1.f - gx, ly = 1.f - gy, lz = 1.f - gz;
    const float i1x = gmin(gx, lz), i1y = gmin(gy, lx), i1z = gmin(gz, ly);
    const float i2x = gmax(gx, lz), i2y = gmax(gy, lx), i2z = gmax(gz, ly);
    ix = mod289(ix); iy = mod289(iy); iz = mod289(iz);
    const float oz[4] = {0.f, i1z, i2z, 1.f}, oy[4] = {0.f, i1y, i2y, 1.f}, ox[4] = {0.f, i1x, i2x, 1.f};
    const float n_ = (float)0.142857142857;
    const float nsx = n_ * 2.f - 0.f, nsy = n_ * 0.5f - 1.f, nsz = n_ * 1.f - 0.f;
    const float K1 = (float)1.79284291400159, K2 = (float)0.85373472095314;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float a = permute(iz + oz[k]);
        const float b = permute((a + iy) + oy[k]);
        const float p = permute((b + ix) + ox[k]);
        const float j = p - 49.f * __builtin_floorf((p * nsz) * nsz);
        const float x_ = __builtin_floorf(j * nsz);
        const float y_ = __builtin_floorf(j - 7.f * x_);
        const float px = x_ * nsx + nsy;
        const float py = y_ * nsx + nsy;
        const float hh = (1.f - __builtin_fabsf(px)) - __builtin_fabsf(py);
        const float sh = -((0.f < hh) ? 0.f : 1.f);
        const float sx = __builtin_floorf(px) * 2.f + 1.f;
        const float sy = __builtin_floorf(py) * 2.f + 1.f;
        const float g0 = px + sx * sh;
        const float g1 = py + sy * sh;
        const float nrm = K1 - K2 * ((g0 * g0 + g1 * g1) + hh * hh);
        q[3 * k] = g0 * nrm; q[3 * k + 1] = g1 * nrm; q[3 * k + 2] = hh * nrm;
    }
}

MM_DEV float simplex3_part3(const Sx3Cell& c, const float* __restrict__ q)
{
    const float Cx = (float)(1.0 / 6.0), Cy = (float)(1.0 / 3.0);
    const float gx = (c.order & 1) ? 1.f : 0.f, gy = (c.order & 2) ? 1.f : 0.f, gz = (c.order & 4) ? 1.f : 0.f;
    const float lx = 1.f - gx, ly = 1.f - gy, lz = 1.f - gz;
    const float i1x = gmin(gx, lz), i1y = gmin(gy, lx), i1z = gmin(gz, ly);
    const float i2x = gmax(gx, lz), i2y = gmax(gy, lx), i2z = gmax(gz, ly);
    const float cx[4] = {c.x0x, (c.x0x - i1x) + Cx, (c.x0x - i2x) + Cy, c.x0x - 0.5f};
    const float cy[4] = {c.x0y, (c.x0y - i1y) + Cx, (c.x0y - i2y) + Cy, c.x0y - 0.5f};
    const float cz[4] = {c.x0z, (c.x0z - i1z) + Cx, (c.x0z - i2z) + Cy, c.x0z - 0.5f};
    float mm4[4], pd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float m = gmax(0.6f - ((cx[k] * cx[k] + cy[k] * cy[k]) + cz[k] * cz[k]), 0.f);
        m = m * m;
        mm4[k] = m * m;
        pd[k] = (q[3 * k] * cx[k] + q[3 * k + 1] * cy[k]) + q[3 * k + 2] * cz[k];
    }
    return 42.f * ((mm4[0] * pd[0] + mm4[1] * pd[1]) + (mm4[2] * pd[2] + mm4[3] * pd[3]));
}

// ---------------------------------------------------------------------------------------------------------
// fbm stacks (rng.hpp:166-191): amplitude halves, frequency doubles, octaves summed in order
// ---------------------------------------------------------------------------------------------------------
template <int OCT>
MM_DEV float fbm2(float x, float y)
{
    float acc = 0.f, amp = 1.f;
#pragma unroll
    for (int i = 0; i < OCT; ++i) {
        amp *= 0.5f;
        acc += amp * simplex2(x, y);
        x *= 2.f; y *= 2.f;
    }
    return acc;
}

template <int OCT>
MM_DEV float fbm3(float x, float y, float z)
{
    float acc = 0.f, amp = 1.f;
#pragma unroll
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
    return mk3(fbm3<OCT>(x, y, z), fbm3<OCT>(x + 5923.45f, y + 4129.42f, z + 5790.48f), fbm3<OCT>(x + 1765.68f, y + 4704.36f, z + 5692.12f));
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
    MM_DEV f3 operator()(int cx, int cy, int cz) const { return rand3from3((float)cx, (float)cy, (float)cz); }
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

template <class Cells>
MM_DEV float special_cave_noise(float px, float py, float pz, const Cells& cells)
{
    const Worley3 w = worley3(px, py, pz, cells);
    return w.d3 / w.d1 - 1.f;
}

}  // namespace mm
