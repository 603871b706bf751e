// mmgen device feature rasterisers: per (voxel, placement) implicit-surface tests for the 21 surface features and the 10 cave
// features.  Behavioural spec: src/terrain/featurePlacement.hpp (helpers :15-142, placeFeature :147-1107, placeCaveFeature
// :1110-1379), line helpers src/util/rng.hpp:9-63.  Random draws that the reference makes inside one constructor call are
// taken left to right (canonical order).  Returns true and sets `out` when the placement claims the voxel.
#pragma once
#include "mm_biome.cuh"

namespace mm {

struct v3 { float x, y, z; };
struct i3 { int x, y, z; };

MM_DEV v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
MM_DEV v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
MM_DEV v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
MM_DEV v3 operator*(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
MM_DEV v3 operator*(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
MM_DEV v3 operator*(float s, v3 a) { return V3(s * a.x, s * a.y, s * a.z); }
MM_DEV v3 operator+(v3 a, float s) { return V3(a.x + s, a.y + s, a.z + s); }
MM_DEV v3 operator-(v3 a, float s) { return V3(a.x - s, a.y - s, a.z - s); }
MM_DEV float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
MM_DEV float len3(v3 a) { return __builtin_sqrtf(dot3(a, a)); }
MM_DEV float len2(float x, float y) { return __builtin_sqrtf(x * x + y * y); }
MM_DEV float dist3(v3 a, v3 b) { return len3(b - a); }
MM_DEV v3 norm3(v3 a) { return a * (1.f / __builtin_sqrtf(dot3(a, a))); }
MM_DEV v3 mix3(v3 a, v3 b, float t) { return a * (1.f - t) + b * t; }
MM_DEV v3 floor3(v3 a) { return V3(__builtin_floorf(a.x), __builtin_floorf(a.y), __builtin_floorf(a.z)); }
MM_DEV v3 ceil3(v3 a) { return V3(__builtin_ceilf(a.x), __builtin_ceilf(a.y), __builtin_ceilf(a.z)); }
MM_DEV v3 cross3(v3 a, v3 b) { return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
MM_DEV int iabs(int v) { return v < 0 ? -v : v; }

#define MM_PI 3.14159265358979323846264338327f
#define MM_TWO_PI 6.28318530717958647692528676655f
#define MM_PI_OVER_TWO 1.57079632679489661923132169163f

// A minstd stream that is seeded (two integer hashes) on its first draw: most voxels a rasteriser looks at never draw from the per-voxel
// stream.  Every use site draws at most once before returning, so the first draw is the only one.
struct LazyRng {
    int x, y, z, w;
    MM_DEV float u01() { MinStd r = rng4(x, y, z, w); return r.u01(); }
};

MM_DEV float u11(MinStd& r) { return (r.u01() * (1.f - -1.f)) + -1.f; }
MM_DEV bool in_range_f(float v, float lo, float hi) { return v >= lo && v <= hi; }
MM_DEV bool in_range_i(int v, int lo, int hi) { return v >= lo && v <= hi; }
MM_DEV bool saturated(float v) { return v >= 0.f && v <= 1.f; }
MM_DEV float get_ratio(float v, float lo, float hi) { return (v - lo) / (hi - lo); }
MM_DEV float rand1from1(float v) { return hash_unit(v * 238.68f); }
MM_DEV void sincos_(float a, float& s, float& c) { s = sinf_(a); c = cosf_(a); }

struct LineParams { float ratio, dist; bool in; };

MM_DEV LineParams line_params(v3 pos, v3 l1, v3 l2)      // calculateLineParams rng.hpp:52-63
{
    const v3 vl = l2 - l1;
    const v3 pp = pos - l1;
    LineParams r;
    r.ratio = dot3(pp, vl) / dot3(vl, vl);
    r.dist = dist3(pp, vl * r.ratio);
    r.in = saturated(r.ratio);
    return r;
}

MM_DEV bool in_rasterized_line(i3 fp, v3 l1, v3 l2)      // featurePlacement.hpp:68-74
{
    const LineParams lp = line_params(V3((float)fp.x, (float)fp.y, (float)fp.z) + V3(0.5f, 0.5f, 0.5f), l1, l2);
    if (!(lp.in && lp.dist < 2.f)) return false;
    const v3 m = floor3(mix3(l1, l2, lp.ratio));
    return fp.x == (int)m.x && fp.y == (int)m.y && fp.z == (int)m.z;
}

MM_DEV bool jungle_leaves(v3 pos, float maxHeight, float minRadius, float maxRadius, float rand)   // :80-90
{
    const float mult = 0.8f + 0.4f * rand;
    if (in_range_f(pos.y, 0.f, maxHeight)) {
        const float r = mixf(maxRadius, minRadius, pos.y / maxHeight) * mult;
        return len2(pos.x, pos.z) < r;
    }
    return false;
}

MM_DEV float crystal_radius(float ratio)                  // :92-105
{
    const float coneStart = 0.8f;
    const float coneN = 1.f / (1.f - coneStart);
    if (ratio < coneStart) return 0.8f + 0.25f * ratio;
    return coneN * (1.f - ratio);
}

MM_DEV bool in_crystal(v3 pos, v3 p1, v3 p2, float radiusMultiplier)   // :107-125
{
    const LineParams lp = line_params(pos, p1, p2);
    if (!lp.in) return false;
    float radius = crystal_radius(lp.ratio) * radiusMultiplier;
    const float p = MM_PI / 6.f;
    const v3 line = p2 - p1;
    const v3 pp = pos - (p1 + lp.ratio * line);
    float posAngle;
    if (len3(pp) == 0.f) posAngle = 0.f;
    else {
        const v3 a = norm3(pp), b = norm3(cross3(line, V3(1.f, 0.f, 0.f)));
        posAngle = acosf_(clampf(dot3(a, b), -1.f, 1.f)) + MM_TWO_PI;
    }
    radius *= cosf_(p) / cosf_(p - fmodf_(posAngle, 2.f * p));
    return lp.dist < radius;
}

MM_DEV uint8_t random_crystal_block(float rand)           // :127-142
{
    const float c = rand * 3.f;
    if (c < 1.f) return MMB_MAGENTA_CRYSTAL;
    if (c < 2.f) return MMB_CYAN_CRYSTAL;
    return MMB_GREEN_CRYSTAL;
}

MM_DEV float sd_sphere(v3 p, float s) { return len3(p) - s; }
MM_DEV float sd_capped_cylinder(v3 p, float r, float h)
{
    const float dx = __builtin_fabsf(len2(p.x, p.z)) - r, dy = __builtin_fabsf(p.y) - h;
    return __builtin_fminf(__builtin_fmaxf(dx, dy), 0.0f) + len2(gmax(dx, 0.f), gmax(dy, 0.f));
}

template <int NC, int NS>
MM_DEV void de_casteljau(const v3* ctrl, v3* spline)      // :40-66
{
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        v3 c[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) c[j] = ctrl[j];
        const float t = float(i) / (NS - 1);
#pragma unroll
        for (int points = NC; points > 1; --points) {
#pragma unroll
            for (int j = 0; j < NC - 1; ++j)
                if (j < points - 1) c[j] = mix3(c[j], c[j + 1], t);
        }
        spline[i] = c[0];
    }
}

// =========================================================================================================
// surface features
// =========================================================================================================
// `fstate` = the placement's own stream right after seeding (surface_feature_stream): it depends on the placement alone, so a caller that
// evaluates many voxels of one placement seeds it once instead of hashing twice per voxel.
MM_DEV uint32_t surface_feature_stream(int fx, int fy, int fz) { return rng4(fx, fy, fz, 1293012).x; }
MM_DEV uint32_t cave_feature_stream(int fx, int fy, int fz) { return rng4(fx, fy, fz, 398132).x; }

MM_DEV bool place_feature(int feature, int fx, int fy, int fz, int wx, int wy, int wz, uint32_t fstate, uint8_t& out)
{
    const i3 fp = {wx - fx, wy - fy, wz - fz};
    v3 pos = V3((float)fp.x, (float)fp.y, (float)fp.z);
    const v3 wbp = V3((float)wx, (float)wy, (float)wz);
    MinStd frng; frng.x = fstate;                 // = rng4(fx, fy, fz, 1293012) (featurePlacement.hpp:153)
    LazyRng brng = {wx, wy, wz, 57847812};       // the per-voxel stream (featurePlacement.hpp:154): seeded only where a rule draws from it

    switch (feature) {
    case MMF_SPHERE: {
        if (dot3(pos, pos) > 25.f) return false;
        out = MMB_GRAVEL;
        return true;
    }
    case MMF_CORAL: {
        if (fy > MMGEN_SEA_LEVEL - 6) return false;
        if (len2(pos.x, pos.z) > 8.f) return false;
        const int kind = (int)(frng.u01() * 5.f);
        if (kind == 0 || kind == 1) {
            pos.y *= kind == 0 ? 1.15f : 1.25f;
            float radius = kind == 0 ? (2.8f + 1.4f * frng.u01()) : (2.2f + 1.7f * frng.u01());
            const float sc = kind == 0 ? 0.2f : 0.3f;
            radius += (kind == 0 ? 0.4f : 1.2f) * simplex3(wbp.x * sc, wbp.y * sc, wbp.z * sc);
            if (len3(pos) < radius) { out = kind == 0 ? MMB_BRAIN_CORAL_BLOCK : MMB_BUBBLE_CORAL_BLOCK; return true; }
            return false;
        }
        if (kind == 2 || kind == 3) {
            const uint8_t block = kind == 2 ? MMB_FIRE_CORAL_BLOCK : MMB_HORN_CORAL_BLOCK;
            const float r0 = u11(frng), r1 = frng.u01(), r2 = u11(frng);
            const v3 p1 = V3(r0, r1, r2) * V3(2.5f, 3.5f, 2.5f);
            if (in_rasterized_line(fp, V3(0.f, 0.f, 0.f), p1)) { out = block; return true; }
            for (int i = 0; i < 5; ++i) {
                v3 p2 = p1;
                p2.x += 4.f * u11(frng);
                p2.y += 2.f + 3.f * frng.u01();
                p2.z += 4.f * u11(frng);
                if (in_rasterized_line(fp, p1, p2)) { out = block; return true; }
            }
            return false;
        }
        if (kind == 4) {
            const Worley2 w = worley2(wbp.x * 0.7f, wbp.z * 0.7f);
            float h = (1.f - w.d1) + (w.d2 - w.d1) * 0.5f;
            h *= 3.5f;
            h *= smoothstep(3.7f, 2.5f, len2(pos.x, pos.z));
            h -= 2.f;
            if (in_range_f(pos.y, -1.f, h)) { out = MMB_TUBE_CORAL_BLOCK; return true; }
            return false;
        }
        return false;
    }
    case MMF_KELP: {
        if (fp.x != 0 || fp.z != 0) return false;
        int height = (int)(5.f + 15.f * frng.u01());
        height = imin(height, MMGEN_SEA_LEVEL - fy - 1);
        if (!in_range_i(fp.y, 0, height)) return false;
        out = (fp.y == height) ? MMB_KELP_END : MMB_KELP_MAIN;
        return true;
    }
    case MMF_ICEBERG: {
        if (fy > MMGEN_SEA_LEVEL - 32) return false;
        pos.y = (float)(wy - MMGEN_SEA_LEVEL);
        const float hd = len2(pos.x, pos.z);
        const float radius = 20.f + 12.f * frng.u01();
        const float ratio = 1.f - (hd / radius);
        if (ratio > 1.15f) return false;
        const float nx = wbp.x * 0.0450f, nz = wbp.z * 0.0450f;
        const float f = fbm2<3>(nx, nz);
        const float startH = (-6.f - 34.f * ratio) + 14.f * f;
        const float endH = (-4.f + 20.f * ratio) + 8.f * f;
        if (endH < startH || !in_range_f(pos.y, startH, endH)) return false;
        if (pos.y < -4.f) { out = MMB_BLUE_ICE; return true; }
        const float packed = (-2.2f + 5.6f * ratio) + 1.2f * simplex2(nx * 0.8000f, nz * 0.8000f);
        out = (pos.y > endH - packed) ? MMB_PACKED_ICE : MMB_BLUE_ICE;
        return true;
    }
    case MMF_ACACIA_TREE: {
        if (imax(iabs(fp.x), iabs(fp.z)) > 15) return false;
        const int trunk = (int)(4.5f + 1.5f * frng.u01());
        if (fp.x == 0 && fp.z == 0 && in_range_i(fp.y, 0, trunk)) { out = MMB_ACACIA_WOOD; return true; }

        float angle = frng.u01() * MM_TWO_PI;
        v3 bs = V3(0.f, (float)trunk, 0.f);
        v3 be = V3(0.f, 0.f, 0.f);
        sincos_(angle, be.z, be.x);
        be = bs + (2.f + 1.5f * frng.u01()) * be;
        be.y += 2.5f + 1.5f * frng.u01();
        if (in_rasterized_line(fp, floor3(bs), ceil3(be))) { out = MMB_ACACIA_WOOD; return true; }
        v3 lp = pos - be;
        lp.y += 0.5f;
        if (jungle_leaves(lp, 2.f, 2.f, 4.f, 0.5f + 0.5f * frng.u01())) { out = MMB_ACACIA_LEAVES; return true; }

        if (frng.u01() < 0.5f) return false;

        angle += MM_PI_OVER_TWO + frng.u01() * MM_PI;
        bs = V3(0.f, ((float)trunk - 0.8f) - 0.8f * frng.u01(), 0.f);
        be = V3(0.f, 0.f, 0.f);
        sincos_(angle, be.z, be.x);
        be = bs + (1.5f + 1.f * frng.u01()) * be;
        be.y += 2.f + 1.f * frng.u01();
        if (in_rasterized_line(fp, floor3(bs), ceil3(be))) { out = MMB_ACACIA_WOOD; return true; }
        lp = pos - be;
        lp.y += 0.5f;
        if (jungle_leaves(lp, 2.001f, 1.5f, 3.5f, 0.5f + 0.5f * frng.u01())) { out = MMB_ACACIA_LEAVES; return true; }
        return false;
    }
    case MMF_REDWOOD_TREE: {
        pos = pos * (0.6f + 0.3f * frng.u01());
        const float height = 27.f + 13.f * frng.u01();
        const float hd = len2(pos.x, pos.z);
        const float leavesStart = 10.f + 4.f * frng.u01();
        if (pos.y > height + 8.f || hd > 12.f || (pos.y < leavesStart - 4.f && hd > 3.f)) return false;

        const float tr = get_ratio(pos.y, -4.f, height);
        if (saturated(tr)) {
            float radius = 2.f / (tr + 2.f) + 0.08f / powf_(tr + 0.4f, 3.f);
            radius += (0.3f * simplex3(wbp.x * 0.1300f, wbp.y * 0.1300f, wbp.z * 0.1300f)) * smoothstep(0.6f, 0.2f, tr);
            if (hd < radius) { out = MMB_REDWOOD_WOOD; return true; }
        }
        const float leavesEnd = (height + 1.5f) + 1.f * frng.u01();
        if (!in_range_f(pos.y, leavesStart, leavesEnd)) return false;

        const int cellBase = (int)__builtin_floorf(pos.y * 0.5f) * 2;
        const float branchSeed = 593.23f * rand1from3((float)fx, (float)fy, (float)fz);
        const float leavesSeed = 412.39f * rand1from1(branchSeed);
        const float leavesSimplex = 1.1f * simplex3(wbp.x * 0.2000f, wbp.y * 0.2000f, wbp.z * 0.2000f);
        bool inLeaves = false;
        for (int dy = -4; dy <= 4; dy += 2) {
            const int cell = cellBase + dy;
            float hr = get_ratio((float)cell, leavesStart, leavesEnd);
            hr = 1.1f - 0.5f * hr;
            const f3 rc = rand3from2((float)cell, leavesSeed);
            v3 center = V3(rc.x - 0.5f, rc.y - 0.5f, rc.z - 0.5f);
            center = center * (V3(7.5f, 1.3f, 7.5f) * hr);
            center.y = gmin(center.y + (float)cell, height + 0.8f);

            const v3 bs = V3(0.f, (center.y - 2.f) - 1.5f * rand1from1((float)cell + branchSeed), 0.f);
            const LineParams lp = line_params(pos, bs, center);
            if (lp.in && lp.dist < 0.5f) { out = MMB_REDWOOD_WOOD; return true; }
            if (inLeaves) continue;
            v3 q = pos - center;
            q.y *= 1.7f;
            const float d = len3(q);
            if (d > 5.0f) continue;
            float radius = (2.5f + 0.5f * rand1from1((float)cell + leavesSeed)) + leavesSimplex;
            radius *= hr;
            if (d < radius) inLeaves = true;
        }
        if (inLeaves) { out = MMB_REDWOOD_LEAVES; return true; }
        return false;
    }
    case MMF_CYPRESS_TREE: {
        const float trunkHeight = 25.f + 12.f * frng.u01();
        const float td = len2(pos.x, pos.z);
        if (pos.y > trunkHeight + 4.f || td > 12.f) return false;
        const float tr = get_ratio(pos.y, -2.f, trunkHeight);
        if (saturated(tr)) {
            float radius = 0.5f * ((1.3f + tr) / powf_(0.73f + tr, 4.f)) + 0.5f;
            radius *= (1.f + (0.3f * simplex3(wbp.x * 0.1500f, wbp.y * 0.1500f, wbp.z * 0.1500f)) * smoothstep(0.55f, 0.15f, tr));
            if (td < radius) { out = MMB_CYPRESS_WOOD; return true; }
        }
        if (jungle_leaves(pos - V3(0.f, trunkHeight, 0.f), 2.f, 3.f, 4.5f, frng.u01())) { out = MMB_CYPRESS_LEAVES; return true; }

        const int numBranches = 6 + (int)(frng.u01() * 5.f);
        float bh = trunkHeight - 1.f;
        float angle = frng.u01() * MM_TWO_PI;
        const float droop = rand1from2(wbp.x, wbp.z);
        for (int i = 0; i < numBranches; ++i) {
            bh -= 1.f + 3.6f * frng.u01();
            angle += MM_PI_OVER_TWO + frng.u01() * MM_PI;
            const v3 bs = V3(0.f, bh, 0.f);
            v3 be = V3(0.f, 0.f, 0.f);
            sincos_(angle, be.z, be.x);
            const float sc = 4.f + 1.5f * frng.u01();
            be.x *= sc; be.z *= sc;
            be.y = 2.2f + 1.2f * frng.u01();
            be = be * (1.f - 0.3f * get_ratio(bh, 0.f, trunkHeight));
            be = be + bs;
            const i3 ip = {(int)pos.x, (int)pos.y, (int)pos.z};
            if (in_rasterized_line(ip, bs, be)) { out = MMB_CYPRESS_WOOD; return true; }
            v3 lp = (pos - be) + 0.3f;
            if (droop < 0.2f && in_range_f(lp.y, gmax(-2.f, droop * -10.f), 0.f)) lp.y = 0.f;
            if (jungle_leaves(lp, 2.f, 2.5f, 4.f, frng.u01())) { out = MMB_CYPRESS_LEAVES; return true; }
        }
        return false;
    }
    case MMF_BIRCH_TREE: {
        int height = (int)(6.2f + 4.f * frng.u01());
        const bool tall = frng.u01() < 0.08f;
        if (tall) height = (int)((float)height * 1.9f);
        if (imax(iabs(fp.x), iabs(fp.z)) > 8 || !in_range_i(fp.y, 0, height + 6)) return false;
        if (fp.x == 0 && fp.z == 0 && in_range_i(fp.y, 0, height)) { out = MMB_BIRCH_WOOD; return true; }
        const float mult = tall ? 1.5f : 1.f;
        const float ls = (float)height - (3.0f - 2.2f * frng.u01()) * mult;
        const float le = (float)height + (4.2f + 1.2f * frng.u01()) * mult;
        const float ratio = (pos.y - ls) / (le - ls);
        if (!in_range_f(ratio, 0.f, 1.f)) return false;
        const float x = powf_(ratio, 0.8f);
        const float radius = (5.f * (((0.5f * x * x * x) - (1.5f * x * x)) + x)) * (2.8f + 0.8f * frng.u01());
        if (len2(pos.x, pos.z) > radius) return false;
        const float lr = frng.u01();
        out = lr < 0.1f ? MMB_YELLOW_BIRCH_LEAVES : (lr < 0.2f ? MMB_ORANGE_BIRCH_LEAVES : MMB_BIRCH_LEAVES);
        return true;
    }
    case MMF_PINE_TREE: {
        const int height = (int)(7.f + 4.f * frng.u01());
        if (fp.y < 0 || fp.y > height + 4 || imax(iabs(fp.x), iabs(fp.z)) > 6) return false;
        if (fp.x == 0 && fp.z == 0 && fp.y <= height) { out = MMB_PINE_WOOD; return true; }
        const float ls = ((float)height - 4.f) - 2.5f * frng.u01();
        const float le = (float)height + 3.f;
        const float ratio = (pos.y - ls) / (le - ls);
        if (!in_range_f(ratio, 0.f, 1.f)) return false;
        const float radius = mixf(3.f, 1.f, ratio);
        if (len2(pos.x, pos.z) < radius) { out = frng.u01() < 0.5f ? MMB_PINE_LEAVES_1 : MMB_PINE_LEAVES_2; return true; }
        return false;
    }
    case MMF_PINE_SHRUB: {
        const int height = (int)(2.f + 2.f * frng.u01());
        if (fp.y < 0 || fp.y > height + 4 || imax(iabs(fp.x), iabs(fp.z)) > 6) return false;
        if (fp.x == 0 && fp.z == 0 && fp.y <= height) { out = MMB_PINE_WOOD; return true; }
        if (jungle_leaves(pos - V3(0.f, (float)height - 1.f, 0.f), 2.5f, 1.5f, 2.5f, frng.u01())) {
            out = frng.u01() < 0.5f ? MMB_PINE_LEAVES_1 : MMB_PINE_LEAVES_2;
            return true;
        }
        return false;
    }
    case MMF_MEDIUM_PURPLE_MUSHROOM: {
        if (iabs(fp.x) + iabs(fp.z) > 8) return false;
        const int height = (int)(1.5f + 2.3f * frng.u01());
        if (fp.x == 0 && in_range_i(fp.y, 0, height) && fp.z == 0) { out = MMB_MUSHROOM_STEM; return true; }
        const float radius = frng.u01() < 0.5f ? 1.8f : 2.5f;
        if (fp.y == height + 1 && len2(pos.x, pos.z) < radius) { out = MMB_PURPLE_MUSHROOM_CAP; return true; }
        return false;
    }
    case MMF_PURPLE_MUSHROOM: {
        const float universalScale = 1.f + frng.u01() * 1.2f;
        pos = pos * universalScale;
        if (frng.u01() < 0.2f) pos = pos * 0.5f;
        const float height = 25.f + frng.u01() * 30.f;
        if (pos.y < -1.f || pos.y > height + 12.f
            || (len2(pos.x, pos.z) > 8.f && (pos.y < height - 12.f || len3(pos - V3(0.f, height, 0.f)) > 35.f)))
            return false;

        v3 ctrl[5];
        ctrl[0] = V3(0.f, 0.f, 0.f);
        const v3 endPoint = V3(0.f, height, 0.f);
#pragma unroll
        for (int i = 1; i < 5; ++i) {
            const float r0 = u11(frng), r1 = u11(frng), r2 = u11(frng);
            v3 off = V3(r0, r1, r2) * V3(6.f, 2.f, 6.f);
            if (i == 4) off = off * 0.6f;
            ctrl[i] = (endPoint * ((float)i / 4.f)) + off;
        }
        v3 spline[7];
        de_casteljau<5, 7>(ctrl, spline);

        for (int i = 0; i < 7; ++i) {
            const v3 p1 = spline[i];
            v3 p2;
            if (i < 6) {
                p2 = spline[i + 1];
                if (pos.y < p1.y - 3.f || pos.y > p2.y + 3.f) continue;
            } else {
                p2 = p1 + norm3(p1 - spline[i - 1]) * (3.f + frng.u01() * 1.5f);
            }
            const LineParams lp = line_params(pos, p1, p2);
            float radius;
            uint8_t block;
            if (i < 6) {
                const float t = ((float)i + clampf(lp.ratio, 0.f, 1.f)) / 6.f;
                const float x = t - 0.5f;
                radius = (4.f * x * x + 1.5f) * 1.2f;
                block = MMB_MUSHROOM_STEM;
            } else {
                radius = (7.f * frng.u01() + 12.f) * mixf(0.8f, 1.2f, (height - 33.f) / 40.f);
                block = (lp.dist < radius - 1.8f && lp.ratio < 0.5f && universalScale < 1.4f) ? MMB_MUSHROOM_UNDERSIDE : MMB_PURPLE_MUSHROOM_CAP;
            }
            if ((lp.in && lp.dist <= radius) || (i < 6 && lp.ratio < 0.f && dist3(pos, p1) < radius)
                || (i < 5 && lp.ratio > 1.f && dist3(pos, p2) < radius)) {
                out = block;
                return true;
            }
        }
        return false;
    }
    case MMF_RAFFLESIA: {
        if (pos.y > 10.f || len3(pos) > 15.f) return false;
        pos = pos * 0.8f;
        v3 c = pos;
        c.y -= 1.f;
        c.y *= 1.4f;
        if (sd_sphere(c, 1.f) < 0.f) { out = MMB_RAFFLESIA_SPIKES; return true; }
        float sdf = __builtin_fabsf(sd_sphere(c - V3(0.f, 1.f, 0.f), 2.0f)) - 0.8f;
        const float hole = sd_sphere(c - V3(0.f, 1.8f, 0.f), 1.8f);
        sdf = __builtin_fmaxf(sdf, -hole);
        if (sdf < 0.f) { out = c.y > 1.f ? MMB_RAFFLESIA_CENTER : MMB_RAFFLESIA_STEM; return true; }

        const float startAngle = frng.u01() * MM_TWO_PI;
        for (int i = 0; i < 5; ++i) {
            const float petalAngle = startAngle + (((float)i * MM_TWO_PI) * 0.2f);
            float st, ct;
            sincos_(-petalAngle, st, ct);
            v3 p = V3(pos.x * ct + pos.z * st, pos.y - 3.2f, (-pos.x) * st + pos.z * ct);
            p.y -= (float)(i % 2) * 0.53f;
            p.y += clampf((__builtin_fabsf(p.x - 3.f) - 1.5f) / 1.5f, 0.f, 1.f) * 1.3f;
            p.x -= 3.8f;
            p.z *= 1.2f;
            if (sd_capped_cylinder(p, 2.5f, 0.5f) < 0.f) { out = MMB_RAFFLESIA_PETAL; return true; }
        }
        return false;
    }
    case MMF_LARGE_JUNGLE_TREE: {
        const float height = 18.f + 10.f * frng.u01();
        if (pos.y > height + 6.f || len2(pos.x, pos.z) > 15.f) return false;
        const int tx = (int)__builtin_floorf(pos.x), tz = (int)__builtin_floorf(pos.z);
        if (in_range_f(pos.y, 0.f, height) && tx >= 0 && tx <= 1 && tz >= 0 && tz <= 1) { out = MMB_JUNGLE_WOOD; return true; }
        pos = pos - V3(0.5f, 0.f, 0.5f);
        v3 lp = pos;
        lp.y -= (height - 2.f);
        if (jungle_leaves(lp, 4.f, 4.f, 7.f, frng.u01())) { out = brng.u01() < 0.5f ? MMB_JUNGLE_LEAVES_FRUITS : MMB_JUNGLE_LEAVES_PLAIN; return true; }

        const float numBranches = 0.5f + 2.5f * frng.u01();
        float bh = height;
        for (int i = 0; (float)i < numBranches; ++i) {
            bh -= (8.f + frng.u01() * 3.f) * (height / 30.f);
            const float angle = MM_TWO_PI * frng.u01();
            const v3 bs = V3(0.f, bh, 0.f);
            v3 be = V3(0.f, 0.f, 0.f);
            sincos_(-angle, be.z, be.x);
            be = ((3.f + 1.5f * frng.u01()) * be) + bs;
            be.y += 1.f + 1.5f * frng.u01();
            const LineParams l = line_params(pos, bs, be);
            const float br = 1.2f - (0.4f * l.ratio);
            if (l.in && l.dist < br) { out = MMB_JUNGLE_WOOD; return true; }
            lp = (pos - be) + V3(0.f, 0.2f, 0.f);
            if (jungle_leaves(lp, 2.f, 2.5f, 3.5f, frng.u01())) { out = brng.u01() < 0.25f ? MMB_JUNGLE_LEAVES_FRUITS : MMB_JUNGLE_LEAVES_PLAIN; return true; }
        }
        return false;
    }
    case MMF_SMALL_JUNGLE_TREE: {
        const float height = 8.f + 4.f * frng.u01();
        const float maxDist = pos.y < height - 2.f ? 2.f : 8.f;
        if (pos.y > height + 4.f || len2(pos.x, pos.z) > maxDist) return false;
        if (in_range_f(pos.y, 0.f, height) && (int)__builtin_floorf(pos.x) == 0 && (int)__builtin_floorf(pos.z) == 0) { out = MMB_JUNGLE_WOOD; return true; }
        if (jungle_leaves(pos - V3(0.f, height - 1.f, 0.f), 3.f, 2.f, 4.f, frng.u01())) {
            out = brng.u01() < 0.25f ? MMB_JUNGLE_LEAVES_FRUITS : MMB_JUNGLE_LEAVES_PLAIN;
            return true;
        }
        return false;
    }
    case MMF_TINY_JUNGLE_TREE: {
        if (fp.x + fp.y + fp.z > 8) return false;
        const int height = (int)(0.5f + 2.5f * frng.u01());
        if (fp.x == 0 && in_range_i(fp.y, 0, height) && fp.z == 0) { out = MMB_JUNGLE_WOOD; return true; }
        if (iabs(fp.x) + iabs(fp.y - height) + iabs(fp.z) == 1) { out = MMB_JUNGLE_LEAVES_PLAIN; return true; }
        return false;
    }
    case MMF_CACTUS: {
        if (iabs(fp.x) > 5 || iabs(fp.z) > 5) return false;
        const int height = (int)(7.5f + frng.u01() * 6.0f);
        if (pos.y > (float)height + 2.f) return false;
        if (fp.x == 0 && in_range_i(fp.y, 0, height) && fp.z == 0) { out = MMB_CACTUS; return true; }
        for (int arm = 0; arm < 4; ++arm) {
            if (frng.u01() >= 0.35f) continue;
            const int armStart = (int)(4.f + frng.u01() * (float)(height - 10));
            const int armLength = (int)(2.f + frng.u01() * 1.f);
            int armHeight = (int)(3.f + frng.u01() * 3.f);
            armHeight = imin(height - armStart - 1, armHeight);
            const int dx = kDirX[arm * 2], dz = kDirZ[arm * 2];
            const int x2 = dx * armLength, z2 = dz * armLength;       // armPos2 = (x2, armStart, z2)
            const bool seg1 = fp.x >= imin(0, x2) && fp.x <= imax(0, x2) && fp.y == armStart && fp.z >= imin(0, z2) && fp.z <= imax(0, z2);
            const int y3 = armStart + armHeight;
            const bool seg2 = fp.x == x2 && fp.z == z2 && fp.y >= imin(armStart, y3) && fp.y <= imax(armStart, y3);
            if (seg1 || seg2) { out = MMB_CACTUS; return true; }
        }
        return false;
    }
    case MMF_PALM_TREE: {
        if (fp.y < -2 || fp.y > 28 || iabs(fp.x) + iabs(fp.z) > 24) return false;
        v3 mn = V3(0.f, 0.f, 0.f), mx = V3(0.f, 0.f, 0.f);
        v3 ctrl[4];
        v3 cur = V3(0.f, 0.f, 0.f);
        ctrl[0] = cur;
#pragma unroll
        for (int i = 1; i < 4; ++i) {
            const float scale = 1.f + ((float)i / 4) * 5.f;
            const float r0 = u11(frng), r1 = frng.u01(), r2 = u11(frng);
            cur = cur + V3(scale * r0, 3.f + 5.f * r1, scale * r2);
            ctrl[i] = cur;
            mn = V3(gmin(mn.x, cur.x), gmin(mn.y, cur.y), gmin(mn.z, cur.z));
            mx = V3(gmax(mx.x, cur.x), gmax(mx.y, cur.y), gmax(mx.z, cur.z));
        }
        {
            const v3 c1 = mn - V3(7.f, 1.f, 7.f), c2 = mx + V3(7.f, 6.f, 7.f);
            const v3 lo = V3(gmin(c1.x, c2.x), gmin(c1.y, c2.y), gmin(c1.z, c2.z)), hi = V3(gmax(c1.x, c2.x), gmax(c1.y, c2.y), gmax(c1.z, c2.z));
            if (!(pos.x >= lo.x && pos.x <= hi.x && pos.y >= lo.y && pos.y <= hi.y && pos.z >= lo.z && pos.z <= hi.z)) return false;
        }
        v3 spline[5];
        de_casteljau<4, 5>(ctrl, spline);
        const v3 tt = floor3(spline[4]);
        const i3 top = {(int)tt.x, (int)tt.y, (int)tt.z};
        const i3 lpos = {fp.x - top.x, fp.y - top.y, fp.z - top.z};
        float ld = len2((float)lpos.x, (float)lpos.z);
        ld *= (0.6f + (0.3f * clampf((float)(20 - top.y) * 0.05f, 0.f, 1.f))) + (0.3f * frng.u01());
        if (in_range_i(lpos.y, -1, 0) && ld < 3.9f && (lpos.x == 0 || lpos.z == 0 || iabs(lpos.x) == iabs(lpos.z))) {
            const int lh = ld > 3.f ? -1 : 0;
            if (lpos.y == lh) { out = MMB_PALM_LEAVES; return true; }
        }
        for (int i = 0; i < 4; ++i) {
            v3 p1 = spline[i], p2 = spline[i + 1];
            const v3 pad = norm3(p2 - p1) * 0.5f;
            if (i > 0) p1 = p1 - pad;
            if (i + 1 < 4) p2 = p2 + pad;
            if (in_rasterized_line(fp, p1, p2)) { out = MMB_PALM_WOOD; return true; }
        }
        return false;
    }
    case MMF_MEDIUM_CRYSTAL:
    case MMF_CRYSTAL: {
        if (fy > 180) return false;
        pos = pos + V3(0.f, 2.f, 0.f);
        pos = pos * (0.55f + 0.4f * frng.u01());
        if (feature == MMF_MEDIUM_CRYSTAL) pos = pos * 2.f;
        if (imax(iabs(fp.x), iabs(fp.z)) > 25) return false;
        const float r0 = u11(frng), r1 = frng.u01(), r2 = u11(frng);
        const v3 endPos = V3(12.f * r0, 18.f + 8.f * r1, 12.f * r2);
        if (pos.y > endPos.y + 2.f) return false;
        const uint8_t block = random_crystal_block(frng.u01());
        if (in_crystal(pos, V3(0.f, 0.f, 0.f), endPos, 4.f + 1.2f * frng.u01())) { out = block; return true; }
        pos = pos * 0.8f;
        const int numSmall = (int)(4.f + 2.f * frng.u01());
        float angle = frng.u01() * MM_TWO_PI;
        for (int i = 0; i < numSmall; ++i) {
            angle += MM_PI_OVER_TWO + MM_PI * frng.u01();
            v3 e = V3(0.f, 0.f, 0.f);
            sincos_(angle, e.z, e.x);
            e = e * (6.f + 3.f * frng.u01());
            e.y = 7.f + 5.f * frng.u01();
            if (in_crystal(pos, V3(0.f, 0.f, 0.f), e, 1.5f + 1.5f * frng.u01())) { out = block; return true; }
        }
        return false;
    }
    default: return false;
    }
}

// =========================================================================================================
// cave features
// =========================================================================================================
MM_DEV bool place_cave_feature(int feature, int fx, int fy, int fz, int layerHeight, int wx, int wy, int wz, uint32_t fstate, uint8_t& out)
{
    const i3 fp = {wx - fx, wy - fy, wz - fz};
    const i3 ftp = {wx - fx, wy - (fy + layerHeight), wz - fz};
    const v3 pos = V3((float)fp.x, (float)fp.y, (float)fp.z);
    v3 topPos = V3((float)ftp.x, (float)ftp.y, (float)ftp.z);
    MinStd frng; frng.x = fstate;                 // = rng4(fx, fy, fz, 398132) (featurePlacement.hpp:1119)
    LazyRng brng = {wx, wy, wz, 9322743};        // the per-voxel stream (featurePlacement.hpp:1120): seeded only where a rule draws from it

    switch (feature) {
    case MMCF_TEST_GLOWSTONE_PILLAR:
        if (fp.x == 0 && fp.z == 0 && in_range_i(fp.y, 0, layerHeight)) { out = MMB_GLOWSTONE; return true; }
        return false;
    case MMCF_TEST_SHROOMLIGHT_PILLAR:
        if (fp.x == 0 && fp.z == 0 && in_range_i(fp.y, 0, layerHeight)) { out = MMB_SHROOMLIGHT; return true; }
        return false;
    case MMCF_CAVE_VINE: {
        if (ftp.x != 0 || ftp.z != 0) return false;
        int height = (int)(3.f + 12.f * frng.u01());
        height = imin(height, layerHeight);
        if (!in_range_i(ftp.y, -height, 0)) return false;
        const bool glowing = brng.u01() < 0.2f;
        if (ftp.y == -height) out = glowing ? MMB_CAVE_VINES_GLOW_END : MMB_CAVE_VINES_END;
        else out = glowing ? MMB_CAVE_VINES_GLOW_MAIN : MMB_CAVE_VINES_MAIN;
        return true;
    }
    case MMCF_GLOWSTONE_CLUSTER: {
        topPos.y *= 1.35f;
        topPos = topPos * (1.f + 0.5f * frng.u01());
        const float r = len3(topPos);
        if (r > 6.f) return false;
        const float a = atan2f_(pos.z, pos.x);
        const float maxR = 3.5f + 2.f * simplex2(a * 1.5f, (float)wy * 1.5f);
        if (r < maxR) { out = MMB_GLOWSTONE; return true; }
        return false;
    }
    case MMCF_STORMLIGHT_SPHERE:
    case MMCF_CEILING_STORMLIGHT_SPHERE: {
        const float radius = 3.5f + 4.f * frng.u01();
        const float d = feature == MMCF_STORMLIGHT_SPHERE ? len3(pos) : len3(topPos);
        if (d > radius) return false;
        const float chance = smoothstep(0.4f, 0.2f, d / radius);
        if (brng.u01() < chance) out = MMB_GLOWSTONE;
        else out = random_crystal_block(frng.u01());
        return true;
    }
    case MMCF_CRYSTAL_PILLAR: {
        if (pos.y < -8.f || topPos.y > 8.f) return false;
        float d = len2(pos.x, pos.z);
        if (d > 7.f) return false;
        float hr = pos.y / (float)layerHeight;
        if (hr < 0.f) { hr = 0.f; d = len3(pos); }
        else if (hr > 1.f) { hr = 1.f; d = len3(topPos); }
        float radius = hr - 0.5f;
        radius = 4.f * (2.f * radius * radius + 0.5f);
        if (d > radius) return false;
        if (d / radius < 0.4f) out = MMB_GLOWSTONE;
        else out = random_crystal_block(frng.u01());
        return true;
    }
    case MMCF_WARPED_FUNGUS: {
        const int ml = iabs(fp.x) + iabs(fp.z);
        if (ml > 6) return false;
        const int height = (int)(2.5f + 3.0f * frng.u01());
        if (fp.y < -2 || fp.y > height + 3) return false;
        if (fp.x == 0 && fp.z == 0 && in_range_i(fp.y, 0, height)) { out = MMB_WARPED_STEM; return true; }
        const int sh = fp.y - (height - 1);
        if (in_range_i(sh, 0, 1) && ml == 1) {
            if (brng.u01() < (sh == 0 ? 0.2f : 0.5f)) { out = MMB_SHROOMLIGHT; return true; }
        }
        const float capRadius = len2(pos.x, pos.z);
        if (capRadius > 3.7f) return false;
        const int capEnd = height + 1 - (int)(capRadius / 2.5f);
        const float s = simplex2(((float)wx + (float)fy) * 3.f, ((float)wz + (float)fy) * 3.f);
        const int capStart = (int)((float)capEnd - ((4.2f * s) * __builtin_fmaxf(capRadius - 2.3f, 0.f)));
        if (in_range_i(fp.y, capStart, capEnd)) { out = MMB_WARPED_WART; return true; }
        return false;
    }
    case MMCF_AMBER_FUNGUS: {
        const int ml = iabs(fp.x) + iabs(fp.z);
        if (ml > 4) return false;
        const int height = (int)(4.5f + 4.5f * frng.u01());
        if (fp.y < -2 || fp.y > height + 3) return false;
        if (fp.x == 0 && fp.z == 0) {
            if (in_range_i(fp.y, 0, height)) { out = MMB_AMBER_STEM; return true; }
            else if (fp.y == height + 1) { out = MMB_AMBER_WART; return true; }
        }
        int capStart = height / 2;
        if (simplex2((float)wx, (float)wz) < 0.f) capStart -= 1;
        if (in_range_i(fp.y, capStart, height)) {
            const int capDist = (fp.y - capStart) < (height / 4 + 1) ? 2 : 1;
            if (ml == capDist) {
                const int cx = (wx / 2) * 2, cy = (wy / 2) * 2, cz = (wz / 2) * 2;
                const f3 r = rand3from3((float)cx, (float)cy, (float)cz);
                const bool isGrid = wx == cx + (int)(r.x * 2.f) && wy == cy + (int)(r.y * 2.f) && wz == cz + (int)(r.z * 2.f);
                if (isGrid && brng.u01() < 0.65f) out = MMB_SHROOMLIGHT;
                else out = MMB_AMBER_WART;
                return true;
            }
        }
        return false;
    }
    default: return false;
    }
}

}  // namespace mm
