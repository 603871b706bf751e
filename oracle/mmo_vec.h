// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// Minimal scalar restatement of the glm 0.9.9.8 vector semantics the reference relies on
// (reference: external/include/glm/detail/func_geometric.inl:25-65 (dot), :8-23 (length,
// distance), :82-88 (normalize); func_common.inl:104-111 (mix), :212-218 (mod),
// :249-255 (step), :564-570 (smoothstep), :505-509 (clamp)).
// Every operation is spelled out component by component in glm's evaluation order so that
// g++ -O2 -ffp-contract=off produces the same fp32 bits as real glm (checked by
// oracle/ref_glm_probe.cpp against the vendored glm headers).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

namespace mmo {

// Like glm's, the constructors take any arithmetic type per component (vec3(0, trunkHeight, 0), vec2(worldBlockPos.x, leavesSeed)) and
// convert it with a static_cast; vectors of another component type convert implicitly (glm declares those constructors explicit only
// under GLM_FORCE_EXPLICIT_CTOR, which the reference does not define: `vec3 pos = floorPos;`, isInRasterizedLine(pos, ...) with a
// float position truncating to ivec3).  With these the oracle can state the reference's expressions as they are written.
struct vec2;
struct vec3;
struct ivec2 {
    int x, y;
    int& operator[](int i) { return i == 0 ? x : y; }
    const int& operator[](int i) const { return i == 0 ? x : y; }
    ivec2() : x(0), y(0) {}
    template <class A, class = std::enable_if_t<std::is_arithmetic<A>::value>> explicit ivec2(A s) : x((int)s), y((int)s) {}
    template <class A, class B> ivec2(A x_, B y_) : x((int)x_), y((int)y_) {}
    inline ivec2(const vec2& v);
};
struct ivec3 {
    int x, y, z;
    ivec3() : x(0), y(0), z(0) {}
    template <class A, class = std::enable_if_t<std::is_arithmetic<A>::value>> explicit ivec3(A s) : x((int)s), y((int)s), z((int)s) {}
    template <class A, class B, class C> ivec3(A x_, B y_, C z_) : x((int)x_), y((int)y_), z((int)z_) {}
    inline ivec3(const vec3& v);
};

struct vec2 {
    float x, y;
    vec2() : x(0), y(0) {}
    template <class A, class = std::enable_if_t<std::is_arithmetic<A>::value>> explicit vec2(A s) : x((float)s), y((float)s) {}
    template <class A, class B> vec2(A x_, B y_) : x((float)x_), y((float)y_) {}
    vec2(ivec2 v) : x((float)v.x), y((float)v.y) {}
};
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    template <class A, class = std::enable_if_t<std::is_arithmetic<A>::value>> explicit vec3(A s) : x((float)s), y((float)s), z((float)s) {}
    template <class A, class B, class C> vec3(A x_, B y_, C z_) : x((float)x_), y((float)y_), z((float)z_) {}
    template <class C> vec3(vec2 v, C z_) : x(v.x), y(v.y), z((float)z_) {}
    template <class C> vec3(ivec2 v, C z_) : x((float)v.x), y((float)v.y), z((float)z_) {}
    vec3(ivec3 v) : x((float)v.x), y((float)v.y), z((float)v.z) {}
};
inline ivec2::ivec2(const vec2& v) : x((int)v.x), y((int)v.y) {}
inline ivec3::ivec3(const vec3& v) : x((int)v.x), y((int)v.y), z((int)v.z) {}
struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
};

// ---- ivec helpers
static inline ivec2 operator+(ivec2 a, ivec2 b) { return {a.x + b.x, a.y + b.y}; }
static inline ivec2 operator-(ivec2 a, ivec2 b) { return {a.x - b.x, a.y - b.y}; }
static inline ivec2 operator*(ivec2 a, int s) { return {a.x * s, a.y * s}; }
static inline bool operator==(ivec2 a, ivec2 b) { return a.x == b.x && a.y == b.y; }
static inline ivec3 operator+(ivec3 a, ivec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline ivec3 operator-(ivec3 a, ivec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline ivec3 operator*(ivec3 a, int s) { return {a.x * s, a.y * s, a.z * s}; }
static inline ivec3 operator/(ivec3 a, int s) { return {a.x / s, a.y / s, a.z / s}; }
static inline bool operator==(ivec3 a, ivec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
static inline ivec2 g_clamp(ivec2 v, int lo, int hi) { return {(v.x < lo) ? lo : ((hi < v.x) ? hi : v.x), (v.y < lo) ? lo : ((hi < v.y) ? hi : v.y)}; }   // glm::clamp = min(max(x, lo), hi)
static inline ivec2 g_abs(ivec2 v) { return {std::abs(v.x), std::abs(v.y)}; }
static inline ivec3 g_abs(ivec3 v) { return {std::abs(v.x), std::abs(v.y), std::abs(v.z)}; }
static inline int compAdd(ivec2 v) { return v.x + v.y; }                  // glm/gtx/component_wise.inl
static inline int compAdd(ivec3 v) { return v.x + v.y + v.z; }
static inline ivec3 g_min(ivec3 a, ivec3 b) { return {(b.x < a.x) ? b.x : a.x, (b.y < a.y) ? b.y : a.y, (b.z < a.z) ? b.z : a.z}; }
static inline ivec3 g_max(ivec3 a, ivec3 b) { return {(a.x < b.x) ? b.x : a.x, (a.y < b.y) ? b.y : a.y, (a.z < b.z) ? b.z : a.z}; }

// ---- vec2
static inline vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }
static inline vec2 operator-(vec2 a, vec2 b) { return vec2(a.x - b.x, a.y - b.y); }
static inline vec2 operator*(vec2 a, vec2 b) { return vec2(a.x * b.x, a.y * b.y); }
static inline vec2 operator/(vec2 a, vec2 b) { return vec2(a.x / b.x, a.y / b.y); }
static inline vec2 operator+(vec2 a, float s) { return vec2(a.x + s, a.y + s); }
static inline vec2 operator-(vec2 a, float s) { return vec2(a.x - s, a.y - s); }
static inline vec2 operator*(vec2 a, float s) { return vec2(a.x * s, a.y * s); }
static inline vec2 operator*(float s, vec2 a) { return vec2(s * a.x, s * a.y); }
static inline vec2 operator/(vec2 a, float s) { return vec2(a.x / s, a.y / s); }
static inline vec2& operator*=(vec2& a, float s) { a.x *= s; a.y *= s; return a; }
static inline vec2& operator+=(vec2& a, vec2 b) { a.x += b.x; a.y += b.y; return a; }

// ---- vec3
static inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 operator/(vec3 a, vec3 b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline vec3 operator+(vec3 a, float s) { return vec3(a.x + s, a.y + s, a.z + s); }
static inline vec3 operator-(vec3 a, float s) { return vec3(a.x - s, a.y - s, a.z - s); }
static inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
static inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
static inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
static inline vec3 operator-(float s, vec3 a) { return vec3(s - a.x, s - a.y, s - a.z); }
static inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
static inline vec3& operator*=(vec3& a, float s) { a.x *= s; a.y *= s; a.z *= s; return a; }
static inline vec3& operator*=(vec3& a, vec3 b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; return a; }
static inline vec3& operator+=(vec3& a, vec3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
static inline vec3& operator-=(vec3& a, vec3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }

// ---- vec4
static inline vec4 operator+(vec4 a, vec4 b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline vec4 operator-(vec4 a, vec4 b) { return vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline vec4 operator*(vec4 a, vec4 b) { return vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline vec4 operator+(vec4 a, float s) { return vec4(a.x + s, a.y + s, a.z + s, a.w + s); }
static inline vec4 operator-(vec4 a, float s) { return vec4(a.x - s, a.y - s, a.z - s, a.w - s); }
static inline vec4 operator*(vec4 a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline vec4 operator*(float s, vec4 a) { return vec4(s * a.x, s * a.y, s * a.z, s * a.w); }
static inline vec4 operator-(float s, vec4 a) { return vec4(s - a.x, s - a.y, s - a.z, s - a.w); }
static inline vec4 operator-(vec4 a) { return vec4(-a.x, -a.y, -a.z, -a.w); }

// ---- scalar glm functions
static inline float g_min(float a, float b) { return (b < a) ? b : a; }   // glm::min
static inline float g_max(float a, float b) { return (a < b) ? b : a; }   // glm::max
static inline int g_min(int a, int b) { return (b < a) ? b : a; }
static inline int g_max(int a, int b) { return (a < b) ? b : a; }
static inline float g_clamp(float x, float lo, float hi) { return g_min(g_max(x, lo), hi); }
static inline int g_clamp(int x, int lo, int hi) { return g_min(g_max(x, lo), hi); }
static inline float g_fract(float x) { return x - floorf(x); }
static inline float g_mod(float a, float b) { return a - b * floorf(a / b); }
static inline float g_step(float edge, float x) { return (x < edge) ? 0.f : 1.f; }
static inline float g_mix(float x, float y, float a) { return x * (1.f - a) + y * a; }
static inline float g_smoothstep(float e0, float e1, float x) {
    float t = g_clamp((x - e0) / (e1 - e0), 0.f, 1.f);
    return t * t * (3.f - 2.f * t);
}
static inline float g_radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// ---- vector glm functions
static inline vec2 g_floor(vec2 v) { return vec2(floorf(v.x), floorf(v.y)); }
static inline vec3 g_floor(vec3 v) { return vec3(floorf(v.x), floorf(v.y), floorf(v.z)); }
static inline vec4 g_floor(vec4 v) { return vec4(floorf(v.x), floorf(v.y), floorf(v.z), floorf(v.w)); }
static inline vec2 g_ceil(vec2 v) { return vec2(ceilf(v.x), ceilf(v.y)); }
static inline vec3 g_ceil(vec3 v) { return vec3(ceilf(v.x), ceilf(v.y), ceilf(v.z)); }
static inline vec2 g_fract(vec2 v) { return vec2(g_fract(v.x), g_fract(v.y)); }
static inline vec3 g_fract(vec3 v) { return vec3(g_fract(v.x), g_fract(v.y), g_fract(v.z)); }
static inline vec2 g_abs(vec2 v) { return vec2(fabsf(v.x), fabsf(v.y)); }
static inline vec3 g_abs(vec3 v) { return vec3(fabsf(v.x), fabsf(v.y), fabsf(v.z)); }
static inline vec4 g_abs(vec4 v) { return vec4(fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)); }
static inline vec2 g_max(vec2 a, vec2 b) { return vec2(g_max(a.x, b.x), g_max(a.y, b.y)); }
static inline vec3 g_min(vec3 a, vec3 b) { return vec3(g_min(a.x, b.x), g_min(a.y, b.y), g_min(a.z, b.z)); }
static inline vec3 g_max(vec3 a, vec3 b) { return vec3(g_max(a.x, b.x), g_max(a.y, b.y), g_max(a.z, b.z)); }
static inline vec4 g_max(vec4 a, vec4 b) { return vec4(g_max(a.x, b.x), g_max(a.y, b.y), g_max(a.z, b.z), g_max(a.w, b.w)); }
static inline vec2 g_mod(vec2 a, float b) { return vec2(g_mod(a.x, b), g_mod(a.y, b)); }
static inline vec3 g_step(vec3 edge, vec3 x) { return vec3(g_step(edge.x, x.x), g_step(edge.y, x.y), g_step(edge.z, x.z)); }
static inline vec4 g_step(vec4 edge, vec4 x) { return vec4(g_step(edge.x, x.x), g_step(edge.y, x.y), g_step(edge.z, x.z), g_step(edge.w, x.w)); }
static inline vec3 g_mix(vec3 x, vec3 y, float a) { return x * (1.f - a) + y * a; }
static inline vec2 g_mix(vec2 x, vec2 y, float a) { return x * (1.f - a) + y * a; }

static inline float g_dot(vec2 a, vec2 b) { vec2 t = a * b; return t.x + t.y; }
static inline float g_dot(vec3 a, vec3 b) { vec3 t = a * b; return t.x + t.y + t.z; }
static inline float g_dot(vec4 a, vec4 b) { vec4 t = a * b; return (t.x + t.y) + (t.z + t.w); }
static inline float g_length(vec2 v) { return sqrtf(g_dot(v, v)); }
static inline float g_length(vec3 v) { return sqrtf(g_dot(v, v)); }
static inline float g_distance(vec2 a, vec2 b) { return g_length(b - a); }
static inline float g_distance(vec3 a, vec3 b) { return g_length(b - a); }
static inline vec3 g_normalize(vec3 v) { return v * (1.f / sqrtf(g_dot(v, v))); }
static inline vec2 g_normalize(vec2 v) { return v * (1.f / sqrtf(g_dot(v, v))); }
static inline vec3 g_cross(vec3 x, vec3 y) {
    return vec3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}

static inline ivec2 to_ivec2(vec2 v) { return {(int)v.x, (int)v.y}; }
static inline ivec3 to_ivec3(vec3 v) { return {(int)v.x, (int)v.y, (int)v.z}; }

// C++ leaves the evaluation order of function-call arguments unspecified, and the reference draws from one random stream in several
// arguments of one constructor call (featurePlacement.hpp:224,700,991,1066).  The canonical order is left to right; a
// braced initialiser list guarantees it.  tools/extract_ref_literals.py reads `vec3_ltr` as `vec3`.
#define vec3_ltr(...) vec3{__VA_ARGS__}

}  // namespace mmo
