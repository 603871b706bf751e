// oracle/_ref probe: the REAL vendored glm 0.9.9.8 of the reference (external/include/glm), compiled where it lies under
// /root/reference with plain g++ (header-only, no stand-ins needed).  It pins the oracle's restatement of glm::simplex
// (gtc/noise.inl:591-721) and of the glm helpers (smoothstep, mix, mod, fract, normalize, length, dot) bit for bit.
// Built only in the build container (the GPU box has no /root/reference); its outputs are frozen as tests/golden/glm_probe.npz
// by tests/golden/make_golden.py.
#include <glm/glm.hpp>
#include <glm/gtc/noise.hpp>
#include <glm/gtx/vector_angle.hpp>
extern "C" {
void ref_simplex2(int n, const float* xy, float* out) { for (int i = 0; i < n; ++i) out[i] = glm::simplex(glm::vec2(xy[2 * i], xy[2 * i + 1])); }
void ref_simplex3(int n, const float* xyz, float* out) { for (int i = 0; i < n; ++i) out[i] = glm::simplex(glm::vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])); }
void ref_smoothstep(int n, const float* e0e1x, float* out) { for (int i = 0; i < n; ++i) out[i] = glm::smoothstep(e0e1x[3 * i], e0e1x[3 * i + 1], e0e1x[3 * i + 2]); }
void ref_mix(int n, const float* xya, float* out) { for (int i = 0; i < n; ++i) out[i] = glm::mix(xya[3 * i], xya[3 * i + 1], xya[3 * i + 2]); }
void ref_mod(int n, const float* ab, float* out) { for (int i = 0; i < n; ++i) out[i] = glm::mod(ab[2 * i], ab[2 * i + 1]); }
void ref_fract(int n, const float* a, float* out) { for (int i = 0; i < n; ++i) out[i] = glm::fract(a[i]); }
void ref_normalize3(int n, const float* xyz, float* out3)
{
    for (int i = 0; i < n; ++i) { glm::vec3 r = glm::normalize(glm::vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])); out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z; }
}
void ref_length3(int n, const float* xyz, float* out) { for (int i = 0; i < n; ++i) out[i] = glm::length(glm::vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])); }
void ref_distance3(int n, const float* a, const float* b, float* out)
{
    for (int i = 0; i < n; ++i) out[i] = glm::distance(glm::vec3(a[3 * i], a[3 * i + 1], a[3 * i + 2]), glm::vec3(b[3 * i], b[3 * i + 1], b[3 * i + 2]));
}
void ref_cross3(int n, const float* a, const float* b, float* out3)
{
    for (int i = 0; i < n; ++i) { glm::vec3 r = glm::cross(glm::vec3(a[3 * i], a[3 * i + 1], a[3 * i + 2]), glm::vec3(b[3 * i], b[3 * i + 1], b[3 * i + 2])); out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z; }
}
float ref_radians(float d) { return glm::radians(d); }
}
