// THIRD-PARTY PIN — test infrastructure only.  The reference draws its random numbers from thrust (src/util/rng.hpp:80-96:
// thrust::default_random_engine seeded with its integer hash; thrust::uniform_real_distribution<float> u01(0, 1) at every call site,
// e.g. biomeFuncs.hpp:205-207, chunk.cu:1053-1054).  Thrust is not vendored by the reference (CUDA toolkit, version unpinned); this image
// carries rocThrust 7.2 (/opt/rocm/include/thrust), the same library for HIP.  This file compiles the REAL thrust/random.h host-side
// (hipcc, no device code) and exposes the engine + distribution exactly as the reference composes them, so that golden seeds -> draws
// (tests/golden/thrust_probe.npz, made by tools/gen_thrust_probe.py) pin oracle/mmo_math.h's MinStd and the device's rng3 / rng4 by
// execution rather than by reading.  The integer hash is the reference's own (rng.hpp:69-78, six rounds), restated here because
// rng.hpp itself needs cuda headers.
#include <thrust/random.h>
#include <cstdint>

static inline unsigned int ref_hash(unsigned int a)
{
    a = (a + 0x7ed55d16) + (a << 12);
    a = (a ^ 0xc761c23c) ^ (a >> 19);
    a = (a + 0x165667b1) + (a << 5);
    a = (a + 0xd3a2646c) ^ (a << 9);
    a = (a + 0xfd7046c5) + (a << 3);
    a = (a ^ 0xb55a4f09) ^ (a >> 16);
    return a;
}

extern "C" {

// raw engine: seed -> n draws of u01 (the seed-0 -> 1 rule of linear_congruential_engine is inside thrust)
void thrust_minstd_u01(unsigned int seed, int n, float* out)
{
    thrust::default_random_engine rng(seed);
    thrust::uniform_real_distribution<float> u01(0, 1);
    for (int i = 0; i < n; ++i) out[i] = u01(rng);
}

// raw engine values (x_1 .. x_n) for the LCG itself
void thrust_minstd_raw(unsigned int seed, int n, unsigned int* out)
{
    thrust::default_random_engine rng(seed);
    for (int i = 0; i < n; ++i) out[i] = rng();
}

// makeSeededRandomEngine(x, y, z) / (x, y, z, w) (rng.hpp:86-96) + n draws of u01; has_w == 0 selects the 3-argument seeding
void thrust_seeded_u01(int x, int y, int z, int w, int has_w, int n, float* out, unsigned int* seed_out)
{
    // the shifts are done on unsigned values: the reference shifts ints (negative coordinates: implementation-defined but universally
    // two's complement); (1 << 31) is INT_MIN
    const unsigned ux = (unsigned)x, uy = (unsigned)y, uw = (unsigned)w;
    const unsigned h = has_w ? (ref_hash((1u << 31) | (ux << 22) | (uy << 11) | uw) ^ ref_hash((unsigned)z))
                             : (ref_hash((1u << 31) | (ux << 22) | uy) ^ ref_hash((unsigned)z));
    if (seed_out) *seed_out = h;
    thrust_minstd_u01(h, n, out);
}

// uniform_real_distribution<float>(a, b), used with other ranges by nothing on the path but cheap to pin: (x - 1) / 2^31 * (b - a) + a
void thrust_uniform(unsigned int seed, float a, float b, int n, float* out)
{
    thrust::default_random_engine rng(seed);
    thrust::uniform_real_distribution<float> d(a, b);
    for (int i = 0; i < n; ++i) out[i] = d(rng);
}

}  // extern "C"
