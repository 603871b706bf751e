// tests/refdrop: the reference's cuda/cudaUtils.hpp includes <thrust/random.h> for chunk.cu's sake; terrain.cpp and the host Chunk use
// nothing of it.  rocThrust's header only parses under the HIP compiler, and compiling terrain.cpp as HIP would put HIP's global
// non-constexpr min(int, int) in front of glm::min in its constexpr initialisers (terrain.cpp:115-126) - so these host translation
// units get an empty header.  (The engine itself is pinned against the real rocThrust in oracle/thrust_probe.cpp.)
#pragma once
