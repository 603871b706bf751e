// tests/refdrop: the OptiX type NAMES that rendering/structs.hpp and util/common.h mention (the ray tracer is outside the generation
// path, SURVEY §8: out of scope).  Nothing here is called.
#pragma once
typedef unsigned long long OptixTraversableHandle;
typedef int OptixResult;
#define OPTIX_SUCCESS 0
inline const char* optixGetErrorName(OptixResult) { return "optix (not built)"; }
inline const char* optixGetErrorString(OptixResult) { return "optix (not built)"; }
