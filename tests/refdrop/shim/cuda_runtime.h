// tests/refdrop: the CUDA runtime NAMES the reference's terrain.cpp / cudaUtils.hpp / structs.hpp spell, mapped onto HIP.  A name map, not
// an implementation: every call lands in the real HIP runtime of this image.  Only what those three files use.
#pragma once
#include <hip/hip_runtime.h>
typedef hipStream_t cudaStream_t;
typedef hipError_t cudaError_t;
typedef hipArray_t cudaArray_t;
typedef hipTextureObject_t cudaTextureObject_t;
#define cudaSuccess hipSuccess
#define cudaMalloc hipMalloc
#define cudaMallocHost hipHostMalloc
#define cudaFree hipFree
#define cudaFreeHost hipHostFree
#define cudaMemcpy hipMemcpy
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaStreamCreate hipStreamCreate
#define cudaStreamDestroy hipStreamDestroy
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaGetLastError hipGetLastError
#define cudaGetErrorName hipGetErrorName
#define cudaGetErrorString hipGetErrorString
