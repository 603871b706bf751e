// tests/refdrop: driver-API names used by cudaUtils.hpp / common.h (CUBuffer::dev_ptr, CU_CHECK), mapped onto HIP
#pragma once
#include "cuda_runtime.h"
typedef hipDeviceptr_t CUdeviceptr;
typedef hipError_t CUresult;
typedef hipCtx_t CUcontext;
#define CUDA_SUCCESS hipSuccess
#define cuGetErrorName hipDrvGetErrorName
#define cuGetErrorString hipDrvGetErrorString
