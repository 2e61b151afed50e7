// TEST INFRASTRUCTURE ONLY (oracle/build_ref.py): force-included when hipcc compiles the reference's UPSNet nms_kernel.cu
// (mmdet/models/utils/upsnet/nms/nms_kernel.cu, raw CUDA runtime API, no ATen) from its original location. It maps the CUDA
// runtime names that file uses onto the HIP runtime; nothing in the product includes it.
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaGetDevice hipGetDevice
#define cudaSetDevice hipSetDevice
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaMemcpy hipMemcpy
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
