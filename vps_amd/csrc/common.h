// Shared helpers for the libvpship kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vps_hip.h"

#define VPS_EARG(x) (-1000 - (x))

static inline int vps_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// grid size for grid-stride HBM-bound kernels: cap at 256 CUs x 8 blocks
static inline int stream_grid(long work_items, int block) {
    long g = (work_items + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ float vps_act(float v, int act, float slope) {
    if (act == VPS_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == VPS_ACT_LEAKY) return v > 0.f ? v : v * slope;
    return v;
}
