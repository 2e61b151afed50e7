// Shared body of the reproducer kernel (tools/pkhazard/README in main.hip): the interpolation arithmetic of
// vps_amd/csrc/flow_ops.hip:flow_stage_s_kernel, the kernel in which the problem was found. Compiled twice, with and
// without packed-FP32 instructions, under two names (STAGE_KERNEL).
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float tap(float v00, float v01, float v10, float v11, float alpha, float beta) {
    return (float)((1.0 - alpha) * (1.0 - beta) * v00 + alpha * (1.0 - beta) * v01 + (1.0 - alpha) * beta * v10 + alpha * beta * v11);
}

extern "C" __global__ __launch_bounds__(256) void STAGE_KERNEL(const float* __restrict__ x6, const float* __restrict__ flo, int flo_ld, int H, int W,
                                                               float mul, float* __restrict__ out) {
    const int Hl = H >> 2, Wl = W >> 2;
    const long HW = (long)H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < HW; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W), y = (int)(idx / W);
        float sy = 0.25f * ((float)y + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
        float sx = 0.25f * ((float)x + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
        const int y0 = (int)sy, x0 = (int)sx;
        const int yp = y0 < Hl - 1 ? 1 : 0, xp = x0 < Wl - 1 ? 1 : 0;
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* p00 = flo + ((size_t)y0 * Wl + x0) * flo_ld;
        const float* p01 = p00 + (size_t)xp * flo_ld;
        const float* p10 = p00 + (size_t)yp * Wl * flo_ld;
        const float* p11 = p10 + (size_t)xp * flo_ld;
        const float fx = hy * (hx * (p00[0] * mul) + lx * (p01[0] * mul)) + ly * (hx * (p10[0] * mul) + lx * (p11[0] * mul));
        const float fy = hy * (hx * (p00[1] * mul) + lx * (p01[1] * mul)) + ly * (hx * (p10[1] * mul) + lx * (p11[1] * mul));
        const f32x4 a = *reinterpret_cast<const f32x4*>(x6 + (size_t)idx * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(x6 + (size_t)idx * 8 + 4);
        const float xf = (float)x + fx, yf = (float)y + fy;
        const float alpha = xf - floorf(xf), beta = yf - floorf(yf);
        const int xL = max(min((int)floorf(xf), W - 1), 0), xR = max(min((int)(floorf(xf) + 1.f), W - 1), 0);
        const int yT = max(min((int)floorf(yf), H - 1), 0), yB = max(min((int)(floorf(yf) + 1.f), H - 1), 0);
        const float* tl = x6 + ((size_t)yT * W + xL) * 8 + 3;
        const float* tr = x6 + ((size_t)yT * W + xR) * 8 + 3;
        const float* bl = x6 + ((size_t)yB * W + xL) * 8 + 3;
        const float* br = x6 + ((size_t)yB * W + xR) * 8 + 3;
        float wv[3], nrm = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            wv[c] = tap(tl[c], tr[c], bl[c], br[c], alpha, beta);
            const float df = a[c] - wv[c];
            nrm += df * df;
        }
        float* o = out + (size_t)idx * 12;
        const f32x4 o1 = {b[0], b[1], wv[0], wv[1]};
        const f32x4 o2 = {wv[2], fx / mul, fy / mul, sqrtf(nrm)};
        *reinterpret_cast<f32x4*>(o) = a;
        *reinterpret_cast<f32x4*>(o + 4) = o1;
        *reinterpret_cast<f32x4*>(o + 8) = o2;
    }
}
