// Pipe-overlap micro-benchmarks for gfx950 (developer tool, not part of the product library).
//   hipcc --offload-arch=gfx950 -O3 tools/pipebench.hip -o gpurun_out/pipebench && gpurun_out/pipebench
// One 512-thread workgroup per CU: waves 0-3 take role A, waves 4-7 role B, so every SIMD hosts one wave of each role.
// Reports the time of each role alone and of both together (sum => the pipes serialize, max => they overlap), plus the
// per-CU global-load bandwidth from an L2-resident buffer for the access shapes the conv kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum Role { NONE = 0, MFMA = 1, VALU = 2, LOAD_LIN = 3, LOAD_ROWS = 4, LDSR = 5, LOAD_HALF = 6, MIX_VALU = 7, MIX_LDS = 8, MFMA16 = 9 };

__device__ __forceinline__ float role_mfma(int iters, int lane) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    return s;
}

// same FLOPs per iteration as role_mfma with the 16x16x32 shape (4 passes, 4 accumulator VGPRs per instruction: half
// the accumulator write-back per FLOP of 32x32x16)
__device__ __forceinline__ float role_mfma16(int iters, int lane) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    return s;
}

// one wave doing both: 48 MFMAs per iteration with 5 independent VALU FMAs (or one ds_read_b128 per 4 MFMAs) between them
template <int KIND>
__device__ __forceinline__ float role_mix(const __bf16* lds, int iters, int lane) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    float x[5];
    for (int i = 0; i < 5; ++i) x[i] = (float)(lane + i);
    float s = 0.f;
    const int row = lane & 31, sw = (row >> 2) & 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            bf16x8 v;
            if (KIND == 1) {
                const unsigned ad = (unsigned)(size_t)lds + (((u & 3) * 32 + row) * 32 + ((((lane >> 5) + 2 * (u >> 2 & 1)) ^ sw) << 3)) * 2;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(ad));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                if (KIND == 0) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) x[j] = __builtin_fmaf(x[j], 1.0001f, 0.5f);
                }
            }
            if (KIND == 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); s += (float)v[0]; }
        }
    }
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 5; ++i) s += x[i];
    return s;
}

__device__ __forceinline__ float role_valu(int iters, int lane) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(lane + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 30; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    return s;
}

// shape 0: one instruction = 1 KB contiguous (64 lanes x 16 B). shape 1: 8 rows x 128 B, rows 1 KB apart (a 256-channel
// NHWC pixel row per 8 lanes). shape 2: 16 rows x 64 B, rows 4.5 KB apart (weight rows of a k-major panel)
__device__ __forceinline__ float role_load(const float* __restrict__ buf, size_t nfloat, int iters, int lane, int wave, int cu, int shape) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    size_t lane_off, step;   // in floats, multiples of 4
    if (shape == 0) { lane_off = (size_t)lane * 4; step = 256; }
    else if (shape == 1) { lane_off = (size_t)(lane >> 3) * 256 + (lane & 7) * 4; step = 8 * 256; }
    else { lane_off = (size_t)(lane >> 2) * 1152 + (lane & 3) * 4; step = 16 * 1152; }
    const unsigned mask = (unsigned)nfloat - 1;   // nfloat is a power of two
    unsigned pos = (unsigned)(((size_t)cu * 7919 + wave * 131) * step) & mask;
    for (int it = 0; it < iters; ++it) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v[u] = *reinterpret_cast<const f32x4*>(buf + ((pos + (unsigned)lane_off) & mask));
            pos += (unsigned)step;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    return s[0] + s[1] + s[2] + s[3];
}

// 12 ds_read_b128 per iteration with the conv kernels' fragment addressing (64-byte rows, XOR-swizzled 16-byte chunks).
// Inline asm: the compiler removed the plain (even volatile) vector reads of the first version of this role.
__device__ __forceinline__ float role_lds(const __bf16* lds, int iters, int lane, int wave) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const int row = lane & 31, sw = (row >> 2) & 3;
    const unsigned base = (unsigned)(size_t)lds;
    unsigned addr[12];
    for (int u = 0; u < 12; ++u) addr[u] = base + (((u & 3) * 32 + row) * 32 + ((((lane >> 5) + 2 * (u >> 2 & 1)) ^ sw) << 3)) * 2;
    for (int it = 0; it < iters; ++it) {
        f32x4 v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(addr[u]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 12; ++u) s += v[u];
    }
    return s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(512) void pipe_kernel(int roleA, int roleB, int itA, int itB, const float* buf, size_t nfloat, float* sink, int prioB) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[128 * 32 * 3];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 128 * 32 * 3; i += 512) lds[i] = (__bf16)(float)(i & 7);
    __syncthreads();
    const int role = wave < 4 ? roleA : roleB;
    if (wave >= 4 && prioB) __builtin_amdgcn_s_setprio(3);
    const int iters = wave < 4 ? itA : itB;
    float r = 0.f;
    if (role == MFMA) r = role_mfma(iters, lane);
    else if (role == VALU) r = role_valu(iters, lane);
    else if (role == LOAD_LIN) r = role_load(buf, nfloat, iters, lane, wave, blockIdx.x, 0);
    else if (role == LOAD_ROWS) r = role_load(buf, nfloat, iters, lane, wave, blockIdx.x, 1);
    else if (role == LOAD_HALF) r = role_load(buf, nfloat, iters, lane, wave, blockIdx.x, 2);
    else if (role == LDSR) r = role_lds(lds, iters, lane, wave);
    else if (role == MFMA16) r = role_mfma16(iters, lane);
    else if (role == MIX_VALU) r = role_mix<0>(lds, iters, lane);
    else if (role == MIX_LDS) r = role_mix<1>(lds, iters, lane);
    if (r == 12345.678f) sink[0] = r;
}

static int g_prio = 0;
static float run(int roleA, int roleB, int itA, int itB, const float* buf, size_t nfloat, float* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(pipe_kernel, dim3(256), dim3(512), 0, 0, roleA, roleB, itA, itB, buf, nfloat, sink, g_prio);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(pipe_kernel, dim3(256), dim3(512), 0, 0, roleA, roleB, itA, itB, buf, nfloat, sink, g_prio);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const size_t nfloat = (size_t)2 << 20;   // 8 MB: L2/MALL resident
    float *buf, *sink;
    hipMalloc(&buf, nfloat * 4); hipMalloc(&sink, 64);
    hipMemset(buf, 0, nfloat * 4);
    const char* names[] = {"none", "mfma", "valu", "load-1KB", "load-8x128B", "lds-read", "load-16x64B", "mfma+valu", "mfma+lds", "mfma16x16x32"};
    const int IT_M = 2000, IT_V = 1000, IT_L = 1500, IT_S = 6000;
    auto its = [&](int role) { return (role == MFMA || role >= MIX_VALU) ? IT_M : role == VALU ? IT_V : role == LDSR ? IT_S : IT_L; };
    auto report = [&](int a, int b) {
        const float ta = a ? run(a, NONE, its(a), 0, buf, nfloat, sink) : 0.f;
        const float tb = b ? run(NONE, b, 0, its(b), buf, nfloat, sink) : 0.f;
        const float tab = run(a, b, its(a), its(b), buf, nfloat, sink);
        printf("A=%-12s B=%-12s  A alone %.3f ms  B alone %.3f ms  both %.3f ms  (sum %.3f, max %.3f)\n", names[a], names[b], ta, tb, tab,
               ta + tb, ta > tb ? ta : tb);
    };
    // absolute rates
    {
        const float t = run(MFMA, MFMA, IT_M, IT_M, buf, nfloat, sink);
        const double fl = 256.0 * 8 * IT_M * 48 * 2.0 * 32 * 32 * 16;
        printf("mfma x8 waves/CU: %.3f ms  %.1f TFLOP/s bf16\n", t, fl / t / 1e9);
        const float t4 = run(MFMA, NONE, IT_M, 0, buf, nfloat, sink);
        printf("mfma x4 waves/CU: %.3f ms  %.1f TFLOP/s bf16 (one wave per SIMD)\n", t4, fl / 2 / t4 / 1e9);
    }
    for (int shape : {LOAD_LIN, LOAD_ROWS, LOAD_HALF}) {
        const float t8 = run(shape, shape, IT_L, IT_L, buf, nfloat, sink);
        const double bytes = 256.0 * 8 * IT_L * 8 * 1024;
        const float t4 = run(NONE, shape, 0, IT_L, buf, nfloat, sink);
        printf("%-12s 8 waves/CU: %.3f ms %.2f TB/s (%.1f B/clk/CU @2.1GHz) | 4 waves/CU: %.3f ms %.2f TB/s\n", names[shape], t8, bytes / t8 / 1e9,
               bytes / t8 / 1e-3 / 256 / 2.1e9, t4, bytes / 2 / t4 / 1e9);
    }
    {
        const float t8 = run(LDSR, LDSR, IT_S, IT_S, buf, nfloat, sink);
        const double bytes = 256.0 * 8 * IT_S * 12 * 1024;
        printf("lds ds_read_b128 8 waves/CU: %.3f ms  %.1f B/clk/CU @2.1GHz\n", t8, bytes / t8 / 1e-3 / 256 / 2.1e9);
    }
    report(MFMA, VALU);
    report(MFMA, LOAD_LIN);
    report(MFMA, LOAD_ROWS);
    report(MFMA, LDSR);
    report(VALU, LOAD_LIN);
    report(VALU, LDSR);
    report(LDSR, LOAD_LIN);
    printf("-- 16x16x32 shape (same FLOPs per iteration)\n");
    report(MFMA16, NONE);
    report(MFMA16, VALU);
    report(MFMA16, LOAD_LIN);
    report(MFMA16, LDSR);
    report(MFMA16, MFMA16);
    printf("-- same wave interleaved (A only): compare with mfma alone\n");
    report(MIX_VALU, NONE);
    report(MIX_LDS, NONE);
    printf("-- role B at s_setprio(3)\n");
    g_prio = 1;
    report(MFMA, VALU);
    report(MFMA, LOAD_LIN);
    report(MFMA, LDSR);
    report(MIX_LDS, LOAD_LIN);
    return 0;
}
