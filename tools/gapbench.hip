// How many non-MFMA instructions fit between two back-to-back v_mfma_f32_32x32x16_bf16 of one wave for free? (developer tool)
//   hipcc --offload-arch=gfx950 -O3 tools/gapbench.hip -o tools/gapbench.bin && tools/gapbench.bin
// MI355X_MICROARCH.md: "<= 5 single-issue instructions hidden per 32x32x16 bf16 MFMA gap (32 cyc/SIMD)". This sweeps the
// number of fillers per gap (pinned with inline asm, 4 independent accumulators), for three filler kinds (independent
// v_fma_f32, ds_read_b128, global_load_dwordx4 from an L2-resident line) with one and two waves per SIMD, and prints the
// shader cycles per MFMA per SIMD (s_memtime) — the budget the conv kernels' interleaving has to respect.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NF, int KIND>
__global__ __launch_bounds__(512) void gap_kernel(int iters, const float* __restrict__ buf, float* sink, long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < 4096; i += blockDim.x) lds[i] = (float)(i & 7);
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(lane + i);
    const float c1 = 1.0001f, c2 = 0.5f;
    const unsigned laddr = (unsigned)(size_t)lds + (unsigned)(lane * 16);
    const float* gp = buf + (size_t)(blockIdx.x % 64) * 4096 + lane * 4;
    f32x4 ld[8];
    for (int i = 0; i < 8; ++i) ld[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long t0 = wall_clock64();
    const long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(i * NF + j) & 7]) : "v"(c1), "v"(c2));
                    else if (KIND == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[(i * NF + j) & 7]) : "v"(laddr));
                    else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[(i * NF + j) & 7]) : "v"(gp));
                }
            }
        if (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long c1c = __builtin_readcyclecounter();
    const long long t1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += x[i] + ld[i][0];
    if (s == 12345.678f) sink[0] = s;
    if (t == 0 && blockIdx.x == 0) { cyc[0] = c1c - c0; cyc[1] = t1 - t0; }
}

template <int NF, int KIND>
static void run(int threads, const float* buf, float* sink, long long* cyc, const char* kind) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    gap_kernel<NF, KIND><<<256, threads>>>(10, buf, sink, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    gap_kernel<NF, KIND><<<256, threads>>>(iters, buf, sink, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double nm = (double)iters * 32;                       // MFMAs per wave
    const int wps = threads / 256;                              // waves per SIMD
    const double tf = nm * (threads / 64) * 256 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    printf("%-12s fillers/gap %d  waves/SIMD %d : %7.3f ms  %7.1f TFLOP/s  %6.1f cyc/MFMA/wave  => %5.1f cyc/MFMA/SIMD  (clock %.2f GHz)\n", kind, NF, wps, ms, tf,
           (double)h[0] / nm, (double)h[0] / nm / wps, (double)h[0] / (ms * 1e6));
}

template <int KIND>
static void sweep(const float* buf, float* sink, long long* cyc, const char* kind) {
    for (int threads : {256, 512}) {
        run<0, KIND>(threads, buf, sink, cyc, kind); run<1, KIND>(threads, buf, sink, cyc, kind); run<2, KIND>(threads, buf, sink, cyc, kind);
        run<3, KIND>(threads, buf, sink, cyc, kind); run<4, KIND>(threads, buf, sink, cyc, kind); run<5, KIND>(threads, buf, sink, cyc, kind);
        run<6, KIND>(threads, buf, sink, cyc, kind); run<8, KIND>(threads, buf, sink, cyc, kind); run<12, KIND>(threads, buf, sink, cyc, kind);
    }
}

int main() {
    float *buf, *sink;
    long long* cyc;
    hipMalloc(&buf, 64 * 4096 * sizeof(float) + 4096);
    hipMemset(buf, 0, 64 * 4096 * sizeof(float) + 4096);
    hipMalloc(&sink, 64);
    hipMalloc(&cyc, 64);
    sweep<0>(buf, sink, cyc, "v_fma_f32");
    sweep<1>(buf, sink, cyc, "ds_read_b128");
    sweep<2>(buf, sink, cyc, "global_load");
    return 0;
}
