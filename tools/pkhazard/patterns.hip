// Which packed-FP32 instruction forms return wrong values beside an MFMA kernel? (second probe of tools/pkhazard, see main.hip)
//
//   hipcc --offload-arch=gfx950 -O3 patterns.hip -o patterns.bin && ./patterns.bin
//
// Every pattern executes ONE packed instruction form (pinned with inline asm) on lane- and iteration-dependent operands and compares
// both halves with scalar arithmetic done by ordinary (non-packed, also asm-pinned) instructions in the same thread; mismatches
// are counted per pattern, alone and while a second stream runs the MFMA load kernel. A count that is non-zero "alone" means the
// expectation coded here is wrong for that form (op_sel semantics), not a hazard; the signal is alone == 0, loaded > 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float s_mul(float a, float b) { float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float s_add(float a, float b) { float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float s_fma(float a, float b, float c) { float d; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ bool ne(float a, float b) { return __float_as_uint(a) != __float_as_uint(b); }

enum { P_MUL, P_ADD, P_FMA, P_FMA_OPSEL, P_MUL_OPSEL_HI0, P_MOV, P_MUL_AFTER_MOV, P_MUL_AFTER_LOAD, P_MUL_SGPR,
       P_FMA_S0, P_FMA_S1, P_FMA_S2, P_FMA_H0, P_FMA_H1, P_FMA_H2, P_MUL_S1, P_ADD_S1, P_MUL_H1, P_FMA_BCAST, NPAT };
static const char* NAMES[NPAT] = {"v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]",
                                  "v_pk_mul_f32 op_sel_hi:[1,0]", "v_pk_mov_b32 op_sel:[1,0]", "v_mov_b32 ; v_pk_mul_f32 (forwarded source)",
                                  "global_load_dwordx2 ; s_waitcnt ; v_pk_mul_f32", "v_pk_mul_f32 with an SGPR pair source",
                                  "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_fma_f32 op_sel:[0,0,1]",
                                  "v_pk_fma_f32 op_sel_hi:[0,1,1]", "v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_fma_f32 op_sel_hi:[1,1,0]",
                                  "v_pk_mul_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel:[0,1]", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]",
                                  "v_pk_fma_f32 op_sel_hi:[1,0,0] (broadcast lo)"};

__global__ __launch_bounds__(256) void pattern_kernel(int iters, const float* __restrict__ mem, float smul, unsigned* __restrict__ counts) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned bad[NPAT];
    for (int p = 0; p < NPAT; ++p) bad[p] = 0;
    const f32x2 sm = {smul, smul};
    for (int it = 0; it < iters; ++it) {
        const float u = (float)((t * 7 + it * 13) & 1023) * 0.03125f - 9.f, v = (float)((t * 3 + it * 29) & 511) * 0.0625f - 5.f;
        const f32x2 a = {u, v}, b = {v + 1.5f, u - 0.25f}, c = {u * 0.5f, v * 2.f};
        f32x2 d;
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
        bad[P_MUL] += ne(d[0], s_mul(a[0], b[0])) || ne(d[1], s_mul(a[1], b[1]));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
        bad[P_ADD] += ne(d[0], s_add(a[0], b[0])) || ne(d[1], s_add(a[1], b[1]));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        bad[P_FMA] += ne(d[0], s_fma(a[0], b[0], c[0])) || ne(d[1], s_fma(a[1], b[1], c[1]));
        // op_sel picks the half of each source that feeds the LOW result, op_sel_hi the half that feeds the HIGH result
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        bad[P_FMA_OPSEL] += ne(d[0], s_fma(a[0], b[1], c[0])) || ne(d[1], s_fma(a[1], b[0], c[1]));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        bad[P_MUL_OPSEL_HI0] += ne(d[0], s_mul(a[0], b[0])) || ne(d[1], s_mul(a[1], b[0]));
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        bad[P_MOV] += ne(d[0], a[1]) || ne(d[1], b[0]);
        {   // a plain VALU write immediately consumed by a packed op (one asm block: nothing can be scheduled in between)
            asm volatile("v_mov_b32 v200, %1\n\tv_mov_b32 v201, %2\n\ts_nop 4\n\tv_mov_b32 v200, %3\n\tv_pk_mul_f32 %0, v[200:201], %4"
                         : "=v"(d) : "v"(a[0]), "v"(a[1]), "v"(c[1]), "v"(b) : "v200", "v201");
            bad[P_MUL_AFTER_MOV] += ne(d[0], s_mul(c[1], b[0])) || ne(d[1], s_mul(a[1], b[1]));
        }
        {   // load -> wait -> packed op
            const float* p = mem + ((t * 2 + it * 2) & 8190);
            f32x2 l;
            asm volatile("global_load_dwordx2 %0, %2, off\n\ts_waitcnt vmcnt(0)\n\tv_pk_mul_f32 %1, %0, %3" : "=&v"(l), "=v"(d) : "v"(p), "v"(b) : "memory");
            bad[P_MUL_AFTER_LOAD] += ne(d[0], s_mul(p[0], b[0])) || ne(d[1], s_mul(p[1], b[1]));
        }
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "s"(sm), "v"(a));
        bad[P_MUL_SGPR] += ne(d[0], s_mul(smul, a[0])) || ne(d[1], s_mul(smul, a[1]));
#define FMA_FORM(P, MODS, L0, L1, L2, H0, H1, H2)                                                                        \
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 " MODS : "=v"(d) : "v"(a), "v"(b), "v"(c));                              \
        bad[P] += ne(d[0], s_fma(a[L0], b[L1], c[L2])) || ne(d[1], s_fma(a[H0], b[H1], c[H2]));
        FMA_FORM(P_FMA_S0, "op_sel:[1,0,0]", 1, 0, 0, 1, 1, 1)
        FMA_FORM(P_FMA_S1, "op_sel:[0,1,0]", 0, 1, 0, 1, 1, 1)
        FMA_FORM(P_FMA_S2, "op_sel:[0,0,1]", 0, 0, 1, 1, 1, 1)
        FMA_FORM(P_FMA_H0, "op_sel_hi:[0,1,1]", 0, 0, 0, 0, 1, 1)
        FMA_FORM(P_FMA_H1, "op_sel_hi:[1,0,1]", 0, 0, 0, 1, 0, 1)
        FMA_FORM(P_FMA_H2, "op_sel_hi:[1,1,0]", 0, 0, 0, 1, 1, 0)
        FMA_FORM(P_FMA_BCAST, "op_sel_hi:[1,0,0]", 0, 0, 0, 1, 0, 0)
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        bad[P_MUL_S1] += ne(d[0], s_mul(a[0], b[1])) || ne(d[1], s_mul(a[1], b[1]));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        bad[P_ADD_S1] += ne(d[0], s_add(a[0], b[1])) || ne(d[1], s_add(a[1], b[1]));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        bad[P_MUL_H1] += ne(d[0], s_mul(a[0], b[1])) || ne(d[1], s_mul(a[1], b[0]));
    }
    for (int p = 0; p < NPAT; ++p) if (bad[p]) atomicAdd(&counts[p], bad[p]);
}

__global__ __launch_bounds__(512) void mfma_load_kernel(int iters, float* sink) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32768; i += 512) lds[i] = (float)(i & 15);
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(float)((lane + e) & 7); b[e] = (_Float16)(float)((lane - e) & 3); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        if ((it & 63) == 0) a[0] = (_Float16)lds[(it + lane) & 32767];
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;
}

int main() {
    float *mem, *sink;
    unsigned* counts;
    CK(hipMalloc(&mem, 8192 * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&counts, NPAT * 4));
    float h[8192];
    for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 37) % 1000) * 0.01f - 4.f;
    CK(hipMemcpy(mem, h, sizeof(h), hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_load_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipStream_t s_main, s_side;
    CK(hipStreamCreate(&s_main)); CK(hipStreamCreate(&s_side));
    for (int loaded = 0; loaded < 2; ++loaded) {
        CK(hipMemset(counts, 0, NPAT * 4));
        for (int rep = 0; rep < 10; ++rep) {
            if (loaded) hipLaunchKernelGGL(mfma_load_kernel, dim3(256), dim3(512), 128 * 1024, s_main, 60000, sink);
            for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(pattern_kernel, dim3(2048), dim3(256), 0, s_side, 400, mem, 1.25f, counts);
            CK(hipDeviceSynchronize());
        }
        unsigned c[NPAT];
        CK(hipMemcpy(c, counts, sizeof(c), hipMemcpyDeviceToHost));
        printf("%s: executions per pattern %.3g\n", loaded ? "beside the MFMA kernel" : "alone", 10.0 * 4 * 2048 * 256 * 400);
        for (int p = 0; p < NPAT; ++p) printf("  %-52s mismatching executions %u\n", NAMES[p], c[p]);
    }
    return 0;
}
