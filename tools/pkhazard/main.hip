// Reproducer / probe for the packed-FP32 problem of DESIGN.md 3.3 (developer tool, not part of the library).
//
//   cd tools/pkhazard
//   hipcc --offload-arch=gfx950 -O3 -c pk.hip -o pk.o                                                      # SLP-vectorised: v_pk_*_f32
//   hipcc --offload-arch=gfx950 -O3 -Xclang -target-feature -Xclang -packed-fp32-ops -c nopk.hip -o nopk.o  # no packed FP32
//   hipcc --offload-arch=gfx950 -O3 -c main.hip -o main.o && hipcc --offload-arch=gfx950 main.o pk.o nopk.o -o pkhazard.bin && ./pkhazard.bin
//
// Both objects hold the SAME source (stage.h = the interpolation arithmetic of flow_stage_s_kernel). Each is run 20 times on
// fixed inputs (a) alone and (b) while a second stream keeps the CUs busy with a register- and LDS-heavy MFMA kernel (the
// footprint of the conv kernels: 2 waves per SIMD), and the outputs of the runs are compared bit for bit with the first run.
// Observation that led here (inside the frame graph, MI355X, ROCm 7.2): the packed build returned a different fx for ~0.05 % of
// the pixels in (b), never in (a); the build without packed FP32 never differed. This program reproduces it stand-alone
// (profiles/r02_pkhazard_reproducer.txt); patterns.hip narrows it down to the instruction form (op_sel = 1 on src1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" __global__ void stage_kernel_pk(const float*, const float*, int, int, int, float, float*);
extern "C" __global__ void stage_kernel_nopk(const float*, const float*, int, int, int, float, float*);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// matrix-pipe load with the conv kernels' footprint: 512 threads, ~200 VGPRs worth of accumulators, 128 KB of LDS per block
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

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename K>
static long run_case(K kernel, const char* name, bool loaded, const float* x6, const float* flo, float* out, float* sink, int H, int W,
                     hipStream_t s_main, hipStream_t s_side) {
    const size_t n = (size_t)H * W * 12;
    std::vector<float> first(n), cur(n);
    long diffs = 0;
    for (int rep = 0; rep < 20; ++rep) {
        if (loaded) hipLaunchKernelGGL(mfma_load_kernel, dim3(256), dim3(512), 128 * 1024, s_main, 60000, sink);
        // a little delay so that the stage kernel starts while the load kernel is resident
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(kernel, dim3(4096), dim3(256), 0, s_side, x6, flo, 4, H, W, 20.0f, out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(rep ? cur.data() : first.data(), out, n * sizeof(float), hipMemcpyDeviceToHost));
        if (rep) for (size_t i = 0; i < n; ++i) diffs += memcmp(&cur[i], &first[i], 4) != 0;
    }
    printf("%-28s %-22s elements differing from the first run over 19 repeats: %ld\n", name, loaded ? "beside the MFMA kernel" : "alone", diffs);
    return diffs;
}

int main() {
    const int H = 1024, W = 2048;
    float *x6, *flo, *out, *sink;
    CK(hipMalloc(&x6, (size_t)H * W * 8 * 4)); CK(hipMalloc(&flo, (size_t)(H / 4) * (W / 4) * 4 * 4));
    CK(hipMalloc(&out, (size_t)H * W * 12 * 4)); CK(hipMalloc(&sink, 64));
    std::vector<float> h((size_t)H * W * 8);
    srand(1);
    for (auto& v : h) v = (float)(rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(x6, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> f((size_t)(H / 4) * (W / 4) * 4);
    for (auto& v : f) v = (float)(rand() % 2001 - 1000) * 2e-4f;
    CK(hipMemcpy(flo, f.data(), f.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_load_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipStream_t s_main, s_side;
    CK(hipStreamCreate(&s_main)); CK(hipStreamCreate(&s_side));
    long bad = 0;
    bad += run_case(stage_kernel_nopk, "no packed FP32", false, x6, flo, out, sink, H, W, s_main, s_side);
    bad += run_case(stage_kernel_nopk, "no packed FP32", true, x6, flo, out, sink, H, W, s_main, s_side);
    run_case(stage_kernel_pk, "packed FP32 (SLP)", false, x6, flo, out, sink, H, W, s_main, s_side);
    run_case(stage_kernel_pk, "packed FP32 (SLP)", true, x6, flo, out, sink, H, W, s_main, s_side);
    return bad ? 1 : 0;
}
