// How fast can a wave write a pixel-major [M][64] fp32 map, depending on which 16 bytes each lane of a store instruction owns?
//   A  the conv epilogue's pattern (transposed 32x32 accumulators): lane (pixel p = l & 31, half h = l >> 5), instruction (b, g) writes
//      16 B at channel 32 b + 8 g + 4 h of pixel p: 32 B contiguous per pixel, 32 pixels (= 32 cache lines of 128 B) per instruction
//   B  row-contiguous: instruction i writes 16 B at channel quad l & 15 of pixel 4 i + (l >> 4): 256 B contiguous per pixel, 1 KB per
//      instruction (what the epilogue could do after a transpose of the accumulators through LDS)
//   C  like B but 128 B segments: quad l & 7 of pixel 8 i + (l >> 3), for the 32-channel half b (two passes)
// hipcc --offload-arch=gfx950 -O3 storebench.hip -o storebench && ./storebench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ out, const int ntiles, const float seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // a tile = 256 pixels (8 x 32 patch in the real kernel; consecutive here), a wave owns 64 of them = 2 sub-tiles of 32
        float* base = out + ((size_t)tile * 256 + wave * 64) * 64;
        const f32x4 v = {seed, seed + lane, seed + tile, seed};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float* sb = base + (size_t)a * 32 * 64;
            if (MODE == 0) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4*>(sb + (size_t)(lane & 31) * 64 + 32 * b + 8 * g + 4 * (lane >> 5)) = v;
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(sb + (size_t)(4 * i + (lane >> 4)) * 64 + 4 * (lane & 15)) = v;
            } else {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(sb + (size_t)(8 * i + (lane >> 3)) * 64 + 32 * b + 4 * (lane & 7)) = v;
            }
        }
    }
}

// the conv kernels' real addressing: a tile = an 8 x 32 patch of a 1024 x 2048 map (rows 512 KB apart), a wave owns 2 patch rows;
// WAIT: every tile's stores are waited for (s_waitcnt vmcnt(0) + barrier) before the next tile, as a kernel does whose next loads
// sit behind the stores in the in-order vmcnt queue
template <int PATTERN, bool WAIT>
__global__ __launch_bounds__(256) void patch_kernel(float* __restrict__ out, const int ntiles, const float seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile & 63, ty = tile >> 6;
        const f32x4 v = {seed, seed + lane, seed + tile, seed};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float* sb = out + ((size_t)(ty * 8 + wave * 2 + a) * 2048 + tx * 32) * 64;
            if (PATTERN == 0) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4*>(sb + (size_t)(lane & 31) * 64 + 32 * b + 8 * g + 4 * (lane >> 5)) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(sb + (size_t)(4 * i + (lane >> 4)) * 64 + 4 * (lane & 15)) = v;
            }
        }
        if (WAIT) {
            __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
            __syncthreads();
        }
    }
}

template <int PATTERN, bool WAIT> void run_patch(float* out, int ntiles, const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {512, 1024, 8192}) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((patch_kernel<PATTERN, WAIT>), dim3(grid), dim3(256), 0, 0, out, ntiles, 1.f);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((patch_kernel<PATTERN, WAIT>), dim3(grid), dim3(256), 0, 0, out, ntiles, 1.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)ntiles * 256 * 64 * 4;
        printf("%s grid %5d: %7.1f us  %.2f TB/s\n", name, grid, ms * 50.0, bytes / (ms / 20 * 1e-3) / 1e12);
    }
}

template <int MODE> void run(float* out, int ntiles, const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {512, 1024, 2048, 8192}) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, ntiles, 1.f);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, ntiles, 1.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)ntiles * 256 * 64 * 4;
        printf("%s grid %5d: %7.1f us  %.2f TB/s\n", name, grid, ms * 50.0, bytes / (ms / 20 * 1e-3) / 1e12);
    }
}

int main() {
    const int ntiles = 8192;                       // 1024 x 2048 pixels x 64 channels = 537 MB
    float* out; hipMalloc(&out, (size_t)ntiles * 256 * 64 * 4);
    run<0>(out, ntiles, "A epilogue pattern (32 B per pixel and instruction)");
    run<1>(out, ntiles, "B row-contiguous   (256 B per pixel, 1 KB per instruction)");
    run<2>(out, ntiles, "C 128 B segments   (8 pixels per instruction)");
    run_patch<0, false>(out, ntiles, "D pattern A on 8x32 patches of a 1024x2048 map");
    run_patch<0, true>(out, ntiles, "E = D + wait for the stores after every tile");
    run_patch<1, false>(out, ntiles, "F pattern B (row-contiguous) on 8x32 patches");
    run_patch<1, true>(out, ntiles, "G = F + wait for the stores after every tile");
    return 0;
}
