// Stride-1 correlation (LiteFlowNetCorr: 9 x 9 = 81 displacements, 256 channels) on the vector ALU with HALF a wavefront per dot product
// (round 6). Reference: mmdet/ops/correlation (correlation_cuda_kernel.cu: correlation_forward, kernel_size 1, stride1 1), as
// correlation4_kernel in flow_ops.hip, which this replaces where it applies; entered through vpsi_launch_corr4h from vps_correlation.
//
// correlation4_kernel gives a dot product of 256 channels to the 64 lanes of a wavefront: 4 multiply-adds per lane and output, then a
// 6-step reduction - per displacement row 144 multiply-adds among ~690 wave instructions. A SIMD issues one vector instruction per 4
// cycles whatever the occupancy, so the kernel runs at its instruction count: 273 us for 2.7 G multiply-adds (the HBM floor is 40 us).
// Here 32 lanes own a dot product (two float4 per lane: 8 multiply-adds per lane and output) and the two halves of a wavefront work on
// two displacement ROWS of the same four pixels: the folding butterfly (bit selects, not `?:` on the register array - that becomes an
// indexed extract) reduces both rows at once over 5 lane bits and leaves two neighbouring pixels of one displacement per lane.
// Per pair of rows: 24 loads, 36 x 8 multiply-adds, ~190 reduction instructions: ~750 instead of ~1380 (the library is built without
// packed FP32: DESIGN.md 3.3).
#include "common.h"
#include "conv_common.h"

namespace {

__device__ __forceinline__ float bit_select(const unsigned m, const float a, const float b) {   // m all ones: a, zero: b
    return __uint_as_float((m & __float_as_uint(a)) | (~m & __float_as_uint(b)));
}

template <int R, int NQ>
__global__ __launch_bounds__(256)
void correlation4h_kernel(const float* __restrict__ in1, int ld1, int coff1, const float* __restrict__ in2, int ld2, int coff2,
                          float* __restrict__ out, int out_ld, int out_coff, int N, int H, int W, int C, int act, float slope) {
    constexpr int D = 2 * R + 1, NCOL = D + 3, NSLOT = 4 * D, NIT = (D + 1) / 2;
    static_assert(NSLOT <= 64 && NSLOT > 32, "slots (displacement, pixel) of one row: folded from 64");
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const int wave = threadIdx.x >> 6;
    const int gpr = W >> 2;                                  // 4-pixel groups per row
    const int ngroups = N * H * gpr;
    const int c4n = C >> 2;
    const float invC = 1.0f / (float)C;
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(in1, (unsigned)((size_t)N * H * W * ld1 * sizeof(float)));
    const __amdgpu_buffer_rsrc_t r2 = make_rsrc(in2, (unsigned)((size_t)N * H * W * ld2 * sizeof(float)));
    unsigned choff[NQ];                                      // byte offset of the lane's channels inside a pixel, or "beyond the buffer"
    bool chok[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { chok[q] = hl + 32 * q < c4n; choff[q] = 16u * (unsigned)(hl + 32 * q); }
    // the two slots this lane holds after the folds: slot = 4 * displacement + pixel
    int mine = 0;
#pragma unroll
    for (int s = 0; s < 5; ++s) mine += (lane >> (4 - s) & 1) ? 32 >> s : 0;
    const int my_ti = mine >> 2, my_p = mine & 3;            // my_p is 0 or 2: pixels my_p, my_p + 1

    const int gstep = (int)gridDim.x * 4;
    int grp = (int)blockIdx.x * 4 + wave;
    int g = grp % gpr, y, n;
    { const int t = grp / gpr; y = t % H; n = t / H; }
    const int dq = gstep / gpr, dr = gstep - dq * gpr;
    for (; grp < ngroups; grp += gstep) {
        const int xb = 4 * g;
        const unsigned pix = (unsigned)((n * H + y) * W + xb);
        f32x4 a[4][NQ];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                a[p][q] = buffer_load16<f32x4>(r1, chok[q] ? (pix + p) * (unsigned)ld1 * 4u + (unsigned)coff1 * 4u + choff[q] : 0xFFFFFFF0u, 0u);
#pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
            const int tj = 2 * it + half - R;                // this half's displacement row
            const int y2 = y + tj;
            const bool rowok = tj <= R && (unsigned)y2 < (unsigned)H;
            const unsigned row2 = (unsigned)((n * H + y2) * W) * (unsigned)ld2 * 4u + (unsigned)coff2 * 4u;
            float v[64];
#pragma unroll
            for (int i = NSLOT; i < 64; ++i) v[i] = 0.f;
            // column uu of in2 feeds slot (ti = uu - p, p) for the p with 0 <= ti < D: every slot is ONE dot product
#pragma unroll
            for (int cb = 0; cb < NCOL; cb += 6) {
                f32x4 b[6][NQ];
#pragma unroll
                for (int du = 0; du < 6; ++du) {
                    const int uu = cb + du;
                    if (uu >= NCOL) continue;
                    const int x2 = xb + uu - R;
                    const bool ok = rowok && (unsigned)x2 < (unsigned)W;
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        b[du][q] = buffer_load16<f32x4>(r2, (ok && chok[q]) ? row2 + (unsigned)x2 * (unsigned)ld2 * 4u + choff[q] : 0xFFFFFFF0u, 0u);
                }
                __builtin_amdgcn_sched_barrier(0);           // keep the six columns in flight together
#pragma unroll
                for (int du = 0; du < 6; ++du) {
                    const int uu = cb + du;
                    if (uu >= NCOL) continue;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int ti = uu - p;
                        if (ti < 0 || ti >= D) continue;
                        float acc = a[p][0][0] * b[du][0][0];
                        acc = __builtin_fmaf(a[p][0][1], b[du][0][1], acc);
                        acc = __builtin_fmaf(a[p][0][2], b[du][0][2], acc);
                        acc = __builtin_fmaf(a[p][0][3], b[du][0][3], acc);
#pragma unroll
                        for (int q = 1; q < NQ; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc = __builtin_fmaf(a[p][q][e], b[du][q][e], acc);
                        v[ti * 4 + p] = acc;
                    }
                }
            }
            // fold over the 5 lane bits of a half, high bit first: every exchange halves the number of slots a lane carries (64 -> 2).
            // Slots >= NSLOT are padding: in the first step a pair (i, i + 32) with i + 32 >= NSLOT is one plain exchange - the upper
            // lanes then carry garbage in padding slots, which never meets a real slot (a fold adds the SAME slot of two lanes)
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const int off = 16 >> s, hf = 32 >> s;
                const unsigned up = (lane & off) ? 0xFFFFFFFFu : 0u;
#pragma unroll
                for (int i = 0; i < hf; ++i) {
                    if (s == 0 && i + hf >= NSLOT) { v[i] += __shfl_xor(v[i], off, 64); continue; }
                    const float send = bit_select(up, v[i], v[i + hf]);
                    const float keep = bit_select(up, v[i + hf], v[i]);
                    v[i] = keep + __shfl_xor(send, off, 64);
                }
            }
            if (tj <= R && my_ti < D) {
                float* __restrict__ o = out + ((size_t)pix + my_p) * out_ld + out_coff + (tj + R) * D + my_ti;
                o[0] = vps_act(v[0] * invC, act, slope);
                o[out_ld] = vps_act(v[1] * invC, act, slope);
            }
        }
        g += dr; y += dq;
        if (g >= gpr) { g -= gpr; ++y; }
        while (y >= H) { y -= H; ++n; }
    }
}

}  // namespace

// true when the layer was launched here: stride2 1, W % 4 == 0, C <= 256 (two float4 per lane), buffers below 4 GB
__attribute__((visibility("hidden")))
bool vpsi_launch_corr4h(const float* in1, int ld1, int coff1, const float* in2, int ld2, int coff2, float* out, int out_ld, int out_coff,
                        int N, int H, int W, int C, int r, int stride2, int act, float slope, hipStream_t s) {
    const char* e = getenv("VPS_CORR_HALF");                 // VPS_CORR_HALF=0: correlation4_kernel (A/B runs)
    if (e && e[0] == '0') return false;
    if (stride2 != 1 || r != 4 || (W & 3) || C > 256 || (C & 3)) return false;
    if ((size_t)N * H * W * ld1 * 4 >= 0xFFFFFFF0ull || (size_t)N * H * W * ld2 * 4 >= 0xFFFFFFF0ull || (long)N * H * (W / 4) >= 0x7fffffffL) return false;
    long g4 = ((long)N * H * (W / 4) + 3) / 4; if (g4 > 2048) g4 = 2048;       // 8 blocks of 4 waves per CU: one resident round
    if (C <= 128) hipLaunchKernelGGL((correlation4h_kernel<4, 1>), dim3((unsigned)g4), dim3(256), 0, s, in1, ld1, coff1, in2, ld2, coff2, out, out_ld, out_coff, N, H, W, C, act, slope);
    else hipLaunchKernelGGL((correlation4h_kernel<4, 2>), dim3((unsigned)g4), dim3(256), 0, s, in1, ld1, coff1, in2, ld2, coff2, out, out_ld, out_coff, N, H, W, C, act, slope);
    return true;
}
