// Correlation on the matrix cores (round 5): the two configurations of the path - FlowNetC (441 channels: stride-2 displacements, radius
// 10, @128x256) and LiteFlowNetCorr (81 channels: stride 1, radius 4, @256x512), C = 256 - as BANDED GRAM MATRICES in split fp16.
// Reference: correlation_cuda_kernel.cu:46-147 (kernel_size 1, stride1 1):
//     out[y][x][tj * D + ti] = 1/C * sum_c in1[y][x][c] * in2[y + (tj-R)*S2][x + (ti-R)*S2][c]        (zero outside the image)
// In plane coordinates (x = S2*u + parity: stride-2 displacements only pair equal column parities) a displacement ROW tj of an output
// row y is the band  G[u][u + ti - R]  of the Gram matrix  G = A B^T,  A = in1 row y [positions x C], B = in2 row y + (tj-R)*S2.
// A workgroup owns 64 plane positions of one (image, row, parity): its A segment is split into fp16 pairs ONCE and stays in LDS for
// all D displacement rows; the B rows are streamed through LDS in 32-channel chunks (loads four chunks ahead, one barrier per chunk).
// Wave w computes one 32 x 32 tile: A positions [32 mi, 32 mi + 32) x B positions [32 ni - 16, 32 ni + 16) relative to the segment,
// (mi, ni) = (w >> 1, (w >> 1) + (w & 1)) - the four tiles that contain the whole band for R <= 16.
// Arithmetic: a = h0 + 2^-11 h1, b = k0 + 2^-11 k1 (fp16 pairs with a scaled residual, 22 significand bits: split_act of conv_common.h);
//     a*b ~ h0 k0 + 2^-11 (h0 k1 + h1 k0)       two fp32 accumulators, three v_mfma_f32_32x32x16_f16 per 16 channels
// (the dropped h1 k1 term is 2^-22 relative). Operands beyond the fp16 range are reported through `status` (bit 0) like the f16x3
// convolutions: the caller repeats the frame with the exact vector-ALU kernels (flow_ops.hip).
// After the channel loop the tile goes through LDS once: the band entries of a pixel sit at other accumulator indices in every lane.
#include "conv_common.h"

namespace {

template <int S2, int R, int NCH>
__global__ __launch_bounds__(256)
void corr_mfma_kernel(const float* __restrict__ in1, int ld1, int coff1, const float* __restrict__ in2, int ld2, int coff2,
                      float* __restrict__ out, int out_ld, int out_coff, int N, int H, int W, int act, float slope,
                      int* __restrict__ status) {
    constexpr int MODE = VPS_PREC_F16X3;
    typedef _Float16 h16;
    typedef vec8<h16> x8;
    typedef vec4<h16> x4;
    constexpr int D = 2 * R + 1;
    constexpr int UW = 64;                      // plane positions (A rows) of a workgroup
    constexpr int BWR = 96;                     // B positions staged per row: [u0 - 16, u0 + 80)
    // B chunks in flight in registers (NCH % PD == 0: static slots). One block per CU and one wave per SIMD: nothing hides a memory
    // round trip but the loads already in flight - with 4 chunks ahead a step took ~1400 cycles = latency / 4 (226 / 315 us for the
    // two shapes); a whole displacement row ahead (8 chunks, 96 VGPRs) keeps 24 x 16 B per lane in flight
    constexpr int PD = 8;
    static_assert(R <= 16 && NCH % PD == 0, "band inside the four tiles; static register slots");
    constexpr int OP = D + 1;                   // pitch of the output staging rows (floats)
    __shared__ __attribute__((aligned(16))) h16 As[NCH * 2 * UW * LDS_LDH];         // [chunk][plane][row][32]
    __shared__ __attribute__((aligned(16))) h16 Bs[2 * 2 * BWR * LDS_LDH];           // [buffer][plane][row][32]
    __shared__ float Gs[4 * 32 * 32];                                                // one 32 x 32 tile per wave: [j][i]
    __shared__ float Os[UW * OP];                                                    // [position][ti] of the current displacement row

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int mi = wave >> 1, ni = mi + (wave & 1);
    const int planeW = W / S2, segs = planeW / UW;
    int bid = blockIdx.x;
    const int useg = bid % segs; bid /= segs;
    const int par = bid % S2; bid /= S2;
    const int y = bid % H, n = bid / H;
    const int u0 = useg * UW;

    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(in1, (unsigned)((size_t)N * H * W * ld1 * 4));
    const __amdgpu_buffer_rsrc_t r2 = make_rsrc(in2, (unsigned)((size_t)N * H * W * ld2 * 4));
    float amax = 0.f;

    // ---- A segment: 64 positions x 32 NCH channels, split and staged once. Entry e -> (row e / 8, channel quad e % 8) of a chunk.
    {
        const int row0 = t >> 3, k4 = t & 7;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            f32x4 v[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = row0 + 32 * i;
                const int x = S2 * (u0 + row) + par;
                v[i] = buffer_load16<f32x4>(r1, (unsigned)(((size_t)(n * H + y) * W + x) * ld1 + coff1 + 32 * c + 4 * k4) * 4u, 0u);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = row0 + 32 * i;
                x4 sp[2];
                split_act<MODE>(v[i], sp, amax);
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    *reinterpret_cast<x4*>(&As[((c * 2 + p) * UW + row) * LDS_LDH + (((k4 >> 1) ^ lds_swz(row)) << 3) + ((k4 & 1) << 2)]) = sp[p];
            }
        }
    }

    // ---- B stream: step s = tj * NCH + chunk. This thread's 3 float4 of a chunk: entries e = t + 256 i -> (row e / 8, quad e % 8)
    unsigned bcol[3];            // byte offset of (position, quad) inside a row of in2, or "outside" (zeros)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = t + 256 * i, row = e >> 3, k4 = e & 7;
        const int u = u0 - 16 + row;
        bcol[i] = (unsigned)u < (unsigned)planeW ? (unsigned)((size_t)(S2 * u + par) * ld2 + coff2 + 4 * k4) * 4u : 0xFFFFFFF0u;
    }
    f32x4 breg[PD][3];
    auto issue = [&](const int slot, const int tj, const int c) {
        const int y2 = y + (tj - R) * S2;
        const bool rowok = tj < D && (unsigned)y2 < (unsigned)H;
        const unsigned rowoff = rowok ? (unsigned)((size_t)(n * H + y2) * W * ld2) * 4u + (unsigned)c * 128u : 0u;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            breg[slot][i] = buffer_load16<f32x4>(r2, (rowok && bcol[i] != 0xFFFFFFF0u) ? rowoff + bcol[i] : 0xFFFFFFF0u, 0u);
    };
    auto stage = [&](const int slot, const int buf) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = t + 256 * i, row = e >> 3, k4 = e & 7;
            x4 sp[2];
            split_act<MODE>(breg[slot][i], sp, amax);
#pragma unroll
            for (int p = 0; p < 2; ++p)
                *reinterpret_cast<x4*>(&Bs[((buf * 2 + p) * BWR + row) * LDS_LDH + (((k4 >> 1) ^ lds_swz(row)) << 3) + ((k4 & 1) << 2)]) = sp[p];
        }
    };
#pragma unroll
    for (int c = 0; c < PD; ++c) issue(c, 0, c);

    const int arow = (32 * mi + (lane & 31)) * LDS_LDH, asw = lds_swz(lane & 31);            // (32 mi is a multiple of 4: same swizzle)
    const int brow = (32 * ni + (lane & 31)) * LDS_LDH;
    const float invC = 1.0f / (float)(32 * NCH);
    float* __restrict__ gs = &Gs[wave * 1024];

    for (int tj = 0; tj < D; ++tj) {
        f32x16 accm, accc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accm[r] = 0.f; accc[r] = 0.f; }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int slot = c % PD, buf = c & 1;
            stage(slot, buf);                                   // waits for this chunk's loads (the oldest in flight)
            // the chunk PD steps ahead: same row while c + PD < NCH, else the next displacement row
            if (c + PD < NCH) issue(slot, tj, c + PD); else issue(slot, tj + 1, c + PD - NCH);
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int chunk = (((2 * m + (lane >> 5)) ^ asw) << 3);
                const x8 a0 = *reinterpret_cast<const x8*>(&As[((c * 2 + 0) * UW) * LDS_LDH + arow + chunk]);
                const x8 a1 = *reinterpret_cast<const x8*>(&As[((c * 2 + 1) * UW) * LDS_LDH + arow + chunk]);
                const x8 b0 = *reinterpret_cast<const x8*>(&Bs[((buf * 2 + 0) * BWR) * LDS_LDH + brow + chunk]);
                const x8 b1 = *reinterpret_cast<const x8*>(&Bs[((buf * 2 + 1) * BWR) * LDS_LDH + brow + chunk]);
                // first operand = B positions (accumulator rows j), second = A positions (accumulator column = lane & 31)
                accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a0, accc, 0, 0, 0);
                accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a1, accc, 0, 0, 0);
                accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0, accm, 0, 0, 0);
            }
        }
        // ---- tile -> LDS: lane holds pixel i = lane & 31 and B positions j = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        const int i = lane & 31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            gs[j * 32 + i] = accm[r] + accc[r] * 0.00048828125f;
        }
        __syncthreads();        // (a wave reads what its own other lanes wrote: LDS is in order per wave; the barrier also keeps the compiler honest)
        // band: position u = u0 + 32 mi + i pairs with B position u - R + ti = tile-local j = i + 16 + 32 (mi - ni) - R + ti
        const int joff = i + 16 + 32 * (mi - ni) - R;
        for (int ti = lane >> 5; ti < D; ti += 2) {
            const int j = joff + ti;
            if ((unsigned)j < 32u) Os[(32 * mi + i) * OP + ti] = vps_act(gs[j * 32 + i] * invC, act, slope);
        }
        __syncthreads();
        // ---- the displacement row of the 64 positions: D consecutive floats per pixel
        for (int e = t; e < UW * D; e += 256) {
            const int p = e / D, ti = e - p * D;
            const int x = S2 * (u0 + p) + par;
            out[((size_t)(n * H + y) * W + x) * out_ld + out_coff + tj * D + ti] = Os[p * OP + ti];
        }
        // (the next row's first barrier orders these reads of Os before its writes)
    }
    if (status && !(amax <= 65504.f)) atomicOr(status, 1);
}

}  // namespace

// -> 1 if an instance exists for this call and was enqueued, 0 if the caller has to use the vector-ALU kernels.
__attribute__((visibility("hidden")))
int vpsi_launch_corr_mfma(const float* in1, int ld1, int coff1, const float* in2, int ld2, int coff2, float* out, int out_ld, int out_coff,
                          int N, int H, int W, int C, int max_disp, int stride2, int act, float slope, int32_t* status, hipStream_t s) {
    // VPS_CORR_MFMA: bit 0 = the stride-2 / radius-10 case (FlowNetC: 212 us against 354 for the exact kernel), bit 1 = the stride-1 /
    // radius-4 case (LiteFlowNetCorr: 328 us against 263 - only 9 of the 64 computed columns of a tile are used there and the fp32 B
    // rows cost the same vector-memory time in both designs: default OFF). Default 1.
    static const int mode = getenv("VPS_CORR_MFMA") ? atoi(getenv("VPS_CORR_MFMA")) : 1;
    const bool off = !(mode & (stride2 == 2 ? 1 : 2));
    if (off || C != 256 || (ld1 & 3) || (ld2 & 3) || (coff1 & 3) || (coff2 & 3)) return 0;
    if ((size_t)N * H * W * ld1 * 4 >= 0xFFFFFFF0ull || (size_t)N * H * W * ld2 * 4 >= 0xFFFFFFF0ull) return 0;
    const int r = max_disp / stride2;
    if (stride2 == 2 && r == 10 && W % 128 == 0) {
        const long nb = (long)N * H * 2 * (W / 128);
        hipLaunchKernelGGL((corr_mfma_kernel<2, 10, 8>), dim3((unsigned)nb), dim3(256), 0, s, in1, ld1, coff1, in2, ld2, coff2, out, out_ld, out_coff,
                           N, H, W, act, slope, status);
        return 1;
    }
    if (stride2 == 1 && r == 4 && W % 64 == 0) {
        const long nb = (long)N * H * (W / 64);
        hipLaunchKernelGGL((corr_mfma_kernel<1, 4, 8>), dim3((unsigned)nb), dim3(256), 0, s, in1, ld1, coff1, in2, ld2, coff2, out, out_ld, out_coff,
                           N, H, W, act, slope, status);
        return 1;
    }
    return 0;
}
