// FlowNet2-side gather/scan kernels: resample2d, channelnorm, correlation, flow-guided feature warp,
// layout transposes, FlowNet2 input prep and the fused inter-stage tensor builder. All HBM-bound.
// Reference (relative to /root/reference/mmdet/models/flow_modules unless noted) is cited per kernel.
#include "common.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ------------------------------------------------------------------------------------------------
// resample2d_package/resample2d_kernel.cu:15-72 — backward warp with border-clamped taps.
// The reference mixes double literals into the weights ((1. - alpha) etc.): reproduced literally.
// One thread per (b,y,x): the flow is read once per pixel instead of once per channel.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float resample_tap(float v00, float v01, float v10, float v11, float alpha, float beta) {
    float val = 0.0f;
    val += static_cast<float>((1. - alpha) * (1. - beta) * v00);
    val += static_cast<float>((alpha) * (1. - beta) * v01);
    val += static_cast<float>((1. - alpha) * (beta) * v10);
    val += static_cast<float>((alpha) * (beta) * v11);
    return val;
}

__global__ __launch_bounds__(256)
void resample2d_kernel(vps_tensor4 in, vps_tensor4 flow, vps_tensor4 out, int B, int C, int H, int W) {
    const long total = (long)B * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        const int y = (int)((idx / W) % H);
        const int b = (int)(idx / ((long)W * H));
        const float dx = flow.p[b * flow.sn + 0 * flow.sc + y * flow.sh + x * flow.sw];
        const float dy = flow.p[b * flow.sn + 1 * flow.sc + y * flow.sh + x * flow.sw];
        const float xf = (float)x + dx, yf = (float)y + dy;
        const float alpha = xf - floorf(xf), beta = yf - floorf(yf);
        const int xL = max(min((int)floorf(xf), W - 1), 0);
        const int xR = max(min((int)(floorf(xf) + 1.f), W - 1), 0);
        const int yT = max(min((int)floorf(yf), H - 1), 0);
        const int yB = max(min((int)(floorf(yf) + 1.f), H - 1), 0);
        const float* ib = in.p + b * in.sn;
        for (int c = 0; c < C; ++c) {
            const float* ic = ib + c * in.sc;
            const float v = resample_tap(ic[yT * in.sh + xL * in.sw], ic[yT * in.sh + xR * in.sw],
                                         ic[yB * in.sh + xL * in.sw], ic[yB * in.sh + xR * in.sw], alpha, beta);
            out.p[b * out.sn + c * out.sc + y * out.sh + x * out.sw] = v;
        }
    }
}

// channelnorm_package/channelnorm_kernel.cu:18-60
__global__ __launch_bounds__(256)
void channelnorm_kernel(vps_tensor4 in, vps_tensor4 out, int B, int C, int H, int W) {
    const long total = (long)B * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        const int y = (int)((idx / W) % H);
        const int b = (int)(idx / ((long)W * H));
        float r = 0.f;
        for (int c = 0; c < C; ++c) {
            const float v = in.p[b * in.sn + c * in.sc + y * in.sh + x * in.sw];
            r += v * v;
        }
        out.p[b * out.sn + y * out.sh + x * out.sw] = sqrtf(r);
    }
}

// ------------------------------------------------------------------------------------------------
// correlation_cuda_kernel.cu:46-147 (kernel_size 1, stride1 1). One 64-lane wavefront per output pixel:
// lanes stride the channel axis with float4 loads (in1 pixel kept in registers), displacements are
// processed 64 at a time and reduced with a transposing butterfly (63 shuffles per 64 displacements
// instead of 6 per displacement), so lane l ends up owning displacement l and the NHWC store of the
// 441/81 output channels is one coalesced wave store. No padded copies (the reference zero-fills and
// writes two padded NHWC tensors first).
// ------------------------------------------------------------------------------------------------
template <int NC4>  // float4 chunks per lane (C <= 256*NC4)
__global__ __launch_bounds__(256)
void correlation_kernel(const float* __restrict__ in1, int ld1, int coff1,
                        const float* __restrict__ in2, int ld2, int coff2,
                        float* __restrict__ out, int out_ld, int out_coff,
                        int N, int H, int W, int C, int r, int s2, int act, float slope) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long npix = (long)N * H * W;
    const int D = 2 * r + 1;
    const int ND = D * D;
    const int c4n = C >> 2;
    const float invC = 1.0f / (float)C;
    for (long pix = (long)blockIdx.x * 4 + wave; pix < npix; pix += (long)gridDim.x * 4) {
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const int n = (int)(pix / ((long)W * H));
        f32x4 a[NC4];
#pragma unroll
        for (int q = 0; q < NC4; ++q) {
            const int c4 = lane + 64 * q;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            a[q] = c4 < c4n ? *reinterpret_cast<const f32x4*>(in1 + (size_t)pix * ld1 + coff1 + 4 * c4) : z;
        }
        for (int d0 = 0; d0 < ND; d0 += 64) {
            float v[64];
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const int dd = d0 + i;
                float p = 0.f;
                if (dd < ND) {
                    const int tj = dd / D - r, ti = dd % D - r;
                    const int y2 = y + tj * s2, x2 = x + ti * s2;
                    if ((unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W) {
                        const float* p2 = in2 + ((size_t)(n * H + y2) * W + x2) * ld2 + coff2;
#pragma unroll
                        for (int q = 0; q < NC4; ++q) {
                            const int c4 = lane + 64 * q;
                            if (c4 < c4n) {
                                const f32x4 b = *reinterpret_cast<const f32x4*>(p2 + 4 * c4);
                                p += a[q][0] * b[0] + a[q][1] * b[1] + a[q][2] * b[2] + a[q][3] * b[3];
                            }
                        }
                    }
                }
                v[i] = p;
            }
            // transposing butterfly: after the last stage lane l holds sum over lanes of v[l]
#pragma unroll
            for (int off = 32, nn = 64; off >= 1; off >>= 1, nn >>= 1) {
                const bool hi = (lane & off) != 0;
#pragma unroll
                for (int i = 0; i < nn / 2; ++i) {
                    const float keep = hi ? v[i + nn / 2] : v[i];
                    const float send = hi ? v[i] : v[i + nn / 2];
                    v[i] = keep + __shfl_xor(send, off, 64);
                }
            }
            const int dd = d0 + lane;
            if (dd < ND) out[(size_t)pix * out_ld + out_coff + dd] = vps_act(v[0] * invC, act, slope);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Correlation with ONE LANE PER OUTPUT PIXEL (round 3). The kernels above and below put the CHANNELS on the lanes, so every
// output value needs a 64-lane reduction (as many shuffle / select instructions as multiply-adds). Here the channel loop is
// sequential and a lane owns P = 2 neighbouring output pixels and all D displacements of one displacement row:
//     acc[p][ti] += A[c][u + p] * B[c][u + p + ti]          (plane coordinates u: x = S2 * u + parity)
// with both operands read from LDS tiles that were transposed when staged ([channel][plane position], lanes on consecutive
// positions: conflict-free 8-byte reads), P + D - 1 reads of B for P * D multiply-adds - no cross-lane traffic at all.
// Work decomposition: workgroup = (image, row y, column parity, 128 plane positions, group of 4 displacement rows), wave w of
// it = displacement row tj = 4 * group + w; the A tile is shared by the 4 waves, each wave stages its own B row tile.
// Stride-2 displacements (FlowNetC) only ever pair pixels of equal column parity: each parity plane is a stride-1 problem.
// A stage covers 32 channels = one whole 128-byte line per pixel (8 lanes x 16 bytes, coalesced; staging 8 channels at a time
// fetched every line four times through a thrashing L1: 1.1 ms instead of 0.27 for the 81-channel case). The loads of stage
// s + 1 are issued before the multiply-adds of stage s (registers -> LDS after the next barrier).
// Out-of-image samples are zeros: the loads are buffer loads whose offset lies beyond the tensor for them.
// Arithmetic: fp32 multiply-adds in channel order, * 1/C at the end (the reference sums per 32-lane warp in another order:
// agreement to ~1e-6 like the kernels above).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 corr_buffer_load16(const __amdgpu_buffer_rsrc_t r, const unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}

template <int S2, int R>
__global__ __launch_bounds__(256)
void correlation_lds_kernel(const float* __restrict__ in1, int ld1, int coff1, const float* __restrict__ in2, int ld2, int coff2,
                            float* __restrict__ out, int out_ld, int out_coff, int N, int H, int W, int C, int act, float slope) {
    constexpr int D = 2 * R + 1;
    constexpr int P = 2;
    constexpr int CC = 32;                // channels per stage: one 128-byte line per pixel
    constexpr int UW = 64 * P;            // plane positions of a workgroup
    constexpr int BW = UW + 2 * R;        // B tile: positions u0 - R .. u0 + UW + R - 1
    constexpr int AWP = UW + 2;           // row pitches (floats): even (8-byte reads), = 2 mod 8 (transposing writes: 2-way conflicts)
    constexpr int BWP = ((BW + 7) & ~7) + 2;
    constexpr int NAL = UW * 8 / 256;         // f32x4 loads per thread and A stage
    constexpr int NBL = (BW * 8 + 63) / 64;   // f32x4 loads per lane and B stage
    __shared__ __attribute__((aligned(16))) float As[CC * AWP];
    __shared__ __attribute__((aligned(16))) float Bs[4][CC * BWP];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int planeW = W / S2, segs = planeW / UW, groups = (D + 3) / 4;
    int bid = blockIdx.x;
    const int grp = bid % groups; bid /= groups;
    const int useg = bid % segs; bid /= segs;
    const int par = bid % S2; bid /= S2;
    const int y = bid % H, n = bid / H;
    const int u0 = useg * UW;
    const int tjw = grp * 4 + wave;                  // this wave's displacement row index 0 .. D-1 (>= D: idle wave)
    const bool active = tjw < D;
    const int y2 = y + (tjw - R) * S2;
    const bool rowok = active && (unsigned)y2 < (unsigned)H;

    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in1), 0, (int)((size_t)N * H * W * ld1 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in2), 0, (int)((size_t)N * H * W * ld2 * 4), 0x00020000);
    // entry e -> (plane position j = e / 8, channel quad q = e % 8): the 8 lanes of a pixel read its 128-byte line
    unsigned aoff[NAL];
#pragma unroll
    for (int i = 0; i < NAL; ++i) {
        const int e = t + 256 * i, j = e >> 3, q = e & 7;
        const int x = S2 * (u0 + j) + par;
        aoff[i] = (unsigned)(((size_t)(n * H + y) * W + x) * ld1 + coff1 + 4 * q) * 4u;
    }
    unsigned boff[NBL];
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
        const int e = lane + 64 * i, j = e >> 3, q = e & 7;
        const int u = u0 - R + j;
        const bool ok = rowok && j < BW && (unsigned)u < (unsigned)planeW;
        const int x = S2 * u + par;
        boff[i] = ok ? (unsigned)(((size_t)(n * H + (rowok ? y2 : 0)) * W + x) * ld2 + coff2 + 4 * q) * 4u : 0xFFFFFFF0u;
    }

    float acc[P][D];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int k = 0; k < D; ++k) acc[p][k] = 0.f;

    float* __restrict__ bs = &Bs[wave][0];
    f32x4 av[NAL], bv[NBL];
    auto issue = [&](const int c0) {
#pragma unroll
        for (int i = 0; i < NAL; ++i) av[i] = corr_buffer_load16(r1, aoff[i] + (unsigned)c0 * 4u);
#pragma unroll
        for (int i = 0; i < NBL; ++i) bv[i] = corr_buffer_load16(r2, boff[i] == 0xFFFFFFF0u ? 0xFFFFFFF0u : boff[i] + (unsigned)c0 * 4u);
    };
    issue(0);
    for (int c0 = 0; c0 < C; c0 += CC) {
        __syncthreads();                               // the previous stage's readers are done with both tiles
#pragma unroll
        for (int i = 0; i < NAL; ++i) {
            const int e = t + 256 * i, j = e >> 3, q = e & 7;
#pragma unroll
            for (int k = 0; k < 4; ++k) As[(4 * q + k) * AWP + j] = av[i][k];
        }
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            const int e = lane + 64 * i, j = e >> 3, q = e & 7;
            if (j < BW) {
#pragma unroll
                for (int k = 0; k < 4; ++k) bs[(4 * q + k) * BWP + j] = bv[i][k];
            }
        }
        __syncthreads();
        if (c0 + CC < C) issue(c0 + CC);               // in flight under this stage's multiply-adds
        if (!active) continue;
        // ---- compute: lane owns positions lane * 2, lane * 2 + 1
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll 4
        for (int c = 0; c < CC; ++c) {
            float b[P + D - 1 + 1];
            const f32x2 a2 = *reinterpret_cast<const f32x2*>(&As[c * AWP + lane * 2]);
#pragma unroll
            for (int k = 0; k < (P + D - 1 + 1) / 2; ++k) {
                const f32x2 q2 = *reinterpret_cast<const f32x2*>(&bs[c * BWP + lane * 2 + 2 * k]);
                b[2 * k] = q2[0]; b[2 * k + 1] = q2[1];
            }
#pragma unroll
            for (int k = 0; k < D; ++k) {
                acc[0][k] = fmaf(a2[0], b[k], acc[0][k]);
                acc[1][k] = fmaf(a2[1], b[k + 1], acc[1][k]);
            }
        }
    }
    if (!active) return;
    const float invC = 1.0f / (float)C;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int x = S2 * (u0 + lane * P + p) + par;
        float* __restrict__ o = out + ((size_t)(n * H + y) * W + x) * out_ld + out_coff + tjw * D;
#pragma unroll
        for (int k = 0; k < D; ++k) o[k] = vps_act(acc[p][k] * invC, act, slope);
    }
}

// Specialised correlation for the two configurations on the path (FlowNetC: stride2 2, radius 10; LiteFlowNetCorr:
// stride2 1, radius 4). The generic kernel above re-reads every in2 sample once per output pixel and is L2-bandwidth
// bound (14.5 GB of L2 reads for FlowNetC at 128x256). Here one wavefront owns FOUR output pixels spaced stride2 apart on
// a row: their displacement windows overlap in all but 3 columns, so each in2 sample (1 KiB coalesced load) is used by
// up to four pixels from registers (3.5x fewer loads). Cross-lane sums use the same transposing butterfly; slot
// (ti, p) -> lane, so each pixel's displacement run is stored as contiguous floats.
template <int S2, int R, int NC4>
__global__ __launch_bounds__(256)
void correlation4_kernel(const float* __restrict__ in1, int ld1, int coff1,
                         const float* __restrict__ in2, int ld2, int coff2,
                         float* __restrict__ out, int out_ld, int out_coff,
                         int N, int H, int W, int C, int act, float slope) {
    constexpr int D = 2 * R + 1;
    constexpr int NBATCH = (D + 15) / 16;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int gpr = W / 4;                       // pixel groups per row
    const long ngroups = (long)N * H * gpr;
    const int c4n = C >> 2;
    const float invC = 1.0f / (float)C;
    for (long grp = (long)blockIdx.x * 4 + wave; grp < ngroups; grp += (long)gridDim.x * 4) {
        const int g = (int)(grp % gpr);
        const int y = (int)((grp / gpr) % H);
        const int n = (int)(grp / ((long)gpr * H));
        const int xb = (S2 == 2) ? (8 * (g >> 1) + (g & 1)) : 4 * g;
        const size_t rowbase = ((size_t)n * H + y) * W;
        f32x4 a[4][NC4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < NC4; ++q) {
                const int c4 = lane + 64 * q;
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                a[p][q] = c4 < c4n ? *reinterpret_cast<const f32x4*>(in1 + (rowbase + xb + p * S2) * ld1 + coff1 + 4 * c4) : z;
            }
        for (int tj = -R; tj <= R; ++tj) {
            const int y2 = y + tj * S2;
            const bool rowok = (unsigned)y2 < (unsigned)H;
            const float* __restrict__ row2 = in2 + ((size_t)n * H + (rowok ? y2 : 0)) * W * ld2 + coff2;
#pragma unroll
            for (int bt = 0; bt < NBATCH; ++bt) {
                constexpr int dummy = 0; (void)dummy;
                const int base = bt * 16;                 // first displacement index of this batch
                float v[64];
#pragma unroll
                for (int i = 0; i < 64; ++i) v[i] = 0.f;
                // columns uu = ti_idx + p needed by slots (ti_idx in [base, base+16) , p in [0,4))
#pragma unroll
                for (int du = 0; du < 19; ++du) {
                    const int uu = base + du;
                    if (uu > D - 1 + 3) continue;
                    const int x2 = xb + S2 * (uu - R);
                    f32x4 b[NC4];
                    const bool ok = rowok && (unsigned)x2 < (unsigned)W;
#pragma unroll
                    for (int q = 0; q < NC4; ++q) {
                        const int c4 = lane + 64 * q;
                        f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        b[q] = (ok && c4 < c4n) ? *reinterpret_cast<const f32x4*>(row2 + (size_t)x2 * ld2 + 4 * c4) : z;
                    }
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int ti = uu - p;            // displacement index of pixel p fed by this column
                        if (ti < base || ti >= base + 16 || ti > D - 1) continue;
                        float acc = 0.f;
#pragma unroll
                        for (int q = 0; q < NC4; ++q)
                            acc += a[p][q][0] * b[q][0] + a[p][q][1] * b[q][1] + a[p][q][2] * b[q][2] + a[p][q][3] * b[q][3];
                        v[(ti - base) * 4 + p] = acc;
                    }
                }
#pragma unroll
                for (int off = 32, nn = 64; off >= 1; off >>= 1, nn >>= 1) {
                    const bool hi = (lane & off) != 0;
#pragma unroll
                    for (int i = 0; i < nn / 2; ++i) {
                        const float keep = hi ? v[i + nn / 2] : v[i];
                        const float send = hi ? v[i] : v[i + nn / 2];
                        v[i] = keep + __shfl_xor(send, off, 64);
                    }
                }
                const int ti = base + (lane >> 2), p = lane & 3;
                if (ti < D)
                    out[(rowbase + xb + p * S2) * out_ld + out_coff + (tj + R) * D + ti] = vps_act(v[0] * invC, act, slope);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// flow_modules.py:126-148 WarpingLayer: grid = linspace(-1,1) + flow/((size-1)/2), then F.grid_sample with
// its defaults (bilinear, zeros, align_corners=False) — the align_corners mismatch is reproduced, not fixed.
// NHWC: one thread per (pixel, float4 of channels) -> 1 KiB coalesced reads per corner at C=256.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float linspace_m1_p1(int i, int n) {
    // torch.linspace(-1, 1, n) fp32 CPU kernel: symmetric evaluation around the midpoint
    const float step = 2.0f / (float)(n - 1);
    return (i < n / 2) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(n - 1 - i));
}

__global__ __launch_bounds__(256)
void flow_warp_kernel(const float* __restrict__ in, int in_ld, int in_coff,
                      const float* __restrict__ flow, int flow_ld, int flow_coff,
                      float* __restrict__ out, int out_ld, int out_coff, int N, int H, int W, int C) {
    const int c4n = C >> 2;
    const long total = (long)N * H * W * c4n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const long pix = idx / c4n;
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const int n = (int)(pix / ((long)W * H));
        const float fx = flow[(size_t)pix * flow_ld + flow_coff + 0];
        const float fy = flow[(size_t)pix * flow_ld + flow_coff + 1];
        const float gx = linspace_m1_p1(x, W) + fx / (((float)W - 1.0f) / 2.0f);
        const float gy = linspace_m1_p1(y, H) + fy / (((float)H - 1.0f) / 2.0f);
        // grid_sampler_unnormalize, align_corners=False
        const float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;
        const float iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
        const float ix_nw = floorf(ix), iy_nw = floorf(iy);
        const float ix_se = ix_nw + 1.f, iy_se = iy_nw + 1.f;
        const float nw = (ix_se - ix) * (iy_se - iy);
        const float ne = (ix - ix_nw) * (iy_se - iy);
        const float sw = (ix_se - ix) * (iy - iy_nw);
        const float se = (ix - ix_nw) * (iy - iy_nw);
        const int x0 = (int)ix_nw, y0 = (int)iy_nw, x1 = x0 + 1, y1 = y0 + 1;
        const float* base = in + (size_t)n * H * W * in_ld + in_coff + 4 * c4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bool xin0 = (unsigned)x0 < (unsigned)W, xin1 = (unsigned)x1 < (unsigned)W;
        const bool yin0 = (unsigned)y0 < (unsigned)H, yin1 = (unsigned)y1 < (unsigned)H;
        // the four corners are requested unconditionally at clamped coordinates and added under the same conditions and in
        // the same order as before (a load behind `if` waits for itself: four serial latencies per pixel)
        const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x1, 0), W - 1);
        const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y1, 0), H - 1);
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + (size_t)(yc0 * W + xc0) * in_ld);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + (size_t)(yc0 * W + xc1) * in_ld);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + (size_t)(yc1 * W + xc0) * in_ld);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + (size_t)(yc1 * W + xc1) * in_ld);
        acc = (yin0 && xin0) ? acc + v00 * nw : acc;
        acc = (yin0 && xin1) ? acc + v01 * ne : acc;
        acc = (yin1 && xin0) ? acc + v10 * sw : acc;
        acc = (yin1 && xin1) ? acc + v11 * se : acc;
        *reinterpret_cast<f32x4*>(out + (size_t)pix * out_ld + out_coff + 4 * c4) = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// layout transposes (operator-API wrappers and tests; the pipeline itself stays NHWC)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int out_ld, int out_coff,
                         int N, int C, int H, int W, int Cpad) {
    __shared__ float tile[32][33];
    // grid: x over pixel tiles of 32, y over channel tiles of 32, z over n
    const long HW = (long)H * W;
    const int n = blockIdx.z;
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        const long p = p0 + tx;
        tile[j][tx] = (c < C && p < HW) ? in[((size_t)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const long p = p0 + j;
        const int c = c0 + tx;
        if (p < HW && c < Cpad) out[((size_t)n * HW + p) * out_ld + out_coff + c] = tile[tx][j];
    }
}

__global__ __launch_bounds__(256)
void nhwc_to_nchw_kernel(const float* __restrict__ in, int in_ld, int in_coff, float* __restrict__ out,
                         int N, int C, int H, int W) {
    __shared__ float tile[32][33];
    const long HW = (long)H * W;
    const int n = blockIdx.z;
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const long p = p0 + j;
        const int c = c0 + tx;
        tile[j][tx] = (p < HW && c < C) ? in[((size_t)n * HW + p) * in_ld + in_coff + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        const long p = p0 + tx;
        if (c < C && p < HW) out[((size_t)n * C + c) * HW + p] = tile[tx][j];
    }
}

// ------------------------------------------------------------------------------------------------
// FlowNet2 input prep: utils/flow_utils.py:5-10 (denormalize x*std+mean, two roundings) and
// flownet2.py:135-139 (rgb_mean jointly over both frames; (x-mean)/rgb_max; channels img|ref).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float denorm(float v, float s, float m) { return __fadd_rn(__fmul_rn(v, s), m); }

// FlowNet2 sees the frame pair zero-padded (in 0..255 RGB space) to [Hp, Wp] for the two sizes the reference special-cases
// (panoptic_fusetrack.py:125-128: 800x1600 -> 832x1664, 200x400 -> 256x448); rgb_mean is taken over the PADDED tensor
// (flownet2.py:135): the pad pixels add 0 to the sums and count in the divisor.
__global__ __launch_bounds__(256)
void flow_prep_sum_kernel(const float* __restrict__ img, const float* __restrict__ ref,
                          const float* __restrict__ mean3, const float* __restrict__ std3,
                          long HW, double* __restrict__ partial) {
    __shared__ double red[3][4];
    double s[3] = {0.0, 0.0, 0.0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s[c] += (double)denorm(img[c * HW + i], std3[c], mean3[c]);
            s[c] += (double)denorm(ref[c * HW + i], std3[c], mean3[c]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double v = s[c];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[c][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        partial[threadIdx.x * gridDim.x + blockIdx.x] =
            red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

__global__ void flow_prep_mean_kernel(const double* __restrict__ partial, int nblk, long HW, float* __restrict__ rgb_mean) {
    const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;  // 192 threads: one wave per channel
    double v = 0.0;
    for (int i = lane; i < nblk; i += 64) v += partial[c * nblk + i];
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) rgb_mean[c] = (float)(v / (double)(2 * HW));
}

__global__ __launch_bounds__(256)
void flow_prep_write_kernel(const float* __restrict__ img, const float* __restrict__ ref,
                            const float* __restrict__ mean3, const float* __restrict__ std3,
                            const float* __restrict__ rgb_mean, float* __restrict__ out, int out_ld, int H, int W, int Hp, int Wp) {
    const long HW = (long)H * W, HWp = (long)Hp * Wp;
    for (long ip = (long)blockIdx.x * blockDim.x + threadIdx.x; ip < HWp; ip += (long)gridDim.x * blockDim.x) {
        const int y = (int)(ip / Wp), x = (int)(ip - (long)y * Wp);
        const bool in = y < H && x < W;
        const long i = in ? (long)y * W + x : 0;
        float* o = out + (size_t)ip * out_ld;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = denorm(img[c * HW + i], std3[c], mean3[c]), b = denorm(ref[c * HW + i], std3[c], mean3[c]);
            o[c] = ((in ? a : 0.f) - rgb_mean[c]) / 255.0f;
            o[3 + c] = ((in ? b : 0.f) - rgb_mean[c]) / 255.0f;
        }
        for (int c = 6; c < out_ld; ++c) o[c] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// FlowNet2 inter-stage builder (flownet2.py:142-151, 154-163, 166-174, 179-187): x4 upsample of the
// 1/4-res flow (bilinear align_corners=False, or nearest), Resample2d of the second image, channel norms
// and the concat writes fused into ONE pass over the full-resolution frame (the reference runs 6-8
// separate full-res kernels and a torch.cat per stage).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float scale_lo(float v, float mul, int div_mode) { return div_mode ? v / mul : v * mul; }

// ---- whole-pixel variants of the inter-network stage: the 12-float pixel of the next network's input is composed in registers and
// stored as three float4 (a wave writes 3 KB contiguous). flow_stage_kernel stores 4-byte channels at a 48-byte pixel stride
// (every store instruction touches 24 cache lines, and the 6 image channels needed a copy pass of their own).
__device__ __forceinline__ void warp_img2(const float* __restrict__ x6, const int x_ld, const int H, const int W, const int x, const int y,
                                          const float fx, const float fy, const float* __restrict__ xi, float (&wv)[3], float& nrm) {
    // Resample2d of image 2 (channels 3..5) at (x + fx, y + fy) and the ChannelNorm of img1 - warped (flownet2.py:142-151)
    const float xf = (float)x + fx, yf = (float)y + fy;
    const float alpha = xf - floorf(xf), beta = yf - floorf(yf);
    const int xL = max(min((int)floorf(xf), W - 1), 0);
    const int xR = max(min((int)(floorf(xf) + 1.f), W - 1), 0);
    const int yT = max(min((int)floorf(yf), H - 1), 0);
    const int yB = max(min((int)(floorf(yf) + 1.f), H - 1), 0);
    const float* tl = x6 + ((size_t)yT * W + xL) * x_ld + 3;
    const float* tr = x6 + ((size_t)yT * W + xR) * x_ld + 3;
    const float* bl = x6 + ((size_t)yB * W + xL) * x_ld + 3;
    const float* br = x6 + ((size_t)yB * W + xR) * x_ld + 3;
    nrm = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        wv[c] = resample_tap(tl[c], tr[c], bl[c], br[c], alpha, beta);
        const float df = xi[c] - wv[c];
        nrm += df * df;
    }
}

// FlowNetS input (flownet2.py:142-163): [img1 img2 | warped img2 | flow / mul | ||img1 - warped||], flow = bilinear x4 of flo * mul
__global__ __launch_bounds__(256)
void flow_stage_s_kernel(const float* __restrict__ x6, const float* __restrict__ flo, int flo_ld, int flo_coff, int H, int W, float mul,
                         float* __restrict__ out) {
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
        const float* p00 = flo + ((size_t)y0 * Wl + x0) * flo_ld + flo_coff;
        const float* p01 = p00 + (size_t)xp * flo_ld;
        const float* p10 = p00 + (size_t)yp * Wl * flo_ld;
        const float* p11 = p10 + (size_t)xp * flo_ld;
        const float fx = hy * (hx * (p00[0] * mul) + lx * (p01[0] * mul)) + ly * (hx * (p10[0] * mul) + lx * (p11[0] * mul));
        const float fy = hy * (hx * (p00[1] * mul) + lx * (p01[1] * mul)) + ly * (hx * (p10[1] * mul) + lx * (p11[1] * mul));
        const f32x4 a = *reinterpret_cast<const f32x4*>(x6 + (size_t)idx * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(x6 + (size_t)idx * 8 + 4);
        const float xi[3] = {a[0], a[1], a[2]};
        float wv[3], nrm;
        warp_img2(x6, 8, H, W, x, y, fx, fy, xi, wv, nrm);
        float* o = out + (size_t)idx * 12;
        const f32x4 o1 = {b[0], b[1], wv[0], wv[1]};
        const f32x4 o2 = {wv[2], fx / mul, fy / mul, sqrtf(nrm)};
        *reinterpret_cast<f32x4*>(o) = a;
        *reinterpret_cast<f32x4*>(o + 4) = o1;
        *reinterpret_cast<f32x4*>(o + 8) = o2;
    }
}

// FlowNetFusion input (flownet2.py:166-187): [img1 | flow_sd | flow_s2 | |flow_sd| | |flow_s2| | diff_sd | diff_s2 | 0],
// flow_s2 = nearest x4 of flo_s2 * mul, flow_sd = nearest x4 of flo_sd / mul
__global__ __launch_bounds__(256)
void flow_stage_f_kernel(const float* __restrict__ x6, const float* __restrict__ flo_s2, int s2_ld, int s2_coff,
                         const float* __restrict__ flo_sd, int sd_ld, int sd_coff, int H, int W, float mul, float* __restrict__ out) {
    const int Wl = W >> 2;
    const long HW = (long)H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < HW; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W), y = (int)(idx / W);
        const size_t lo = (size_t)(y >> 2) * Wl + (x >> 2);
        const float* p2 = flo_s2 + lo * s2_ld + s2_coff;
        const float* pd = flo_sd + lo * sd_ld + sd_coff;
        const float f2x = p2[0] * mul, f2y = p2[1] * mul;
        const float fdx = pd[0] / mul, fdy = pd[1] / mul;
        const f32x4 a = *reinterpret_cast<const f32x4*>(x6 + (size_t)idx * 8);
        const float xi[3] = {a[0], a[1], a[2]};
        float wv[3], n2, nd;
        warp_img2(x6, 8, H, W, x, y, f2x, f2y, xi, wv, n2);
        warp_img2(x6, 8, H, W, x, y, fdx, fdy, xi, wv, nd);
        float* o = out + (size_t)idx * 12;
        const f32x4 o0 = {a[0], a[1], a[2], fdx};
        const f32x4 o1 = {fdy, f2x, f2y, sqrtf(fdx * fdx + fdy * fdy)};
        const f32x4 o2 = {sqrtf(f2x * f2x + f2y * f2y), sqrtf(nd), sqrtf(n2), 0.f};
        *reinterpret_cast<f32x4*>(o) = o0;
        *reinterpret_cast<f32x4*>(o + 4) = o1;
        *reinterpret_cast<f32x4*>(o + 8) = o2;
    }
}

__global__ __launch_bounds__(256)
void flow_stage_kernel(const float* __restrict__ x6, int x_ld, const float* __restrict__ flo, int flo_ld, int flo_coff,
                       int H, int W, int up_mode, float mul, int div_mode,
                       float* __restrict__ out, int out_ld, int flow_off, float flow_out_div, int warp_off,
                       int diffnorm_off, int flownorm_off, int img_off) {
    const int Hl = H >> 2, Wl = W >> 2;
    const long HW = (long)H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < HW; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W), y = (int)(idx / W);
        float fx, fy;
        if (up_mode == 1) {  // nearest: src = floor(dst * 0.25)
            const float* p = flo + ((size_t)(y >> 2) * Wl + (x >> 2)) * flo_ld + flo_coff;
            fx = scale_lo(p[0], mul, div_mode);
            fy = scale_lo(p[1], mul, div_mode);
        } else {  // aten upsample_bilinear2d, align_corners=False, scale 1/4
            float sy = 0.25f * ((float)y + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
            float sx = 0.25f * ((float)x + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
            const int y0 = (int)sy, x0 = (int)sx;
            const int yp = y0 < Hl - 1 ? 1 : 0, xp = x0 < Wl - 1 ? 1 : 0;
            const float ly = sy - (float)y0, lx = sx - (float)x0;
            const float hy = 1.f - ly, hx = 1.f - lx;
            const float* p00 = flo + ((size_t)y0 * Wl + x0) * flo_ld + flo_coff;
            const float* p01 = p00 + (size_t)xp * flo_ld;
            const float* p10 = p00 + (size_t)yp * Wl * flo_ld;
            const float* p11 = p10 + (size_t)xp * flo_ld;
            fx = hy * (hx * scale_lo(p00[0], mul, div_mode) + lx * scale_lo(p01[0], mul, div_mode)) +
                 ly * (hx * scale_lo(p10[0], mul, div_mode) + lx * scale_lo(p11[0], mul, div_mode));
            fy = hy * (hx * scale_lo(p00[1], mul, div_mode) + lx * scale_lo(p01[1], mul, div_mode)) +
                 ly * (hx * scale_lo(p10[1], mul, div_mode) + lx * scale_lo(p11[1], mul, div_mode));
        }
        float* o = out + (size_t)idx * out_ld;
        const float* xi = x6 + (size_t)idx * x_ld;
        if (img_off >= 0) { o[img_off] = xi[0]; o[img_off + 1] = xi[1]; o[img_off + 2] = xi[2]; }
        if (flow_off >= 0) {
            o[flow_off] = flow_out_div > 0.f ? fx / flow_out_div : fx;
            o[flow_off + 1] = flow_out_div > 0.f ? fy / flow_out_div : fy;
        }
        if (flownorm_off >= 0) o[flownorm_off] = sqrtf(fx * fx + fy * fy);
        if (warp_off >= 0 || diffnorm_off >= 0) {
            const float xf = (float)x + fx, yf = (float)y + fy;
            const float alpha = xf - floorf(xf), beta = yf - floorf(yf);
            const int xL = max(min((int)floorf(xf), W - 1), 0);
            const int xR = max(min((int)(floorf(xf) + 1.f), W - 1), 0);
            const int yT = max(min((int)floorf(yf), H - 1), 0);
            const int yB = max(min((int)(floorf(yf) + 1.f), H - 1), 0);
            const float* tl = x6 + ((size_t)yT * W + xL) * x_ld + 3;
            const float* tr = x6 + ((size_t)yT * W + xR) * x_ld + 3;
            const float* bl = x6 + ((size_t)yB * W + xL) * x_ld + 3;
            const float* br = x6 + ((size_t)yB * W + xR) * x_ld + 3;
            float nrm = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float wv = resample_tap(tl[c], tr[c], bl[c], br[c], alpha, beta);
                if (warp_off >= 0) o[warp_off + c] = wv;
                const float df = xi[c] - wv;
                nrm += df * df;
            }
            if (diffnorm_off >= 0) o[diffnorm_off] = sqrtf(nrm);
        }
    }
}

}  // namespace

extern "C" int vps_resample2d(vps_tensor4 in, vps_tensor4 flow, vps_tensor4 out, int B, int C, int H, int W, void* stream) {
    if (!in.p || !flow.p || !out.p || B <= 0 || C <= 0 || H <= 0 || W <= 0) return VPS_EARG(1);
    hipLaunchKernelGGL(resample2d_kernel, dim3(stream_grid((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                       in, flow, out, B, C, H, W);
    return vps_launch_status();
}

extern "C" int vps_channelnorm(vps_tensor4 in, vps_tensor4 out, int B, int C, int H, int W, void* stream) {
    if (!in.p || !out.p || B <= 0 || C <= 0 || H <= 0 || W <= 0) return VPS_EARG(1);
    hipLaunchKernelGGL(channelnorm_kernel, dim3(stream_grid((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                       in, out, B, C, H, W);
    return vps_launch_status();
}

bool vpsi_launch_corr4h(const float* in1, int ld1, int coff1, const float* in2, int ld2, int coff2, float* out, int out_ld, int out_coff,
                        int N, int H, int W, int C, int r, int stride2, int act, float slope, hipStream_t s);
int vpsi_launch_corr_mfma(const float* in1, int ld1, int coff1, const float* in2, int ld2, int coff2, float* out, int out_ld, int out_coff,
                          int N, int H, int W, int C, int max_disp, int stride2, int act, float slope, int32_t* status, hipStream_t s);

// the same operator in split fp16 on the matrix cores where an instance exists (corr_mfma.hip: the two configurations of the path),
// else the exact kernels below. status: device word that receives bit 0 when an operand lies beyond the fp16 range (|x| > 65504) -
// the result is then not fp32-grade and the caller repeats the call with vps_correlation.
extern "C" int vps_correlation_f16(const float* in1, int ld1, int coff1, const float* in2, int ld2, int coff2,
                                   float* out, int out_ld, int out_coff, int N, int H, int W, int C,
                                   int max_disp, int stride2, int act, float slope, int32_t* status, void* stream) {
    if (!in1 || !in2 || !out || !status || N <= 0 || H <= 0 || W <= 0) return VPS_EARG(1);
    if (C <= 0 || (C & 3) || C > 1024 || (ld1 & 3) || (ld2 & 3) || (coff1 & 3) || (coff2 & 3)) return VPS_EARG(2);
    if (stride2 <= 0 || max_disp < 0) return VPS_EARG(3);
    if (vpsi_launch_corr_mfma(in1, ld1, coff1, in2, ld2, coff2, out, out_ld, out_coff, N, H, W, C, max_disp, stride2, act, slope, status,
                              (hipStream_t)stream))
        return vps_launch_status();
    return vps_correlation(in1, ld1, coff1, in2, ld2, coff2, out, out_ld, out_coff, N, H, W, C, max_disp, stride2, act, slope, stream);
}

extern "C" int vps_correlation(const float* in1, int ld1, int coff1, const float* in2, int ld2, int coff2,
                               float* out, int out_ld, int out_coff, int N, int H, int W, int C,
                               int max_disp, int stride2, int act, float slope, void* stream) {
    if (!in1 || !in2 || !out || N <= 0 || H <= 0 || W <= 0) return VPS_EARG(1);
    if (C <= 0 || (C & 3) || C > 1024 || (ld1 & 3) || (ld2 & 3) || (coff1 & 3) || (coff2 & 3)) return VPS_EARG(2);
    if (stride2 <= 0 || max_disp < 0) return VPS_EARG(3);
    const int r = max_disp / stride2;
    const long npix = (long)N * H * W;
    long g = (npix + 3) / 4; if (g > 65536) g = 65536;
    hipStream_t s = (hipStream_t)stream;
#define CORR4_LAUNCH(S2, R, NC4)                                                                                       \
    hipLaunchKernelGGL((correlation4_kernel<S2, R, NC4>), dim3((unsigned)g4), dim3(256), 0, s, in1, ld1, coff1, in2, ld2, \
                       coff2, out, out_ld, out_coff, N, H, W, C, act, slope)
    long g4 = ((long)N * H * (W / 4) + 3) / 4; if (g4 > 65536) g4 = 65536;
    // lane-per-pixel kernel (LDS-transposed tiles, no cross-lane reduction) when the row splits into whole 64 P position segments
    static const bool lds_off = getenv("VPS_CORR_LDS") && getenv("VPS_CORR_LDS")[0] == '0';        // A/B switch
    const bool small4g = (size_t)N * H * W * ld1 * 4 < 0xFFFFFFF0ull && (size_t)N * H * W * ld2 * 4 < 0xFFFFFFF0ull && (C & 31) == 0;
    if (!lds_off && small4g && stride2 == 2 && r == 10 && W % 256 == 0) {
        const long nb = (long)N * H * 2 * (W / 256) * 6;
        hipLaunchKernelGGL((correlation_lds_kernel<2, 10>), dim3((unsigned)nb), dim3(256), 0, s, in1, ld1, coff1, in2, ld2, coff2, out, out_ld,
                           out_coff, N, H, W, C, act, slope);
        return vps_launch_status();
    }
    // the 81-channel (9 x 9) case stays on the channel-lane kernel below: with only 9 displacements per row the lane-per-pixel kernel
    // reads 10 B values from LDS per 18 multiply-adds and is LDS-bandwidth bound (measured 409 us against 271 us); VPS_CORR_LDS=2
    // forces it (A/B)
    static const bool lds_all = getenv("VPS_CORR_LDS") && getenv("VPS_CORR_LDS")[0] == '2';
    if (lds_all && small4g && stride2 == 1 && r == 4 && W % 128 == 0) {
        const long nb = (long)N * H * (W / 128) * 3;
        hipLaunchKernelGGL((correlation_lds_kernel<1, 4>), dim3((unsigned)nb), dim3(256), 0, s, in1, ld1, coff1, in2, ld2, coff2, out, out_ld,
                           out_coff, N, H, W, C, act, slope);
        return vps_launch_status();
    }
    if (stride2 == 2 && r == 10 && (W % 8) == 0 && C <= 256) { CORR4_LAUNCH(2, 10, 1); return vps_launch_status(); }
    // stride 1, radius 4 (LiteFlowNetCorr): half a wavefront per dot product, two displacement rows per wavefront (corr_half.hip, round 6)
    if (vpsi_launch_corr4h(in1, ld1, coff1, in2, ld2, coff2, out, out_ld, out_coff, N, H, W, C, r, stride2, act, slope, s)) return vps_launch_status();
    if (stride2 == 1 && r == 4 && (W % 4) == 0 && C <= 256) { CORR4_LAUNCH(1, 4, 1); return vps_launch_status(); }
#undef CORR4_LAUNCH
#define CORR_LAUNCH(NC4)                                                                                        \
    hipLaunchKernelGGL((correlation_kernel<NC4>), dim3((unsigned)g), dim3(256), 0, s, in1, ld1, coff1, in2, ld2, \
                       coff2, out, out_ld, out_coff, N, H, W, C, r, stride2, act, slope)
    if (C <= 256) CORR_LAUNCH(1);
    else if (C <= 512) CORR_LAUNCH(2);
    else CORR_LAUNCH(4);
#undef CORR_LAUNCH
    return vps_launch_status();
}

extern "C" int vps_flow_warp(const float* in, int in_ld, int in_coff, const float* flow, int flow_ld, int flow_coff,
                             float* out, int out_ld, int out_coff, int N, int H, int W, int C, void* stream) {
    if (!in || !flow || !out || N <= 0 || H <= 1 || W <= 1) return VPS_EARG(1);
    if (C <= 0 || (C & 3) || (in_ld & 3) || (in_coff & 3) || (out_ld & 3) || (out_coff & 3)) return VPS_EARG(2);
    hipLaunchKernelGGL(flow_warp_kernel, dim3(stream_grid((long)N * H * W * (C >> 2), 256)), dim3(256), 0,
                       (hipStream_t)stream, in, in_ld, in_coff, flow, flow_ld, flow_coff, out, out_ld, out_coff, N, H, W, C);
    return vps_launch_status();
}

extern "C" int vps_nchw_to_nhwc(const float* in, float* out, int out_ld, int out_coff, int N, int C, int H, int W,
                                int Cpad, void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C) return VPS_EARG(1);
    dim3 grid(cdiv((long)H * W, 32), cdiv(Cpad, 32), N);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, out_ld, out_coff, N, C, H, W, Cpad);
    return vps_launch_status();
}

extern "C" int vps_nhwc_to_nchw(const float* in, int in_ld, int in_coff, float* out, int N, int C, int H, int W, void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return VPS_EARG(1);
    dim3 grid(cdiv((long)H * W, 32), cdiv(C, 32), N);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, in_ld, in_coff, out, N, C, H, W);
    return vps_launch_status();
}

extern "C" int vps_flow_prep_pad(const float* img, const float* ref, const float* mean3, const float* std3,
                                 float* out, int out_ld, int H, int W, int Hp, int Wp, double* partial, int nblk, float* rgb_mean_out,
                                 void* stream) {
    if (!img || !ref || !mean3 || !std3 || !out || !partial || !rgb_mean_out) return VPS_EARG(1);
    if (out_ld < 6 || nblk <= 0 || nblk > 2048 || H <= 0 || W <= 0 || Hp < H || Wp < W) return VPS_EARG(2);
    hipStream_t s = (hipStream_t)stream;
    const long HW = (long)H * W, HWp = (long)Hp * Wp;
    hipLaunchKernelGGL(flow_prep_sum_kernel, dim3(nblk), dim3(256), 0, s, img, ref, mean3, std3, HW, partial);
    hipLaunchKernelGGL(flow_prep_mean_kernel, dim3(1), dim3(192), 0, s, partial, nblk, HWp, rgb_mean_out);
    hipLaunchKernelGGL(flow_prep_write_kernel, dim3(stream_grid(HWp, 256)), dim3(256), 0, s, img, ref, mean3, std3,
                       rgb_mean_out, out, out_ld, H, W, Hp, Wp);
    return vps_launch_status();
}

extern "C" int vps_flow_prep(const float* img, const float* ref, const float* mean3, const float* std3,
                             float* out, int out_ld, int H, int W, double* partial, int nblk, float* rgb_mean_out,
                             void* stream) {
    return vps_flow_prep_pad(img, ref, mean3, std3, out, out_ld, H, W, H, W, partial, nblk, rgb_mean_out, stream);
}

extern "C" int vps_flow_stage_full(const float* x6, int x_ld, const float* flow_a, int a_ld, int a_coff,
                                   const float* flow_b, int b_ld, int b_coff, int H, int W, int mode, float mul,
                                   float* out, int out_ld, void* stream) {
    if (!x6 || !flow_a || !out || H <= 0 || W <= 0 || (H & 3) || (W & 3) || mul == 0.f) return VPS_EARG(1);
    if (x_ld != 8 || out_ld != 12 || (((uintptr_t)x6 | (uintptr_t)out) & 15)) return VPS_EARG(2);
    if (mode == 0) {
        hipLaunchKernelGGL(flow_stage_s_kernel, dim3(stream_grid((long)H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                           x6, flow_a, a_ld, a_coff, H, W, mul, out);
    } else if (mode == 1) {
        if (!flow_b) return VPS_EARG(3);
        hipLaunchKernelGGL(flow_stage_f_kernel, dim3(stream_grid((long)H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                           x6, flow_a, a_ld, a_coff, flow_b, b_ld, b_coff, H, W, mul, out);
    } else {
        return VPS_EARG(4);
    }
    return vps_launch_status();
}

extern "C" int vps_flow_stage(const float* x6, int x_ld, const float* flow_lo, int flo_ld, int flo_coff,
                              int H, int W, int up_mode, float mul, int div_mode,
                              float* out, int out_ld, int flow_off, float flow_out_div, int warp_off,
                              int diffnorm_off, int flownorm_off, int img_off, void* stream) {
    if (!x6 || !flow_lo || !out || H <= 0 || W <= 0 || (H & 3) || (W & 3)) return VPS_EARG(1);
    hipLaunchKernelGGL(flow_stage_kernel, dim3(stream_grid((long)H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                       x6, x_ld, flow_lo, flo_ld, flo_coff, H, W, up_mode, mul, div_mode, out, out_ld, flow_off,
                       flow_out_div, warp_off, diffnorm_off, flownorm_off, img_off);
    return vps_launch_status();
}
