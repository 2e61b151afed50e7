// NHWC resize / pool / pyramid gather-scatter / GroupNorm / TCEA attention kernels (all HBM-bound).
// Reference files are cited relative to /root/reference/mmdet/models.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ------------------------------------------------------------------------------------------------
// F.interpolate bilinear (align_corners=False) / nearest, as ATen's upsample_*2d CPU kernels compute them.
// ------------------------------------------------------------------------------------------------
// V = 4: float4 over channels (C, leading dimensions and channel offsets multiples of 4), V = 1: any layout.
template <int V> struct vecn { typedef float type; };
template <> struct vecn<4> { typedef f32x4 type; };

template <int V>
__global__ __launch_bounds__(256)
void resize_kernel(const float* __restrict__ in, int in_ld, int in_coff, int Hi, int Wi,
                   float* __restrict__ out, int out_ld, int out_coff, int Ho, int Wo,
                   int N, int C, int mode, float alpha, float sh, float sw) {
    typedef typename vecn<V>::type vt;
    const int cv = C / V;
    const long total = (long)N * Ho * Wo * cv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv) * V;
        const int pix = (int)(idx / cv);
        const int x = pix % Wo;
        const int yy = pix / Wo;
        const int y = yy % Ho, n = yy / Ho;
        const float* ib = in + (size_t)n * Hi * Wi * in_ld + in_coff + c;
        vt v;
        if (mode == 1) {
            const int ys = min((int)floorf((float)y * sh), Hi - 1);
            const int xs = min((int)floorf((float)x * sw), Wi - 1);
            v = *reinterpret_cast<const vt*>(ib + ((size_t)ys * Wi + xs) * in_ld);
        } else {
            float sy = sh * ((float)y + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
            float sx = sw * ((float)x + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
            const int y0 = min((int)sy, Hi - 1), x0 = min((int)sx, Wi - 1);
            const int yp = y0 < Hi - 1 ? 1 : 0, xp = x0 < Wi - 1 ? 1 : 0;
            const float ly = sy - (float)y0, lx = sx - (float)x0;
            const float hy = 1.f - ly, hx = 1.f - lx;
            const vt p00 = *reinterpret_cast<const vt*>(ib + ((size_t)y0 * Wi + x0) * in_ld);
            const vt p01 = *reinterpret_cast<const vt*>(ib + ((size_t)y0 * Wi + x0 + xp) * in_ld);
            const vt p10 = *reinterpret_cast<const vt*>(ib + ((size_t)(y0 + yp) * Wi + x0) * in_ld);
            const vt p11 = *reinterpret_cast<const vt*>(ib + ((size_t)(y0 + yp) * Wi + x0 + xp) * in_ld);
            v = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
        }
        *reinterpret_cast<vt*>(out + (size_t)pix * out_ld + out_coff + c) = v * alpha;
    }
}

// 3x3 stride 2 pad 1 max / avg (count_include_pad) pooling. All nine loads are issued unconditionally at clamped
// coordinates (a duplicate never changes a max; the average masks them afterwards): no load waits behind a branch.
template <int V>
__global__ __launch_bounds__(256)
void pool3x3s2_kernel(const float* __restrict__ in, int in_ld, int in_coff, int Hi, int Wi,
                      float* __restrict__ out, int out_ld, int out_coff, int Ho, int Wo, int N, int C, int mode) {
    typedef typename vecn<V>::type vt;
    const int cv = C / V;
    const long total = (long)N * Ho * Wo * cv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv) * V;
        const int pix = (int)(idx / cv);
        const int x = pix % Wo;
        const int yy0 = pix / Wo;
        const int y = yy0 % Ho, n = yy0 / Ho;
        const float* ib = in + (size_t)n * Hi * Wi * in_ld + in_coff + c;
        vt t[9];
        bool ok[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = 2 * y + k / 3 - 1, xx = 2 * x + k % 3 - 1;
            ok[k] = (unsigned)yy < (unsigned)Hi && (unsigned)xx < (unsigned)Wi;
            const int yc = min(max(yy, 0), Hi - 1), xc = min(max(xx, 0), Wi - 1);
            t[k] = *reinterpret_cast<const vt*>(ib + ((size_t)yc * Wi + xc) * in_ld);
        }
        vt m = t[4], s = vt{}, zero = vt{};   // the centre tap is always inside
#pragma unroll
        for (int k = 0; k < 9; ++k) {     // same visiting order as the reference loops (dy outer, dx inner)
            if constexpr (V == 1) { m = fmaxf(m, t[k]); }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], t[k][e]);
            }
            s += ok[k] ? t[k] : zero;
        }
        *reinterpret_cast<vt*>(out + (size_t)pix * out_ld + out_coff + c) = mode == 0 ? m : s / 9.0f;
    }
}

// extra_necks/bfp_tcea.py:96-109 with refine_level=0: every level is nearest-resized to level-0 size,
// summed in level order and divided by the level count. One pass, float4 over channels.
struct GatherArgs {
    const float* lv[5];
    int ld[5];
    int ratio[5];  // H0 / H_l
    int n;
};

__global__ __launch_bounds__(256)
void bfp_gather_kernel(GatherArgs a, float* __restrict__ out, int out_ld, int out_coff, int N, int H0, int W0, int C) {
    const int c4n = C >> 2;
    const long total = (long)N * H0 * W0 * c4n;
    const float div = (float)a.n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const long pix = idx / c4n;
        const int x = (int)(pix % W0);
        const int y = (int)((pix / W0) % H0);
        const int n = (int)(pix / ((long)W0 * H0));
        // all (up to 5) level values are requested first (absent levels re-read level 0), then summed in level order
        f32x4 v[5];
#pragma unroll
        for (int l = 0; l < 5; ++l) {
            const int ll = l < a.n ? l : 0;
            const int r = a.ratio[ll];
            const int Hl = H0 / r, Wl = W0 / r;
            v[l] = *reinterpret_cast<const f32x4*>(a.lv[ll] + (((size_t)n * Hl + y / r) * Wl + x / r) * a.ld[ll] + 4 * c4);
        }
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l < 5; ++l) s = l < a.n ? s + v[l] : s;
        *reinterpret_cast<f32x4*>(out + (size_t)pix * out_ld + out_coff + 4 * c4) = s / div;
    }
}

// extra_necks/bfp_tcea.py:139-147 with refine_level=0: adaptive_max_pool2d(bsf, size_l) (exact 2^l windows) + inputs[l]
__global__ __launch_bounds__(256)
void bfp_scatter_kernel(const float* __restrict__ bsf, int bsf_ld, const float* __restrict__ level, int lvl_ld,
                        float* __restrict__ out, int out_ld, int N, int H0, int W0, int C, int r) {
    const int c4n = C >> 2;
    const int Hl = H0 / r, Wl = W0 / r;
    const long total = (long)N * Hl * Wl * c4n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const long pix = idx / c4n;
        const int x = (int)(pix % Wl);
        const int y = (int)((pix / Wl) % Hl);
        const int n = (int)(pix / ((long)Wl * Hl));
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        // four window columns per step, requested together (a clamped duplicate does not change a maximum)
        for (int dy = 0; dy < r; ++dy)
            for (int dx0 = 0; dx0 < r; dx0 += 4) {
                f32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[j] = *reinterpret_cast<const f32x4*>(
                        bsf + (((size_t)n * H0 + y * r + dy) * W0 + x * r + min(dx0 + j, r - 1)) * bsf_ld + 4 * c4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    m[0] = fmaxf(m[0], v[j][0]); m[1] = fmaxf(m[1], v[j][1]); m[2] = fmaxf(m[2], v[j][2]); m[3] = fmaxf(m[3], v[j][3]);
                }
            }
        const f32x4 lv = *reinterpret_cast<const f32x4*>(level + (size_t)pix * lvl_ld + 4 * c4);
        *reinterpret_cast<f32x4*>(out + (size_t)pix * out_ld + 4 * c4) = m + lv;
    }
}

// The same for ALL levels of the pyramid in one pass (round 6): the five per-level launches read the full-resolution map five times
// (1.03 GB per frame at 256x512x256); here a block owns a 2^(L-1) x 2^(L-1) cell of it (16 x 16 for five levels) x 32 channels, reads
// it ONCE, writes level 0 and reduces 2 x 2 windows level by level through LDS - a maximum does not depend on the order it is taken
// in, so the outputs are bitwise those of bfp_scatter_kernel. Thread = (pixel, 4-channel group) with the group fastest: 8 threads
// cover the 128 bytes of a pixel's 32 channels.
struct ScatterAll {
    const float* lvl[5];
    float* out[5];
    int lvl_ld[5], out_ld[5];
};
template <int NL>
__global__ __launch_bounds__(256)
void bfp_scatter_all_kernel(const float* __restrict__ bsf, int bsf_ld, const ScatterAll a, int N, int H0, int W0, int C) {
    constexpr int T = 1 << (NL - 1);                 // cell edge at level 0
    __shared__ f32x4 buf[2][T * T * 8];              // [level parity][cell position][4-channel group]
    const int t = threadIdx.x, q = t & 7, p0 = t >> 3;
    const int cgroups = C >> 5, cells_x = W0 / T, cells_y = H0 / T;
    long b = blockIdx.x;
    const int cg = (int)(b % cgroups); b /= cgroups;
    const int cx = (int)(b % cells_x); b /= cells_x;
    const int cy = (int)(b % cells_y); const int n = (int)(b / cells_y);
    const int c = cg * 32 + 4 * q;
    // ---- level 0: T*T pixels, 32 per pass
    for (int p = p0; p < T * T; p += 32) {
        const int py = p / T, px = p - py * T;
        const size_t pix = ((size_t)n * H0 + cy * T + py) * W0 + cx * T + px;
        const f32x4 v = *reinterpret_cast<const f32x4*>(bsf + pix * bsf_ld + c);
        const f32x4 l = *reinterpret_cast<const f32x4*>(a.lvl[0] + pix * a.lvl_ld[0] + c);
        *reinterpret_cast<f32x4*>(a.out[0] + pix * a.out_ld[0] + c) = v + l;
        buf[0][p * 8 + q] = v;
    }
    __syncthreads();
    int e = T;                                       // edge of the source level inside the cell
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        const int eo = e >> 1, Hl = H0 >> l, Wl = W0 >> l;
        const f32x4* __restrict__ src = buf[(l - 1) & 1];
        f32x4* __restrict__ dst = buf[l & 1];
        for (int p = p0; p < eo * eo; p += 32) {
            const int py = p / eo, px = p - py * eo;
            const f32x4 v0 = src[((2 * py) * e + 2 * px) * 8 + q], v1 = src[((2 * py) * e + 2 * px + 1) * 8 + q];
            const f32x4 v2 = src[((2 * py + 1) * e + 2 * px) * 8 + q], v3 = src[((2 * py + 1) * e + 2 * px + 1) * 8 + q];
            f32x4 m;
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = fmaxf(fmaxf(v0[k], v1[k]), fmaxf(v2[k], v3[k]));
            const size_t pix = ((size_t)n * Hl + cy * eo + py) * Wl + cx * eo + px;
            const f32x4 lv = *reinterpret_cast<const f32x4*>(a.lvl[l] + pix * a.lvl_ld[l] + c);
            *reinterpret_cast<f32x4*>(a.out[l] + pix * a.out_ld[l] + c) = m + lv;
            dst[p * 8 + q] = m;
        }
        __syncthreads();
        e = eo;
    }
}

__global__ __launch_bounds__(256)
void axpb_kernel(const float* __restrict__ in, int in_ld, int in_coff, float* __restrict__ out, int out_ld, int out_coff,
                 long npix, int C, float a, float b) {
    const long total = npix * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const long pix = idx / C;
        out[(size_t)pix * out_ld + out_coff + c] = in[(size_t)pix * in_ld + in_coff + c] * a + b;
    }
}

// ------------------------------------------------------------------------------------------------
// GroupNorm(G)+ReLU, N=1 (panoptic/upsnetFPN.py:39-52). Pass 1: per-group sum / sum of squares in fp64
// (block partials -> one fp64 atomic per group per block). Pass 2: normalise, affine, ReLU.
// Requires C <= 256 and 256 % C == 0 (C = 256 or 128 here).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void groupnorm_stats_kernel(const float* __restrict__ in, int in_ld, long npix, int C, int G, double* __restrict__ stats) {
    __shared__ double ssum[256], ssq[256];
    const int t = threadIdx.x;
    const int c = t % C;
    const int ppb = 256 / C;   // pixels handled per block iteration
    const int sub = t / C;
    double s = 0.0, q = 0.0;
    for (long p = (long)blockIdx.x * ppb + sub; p < npix; p += (long)gridDim.x * ppb) {
        const float v = in[(size_t)p * in_ld + c];
        s += (double)v;
        q += (double)v * (double)v;
    }
    ssum[t] = s; ssq[t] = q;
    __syncthreads();
    const int cpg = C / G;
    if (t < G) {
        double gs = 0.0, gq = 0.0;
        for (int k = 0; k < ppb; ++k)
            for (int j = 0; j < cpg; ++j) {
                gs += ssum[k * C + t * cpg + j];
                gq += ssq[k * C + t * cpg + j];
            }
        atomicAdd(&stats[2 * t], gs);
        atomicAdd(&stats[2 * t + 1], gq);
    }
}

__global__ __launch_bounds__(256)
void groupnorm_apply_kernel(const float* __restrict__ in, int in_ld, float* __restrict__ out, int out_ld, int out_coff, long npix,
                            int C, int G, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                            int relu, const double* __restrict__ stats_g, int nrep) {
    // the sums may come as nrep partial copies (conv epilogue): add them up once per block
    __shared__ double stats[512];
    for (int i = threadIdx.x; i < 2 * G; i += blockDim.x) {
        double a = 0.0;
        for (int r = 0; r < nrep; ++r) a += stats_g[(size_t)r * 2 * G + i];
        stats[i] = a;
    }
    __syncthreads();
    const long total = npix * C;
    const int cpg = C / G;
    const double cnt = (double)npix * cpg;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const long pix = idx / C;
        const int g = c / cpg;
        const double mean = stats[2 * g] / cnt;
        double var = stats[2 * g + 1] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = rstd * gamma[c];
        const float bi = -sc * (float)mean + beta[c];
        float v = in[(size_t)pix * in_ld + c] * sc + bi;
        if (relu) v = v > 0.f ? v : 0.f;
        out[(size_t)pix * out_ld + out_coff + c] = v;
    }
}

// float4 variants (C, leading dimensions, offsets and channels-per-group multiples of 4: the layers of this path).
// One thread owns 4 consecutive channels = a whole number of groups never splits inside a float4 when cpg % 4 == 0 or
// cpg divides 4; the per-channel sums go through the same LDS table as the scalar kernel.
__global__ __launch_bounds__(256)
void groupnorm_stats4_kernel(const float* __restrict__ in, int in_ld, long npix, int C, int G, double* __restrict__ stats) {
    __shared__ double ssum[1024], ssq[1024];     // [pixel slot][channel], 256 threads x 4 channels
    const int t = threadIdx.x;
    const int c4n = C >> 2;
    const int c4 = t % c4n;
    const int ppb = 256 / c4n;   // pixels handled per block iteration
    const int sub = t / c4n;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (long p = (long)blockIdx.x * ppb + sub; p < npix; p += (long)gridDim.x * ppb) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + (size_t)p * in_ld + 4 * c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += (double)v[e]; q[e] += (double)v[e] * (double)v[e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { ssum[sub * C + 4 * c4 + e] = s[e]; ssq[sub * C + 4 * c4 + e] = q[e]; }
    __syncthreads();
    const int cpg = C / G;
    if (t < G) {
        double gs = 0.0, gq = 0.0;
        for (int k = 0; k < ppb; ++k)
            for (int j = 0; j < cpg; ++j) {
                gs += ssum[k * C + t * cpg + j];
                gq += ssq[k * C + t * cpg + j];
            }
        atomicAdd(&stats[2 * t], gs);
        atomicAdd(&stats[2 * t + 1], gq);
    }
}

__global__ __launch_bounds__(256)
void groupnorm_apply4_kernel(const float* __restrict__ in, int in_ld, float* __restrict__ out, int out_ld, int out_coff, long npix,
                             int C, int G, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                             int relu, const double* __restrict__ stats_g, int nrep) {
    // the sums may come as nrep partial copies (conv epilogue): add them up once per block
    __shared__ double stats[512];
    __shared__ float sc_t[256], bi_t[256];
    for (int i = threadIdx.x; i < 2 * G; i += blockDim.x) {
        double a = 0.0;
        for (int r = 0; r < nrep; ++r) a += stats_g[(size_t)r * 2 * G + i];
        stats[i] = a;
    }
    __syncthreads();
    const int c4n = C >> 2;
    const long total = npix * c4n;
    const int cpg = C / G;
    const double cnt = (double)npix * cpg;
    // per-channel scale / shift ONCE per block (C <= 256 entries): the fp64 divide + square root per ELEMENT made this pass 2x slower
    // than the memory system (125 us for 268 MB at 256x512x256; round 5). Same expressions in the same precisions: bit-identical output.
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        const double mean = stats[2 * g] / cnt;
        double var = stats[2 * g + 1] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = rstd * gamma[c];
        const float bi = -sc * (float)mean + beta[c];
        sc_t[c] = sc; bi_t[c] = bi;
    }
    __syncthreads();
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const long pix = idx / c4n;
        const f32x4 x = *reinterpret_cast<const f32x4*>(in + (size_t)pix * in_ld + c);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sc = sc_t[c + e], bi = bi_t[c + e];
            float v = x[e] * sc + bi;
            if (relu) v = v > 0.f ? v : 0.f;
            y[e] = v;
        }
        *reinterpret_cast<f32x4*>(out + (size_t)pix * out_ld + out_coff + c) = y;
    }
}

// ------------------------------------------------------------------------------------------------
// utils/tcea_modules.py:50-65 temporal attention for N=2 frames, center frame 0. One wavefront per pixel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void tcea_temporal_kernel(const float* __restrict__ emb, int emb_ld, const float* __restrict__ emb_ref, int ref_ld,
                          const float* __restrict__ fea0, int f0_ld, const float* __restrict__ fea1, int f1_ld,
                          float* __restrict__ out, int out_ld, long npix, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4n = C >> 2;
    for (long pix = (long)blockIdx.x * 4 + wave; pix < npix; pix += (long)gridDim.x * 4) {
        float cor0 = 0.f, cor1 = 0.f;
        for (int c4 = lane; c4 < c4n; c4 += 64) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(emb_ref + (size_t)pix * ref_ld + 4 * c4);
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(emb + (size_t)pix * emb_ld + 4 * c4);
            const f32x4 e1 = *reinterpret_cast<const f32x4*>(emb + (size_t)pix * emb_ld + C + 4 * c4);
            cor0 += e0[0] * r[0] + e0[1] * r[1] + e0[2] * r[2] + e0[3] * r[3];
            cor1 += e1[0] * r[0] + e1[1] * r[1] + e1[2] * r[2] + e1[3] * r[3];
        }
        for (int off = 32; off >= 1; off >>= 1) {
            cor0 += __shfl_xor(cor0, off, 64);
            cor1 += __shfl_xor(cor1, off, 64);
        }
        const float p0 = 1.f / (1.f + expf(-cor0));
        const float p1 = 1.f / (1.f + expf(-cor1));
        for (int c4 = lane; c4 < c4n; c4 += 64) {
            const f32x4 f0 = *reinterpret_cast<const f32x4*>(fea0 + (size_t)pix * f0_ld + 4 * c4);
            const f32x4 f1 = *reinterpret_cast<const f32x4*>(fea1 + (size_t)pix * f1_ld + 4 * c4);
            *reinterpret_cast<f32x4*>(out + (size_t)pix * out_ld + 4 * c4) = f0 * p0;
            *reinterpret_cast<f32x4*>(out + (size_t)pix * out_ld + C + 4 * c4) = f1 * p1;
        }
    }
}

// utils/tcea_modules.py:74-77: fea * sigmoid(att) * 2 + att_add
__global__ __launch_bounds__(256)
void tcea_modulate_kernel(const float* __restrict__ fea, const float* __restrict__ att, const float* __restrict__ att_add,
                          float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float s = 1.f / (1.f + expf(-att[i]));
        out[i] = fea[i] * s * 2.f + att_add[i];
    }
}

// the same on channel WINDOWS (leading dimensions, 4 | C): the fused feature lives in one half of the merged fea_fusion | sAtt_1 output
__global__ __launch_bounds__(256)
void tcea_modulate_ld_kernel(const float* __restrict__ fea, int fea_ld, const float* __restrict__ att, int att_ld,
                             const float* __restrict__ att_add, int add_ld, float* __restrict__ out, int out_ld, long npix, int C) {
    const int c4n = C >> 2;
    const long total = npix * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / c4n;
        const int c = (int)(i - pix * c4n) * 4;
        const f32x4 f = *reinterpret_cast<const f32x4*>(fea + (size_t)pix * fea_ld + c);
        const f32x4 a = *reinterpret_cast<const f32x4*>(att + (size_t)pix * att_ld + c);
        const f32x4 d = *reinterpret_cast<const f32x4*>(att_add + (size_t)pix * add_ld + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float s = 1.f / (1.f + expf(-a[e]));
            o[e] = f[e] * s * 2.f + d[e];
        }
        *reinterpret_cast<f32x4*>(out + (size_t)pix * out_ld + c) = o;
    }
}

}  // namespace

extern "C" int vps_resize(const float* in, int in_ld, int in_coff, int Hi, int Wi, float* out, int out_ld, int out_coff,
                          int Ho, int Wo, int N, int C, int mode, float alpha, void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return VPS_EARG(1);
    const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
    if ((long)N * Ho * Wo > 0x7fffffffL) return VPS_EARG(2);
    if (!((C | in_ld | in_coff | out_ld | out_coff) & 3))
        hipLaunchKernelGGL(resize_kernel<4>, dim3(stream_grid((long)N * Ho * Wo * (C >> 2), 256)), dim3(256), 0, (hipStream_t)stream,
                           in, in_ld, in_coff, Hi, Wi, out, out_ld, out_coff, Ho, Wo, N, C, mode, alpha, sh, sw);
    else
        hipLaunchKernelGGL(resize_kernel<1>, dim3(stream_grid((long)N * Ho * Wo * C, 256)), dim3(256), 0, (hipStream_t)stream,
                           in, in_ld, in_coff, Hi, Wi, out, out_ld, out_coff, Ho, Wo, N, C, mode, alpha, sh, sw);
    return vps_launch_status();
}

extern "C" int vps_pool3x3s2(const float* in, int in_ld, int in_coff, int Hi, int Wi, float* out, int out_ld, int out_coff,
                             int N, int C, int mode, void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || Hi <= 0 || Wi <= 0) return VPS_EARG(1);
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    if ((long)N * Ho * Wo > 0x7fffffffL) return VPS_EARG(2);
    if (!((C | in_ld | in_coff | out_ld | out_coff) & 3))
        hipLaunchKernelGGL(pool3x3s2_kernel<4>, dim3(stream_grid((long)N * Ho * Wo * (C >> 2), 256)), dim3(256), 0, (hipStream_t)stream,
                           in, in_ld, in_coff, Hi, Wi, out, out_ld, out_coff, Ho, Wo, N, C, mode);
    else
        hipLaunchKernelGGL(pool3x3s2_kernel<1>, dim3(stream_grid((long)N * Ho * Wo * C, 256)), dim3(256), 0, (hipStream_t)stream,
                           in, in_ld, in_coff, Hi, Wi, out, out_ld, out_coff, Ho, Wo, N, C, mode);
    return vps_launch_status();
}

extern "C" int vps_bfp_gather(const float* const* levels, const int* ld, const int* ratio, int nlevels,
                              float* out, int out_ld, int out_coff, int N, int H0, int W0, int C, void* stream) {
    if (!levels || !ld || !ratio || !out || nlevels < 1 || nlevels > 5 || (C & 3) || (out_ld & 3) || (out_coff & 3)) return VPS_EARG(1);
    GatherArgs a;
    a.n = nlevels;
    for (int l = 0; l < 5; ++l) {
        a.lv[l] = l < nlevels ? levels[l] : nullptr;
        a.ld[l] = l < nlevels ? ld[l] : 0;
        a.ratio[l] = l < nlevels ? ratio[l] : 1;
        if (l < nlevels && (!levels[l] || (ld[l] & 3) || ratio[l] < 1 || H0 % ratio[l] || W0 % ratio[l])) return VPS_EARG(2);
    }
    hipLaunchKernelGGL(bfp_gather_kernel, dim3(stream_grid((long)N * H0 * W0 * (C >> 2), 256)), dim3(256), 0,
                       (hipStream_t)stream, a, out, out_ld, out_coff, N, H0, W0, C);
    return vps_launch_status();
}

extern "C" int vps_bfp_scatter(const float* bsf, int bsf_ld, const float* level, int lvl_ld, float* out, int out_ld,
                               int N, int H0, int W0, int C, int ratio, void* stream) {
    if (!bsf || !level || !out || (C & 3) || (bsf_ld & 3) || (lvl_ld & 3) || (out_ld & 3) || ratio < 1 || H0 % ratio || W0 % ratio)
        return VPS_EARG(1);
    hipLaunchKernelGGL(bfp_scatter_kernel, dim3(stream_grid((long)N * (H0 / ratio) * (W0 / ratio) * (C >> 2), 256)),
                       dim3(256), 0, (hipStream_t)stream, bsf, bsf_ld, level, lvl_ld, out, out_ld, N, H0, W0, C, ratio);
    return vps_launch_status();
}

extern "C" int vps_bfp_scatter_all(const float* bsf, int bsf_ld, const float* const* levels, const int* lvl_ld, float* const* outs,
                                   const int* out_ld, int nlevels, int N, int H0, int W0, int C, void* stream) {
    if (!bsf || !levels || !lvl_ld || !outs || !out_ld || N <= 0 || H0 <= 0 || W0 <= 0) return VPS_EARG(1);
    // level l has ratio 2^l; whole cells; 32-channel groups; float4 rows
    if (nlevels < 2 || nlevels > 5 || (C & 31) || (bsf_ld & 3) || H0 % (1 << (nlevels - 1)) || W0 % (1 << (nlevels - 1))) return VPS_EARG(2);
    ScatterAll a;
    for (int l = 0; l < 5; ++l) {
        const int k = l < nlevels ? l : nlevels - 1;
        if (!levels[k] || !outs[k] || (lvl_ld[k] & 3) || (out_ld[k] & 3)) return VPS_EARG(3);
        a.lvl[l] = levels[k]; a.out[l] = outs[k]; a.lvl_ld[l] = lvl_ld[k]; a.out_ld[l] = out_ld[k];
    }
    const int T = 1 << (nlevels - 1);
    const long nblk = (long)N * (H0 / T) * (W0 / T) * (C >> 5);
    if (nblk > 0x7fffffffL) return VPS_EARG(4);
    hipStream_t s = (hipStream_t)stream;
    switch (nlevels) {
        case 5: hipLaunchKernelGGL(bfp_scatter_all_kernel<5>, dim3((unsigned)nblk), dim3(256), 0, s, bsf, bsf_ld, a, N, H0, W0, C); break;
        case 4: hipLaunchKernelGGL(bfp_scatter_all_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, s, bsf, bsf_ld, a, N, H0, W0, C); break;
        case 3: hipLaunchKernelGGL(bfp_scatter_all_kernel<3>, dim3((unsigned)nblk), dim3(256), 0, s, bsf, bsf_ld, a, N, H0, W0, C); break;
        default: hipLaunchKernelGGL(bfp_scatter_all_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, s, bsf, bsf_ld, a, N, H0, W0, C); break;
    }
    return vps_launch_status();
}

extern "C" int vps_axpb(const float* in, int in_ld, int in_coff, float* out, int out_ld, int out_coff, int64_t npix, int C,
                        float a, float b, void* stream) {
    if (!in || !out || npix <= 0 || C <= 0) return VPS_EARG(1);
    hipLaunchKernelGGL(axpb_kernel, dim3(stream_grid((long)npix * C, 256)), dim3(256), 0, (hipStream_t)stream,
                       in, in_ld, in_coff, out, out_ld, out_coff, (long)npix, C, a, b);
    return vps_launch_status();
}

extern "C" int vps_groupnorm_relu(const float* in, int in_ld, float* out, int out_ld, int out_coff, int64_t npix, int C, int G,
                                  const float* gamma, const float* beta, float eps, int relu, double* stats, void* stream) {
    if (!in || !out || !gamma || !beta || !stats || npix <= 0) return VPS_EARG(1);
    if (C <= 0 || C > 256 || 256 % C || G <= 0 || C % G || G > 256) return VPS_EARG(2);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * 2 * G, s);
    if (e != hipSuccess) return -(int)e;
    if (!((C | in_ld | out_ld | out_coff) & 3) && 1024 % C == 0) {
        const int ppb4 = 1024 / C;
        // <= 256 blocks: every block ends with 2 G fp64 atomics on the same 2 G addresses - with 1024 blocks the statistics of a 64x128
        // map (8 MB) took 40 us of atomic traffic (round 5)
        long g4 = (npix + ppb4 - 1) / ppb4; if (g4 > 256) g4 = 256;
        hipLaunchKernelGGL(groupnorm_stats4_kernel, dim3((unsigned)g4), dim3(256), 0, s, in, in_ld, (long)npix, C, G, stats);
        hipLaunchKernelGGL(groupnorm_apply4_kernel, dim3(stream_grid((long)npix * (C >> 2), 256)), dim3(256), 0, s, in, in_ld, out,
                           out_ld, out_coff, (long)npix, C, G, gamma, beta, eps, relu, stats, 1);
        return vps_launch_status();
    }
    const int ppb = 256 / C;
    long g = (npix + ppb - 1) / ppb; if (g > 1024) g = 1024;
    hipLaunchKernelGGL(groupnorm_stats_kernel, dim3((unsigned)g), dim3(256), 0, s, in, in_ld, (long)npix, C, G, stats);
    hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(stream_grid((long)npix * C, 256)), dim3(256), 0, s, in, in_ld, out, out_ld,
                       out_coff, (long)npix, C, G, gamma, beta, eps, relu, stats, 1);
    return vps_launch_status();
}

extern "C" int vps_groupnorm_apply(const float* in, int in_ld, float* out, int out_ld, int out_coff, int64_t npix, int C, int G,
                                   const float* gamma, const float* beta, float eps, int relu, const double* stats, int nrep, void* stream) {
    if (!in || !out || !gamma || !beta || !stats || npix <= 0) return VPS_EARG(1);
    if (C <= 0 || C > 256 || 256 % C || G <= 0 || C % G || G > 256 || nrep < 1) return VPS_EARG(2);
    hipStream_t s = (hipStream_t)stream;
    if (!((C | in_ld | out_ld | out_coff) & 3))
        hipLaunchKernelGGL(groupnorm_apply4_kernel, dim3(stream_grid((long)npix * (C >> 2), 256)), dim3(256), 0, s, in, in_ld, out,
                           out_ld, out_coff, (long)npix, C, G, gamma, beta, eps, relu, stats, nrep);
    else
        hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(stream_grid((long)npix * C, 256)), dim3(256), 0, s, in, in_ld, out, out_ld,
                           out_coff, (long)npix, C, G, gamma, beta, eps, relu, stats, nrep);
    return vps_launch_status();
}

extern "C" int vps_tcea_temporal(const float* emb, int emb_ld, const float* emb_ref, int ref_ld,
                                 const float* fea0, int f0_ld, const float* fea1, int f1_ld,
                                 float* out, int out_ld, int64_t npix, int C, void* stream) {
    if (!emb || !emb_ref || !fea0 || !fea1 || !out || npix <= 0 || C <= 0 || (C & 3)) return VPS_EARG(1);
    if ((emb_ld & 3) || (ref_ld & 3) || (f0_ld & 3) || (f1_ld & 3) || (out_ld & 3)) return VPS_EARG(2);
    if (((uintptr_t)emb | (uintptr_t)emb_ref | (uintptr_t)fea0 | (uintptr_t)fea1 | (uintptr_t)out) & 15) return VPS_EARG(3);
    long g = (npix + 3) / 4; if (g > 16384) g = 16384;
    hipLaunchKernelGGL(tcea_temporal_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, emb, emb_ld, emb_ref, ref_ld,
                       fea0, f0_ld, fea1, f1_ld, out, out_ld, (long)npix, C);
    return vps_launch_status();
}

extern "C" int vps_tcea_modulate_ld(const float* fea, int fea_ld, const float* att, int att_ld, const float* att_add, int add_ld,
                                    float* out, int out_ld, int64_t npix, int C, void* stream) {
    if (!fea || !att || !att_add || !out || npix <= 0 || C <= 0) return VPS_EARG(1);
    if ((C | fea_ld | att_ld | add_ld | out_ld) & 3 || (((uintptr_t)fea | (uintptr_t)att | (uintptr_t)att_add | (uintptr_t)out) & 15)) return VPS_EARG(2);
    hipLaunchKernelGGL(tcea_modulate_ld_kernel, dim3(stream_grid((long)npix * (C >> 2), 256)), dim3(256), 0, (hipStream_t)stream, fea, fea_ld,
                       att, att_ld, att_add, add_ld, out, out_ld, (long)npix, C);
    return vps_launch_status();
}

extern "C" int vps_tcea_modulate(const float* fea, const float* att, const float* att_add, float* out, int64_t n, void* stream) {
    if (!fea || !att || !att_add || !out || n <= 0) return VPS_EARG(1);
    hipLaunchKernelGGL(tcea_modulate_kernel, dim3(stream_grid((long)n, 256)), dim3(256), 0, (hipStream_t)stream, fea, att,
                       att_add, out, (long)n);
    return vps_launch_status();
}
