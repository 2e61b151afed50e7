// The narrow-output kernels of vps_conv2d (cout <= 4 in exact fp32 on the vector ALU: the FlowNet predict_flow / upsampled_flow layers, the RPN
// objectness layer). Own translation unit since round 6; entered through vpsi_launch_conv_small from launch_conv (conv_mfma.hip), which
// decides WHEN a layer comes here (cout <= 4, exact-fp32 descriptor, no offsets, no split-K, 32-column packing).
#include "conv_common.h"
#include <cstdlib>

namespace {

// ================================================================================================
// Narrow-output convolution (cout <= 4: the FlowNet predict_flow / upsampled_flow layers, 2 channels). A 32-column MFMA
// tile would waste 94 % of the matrix pipe and still stage the whole activation tile through LDS; these layers are
// pure activation streaming, so they run on the vector ALU in exact fp32: G lanes (G = pow2 >= cin_pad/4, <= 64) share
// one output pixel and stride its channels with float4 loads, all taps accumulate in registers, one G-lane shuffle
// reduction per pixel. Same descriptor, k ordering, parity classes and epilogue as the MFMA kernels.
// ================================================================================================
template <int CO>
__global__ __launch_bounds__(256)
void conv_small_kernel(const vps_conv_desc d, const int M, const int G, const int logG) {
    const int lane = threadIdx.x & 63;
    const int sub = lane & (G - 1);                       // channel slot inside the pixel group
    const int ppw = 64 >> logG;                           // pixels per wavefront
    const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    const long total = (long)d.nclass * M;
    const int H = d.H, W = d.W, KH = d.KH, KW = d.KW, cin_pad = d.cin_pad, ntap = KH * KW;
    const int c4n = cin_pad >> 2;
    for (long base = wave_id * ppw; base < total; base += nwaves * ppw) {
        const long idx = base + (lane >> logG);
        const bool pv = idx < total;
        const int cls = pv ? (int)(idx / M) : 0;
        const int m = pv ? (int)(idx - (long)cls * M) : 0;
        const int py = cls / d.os_x, px = cls - py * d.os_x;
        const int qx = m % d.Qw;
        const int tq = m / d.Qw;
        const int qy = tq % d.Qh;
        const int n = tq / d.Qh;
        const int iy0 = qy * d.stride - d.pad_y[py], ix0 = qx * d.stride - d.pad_x[px];
        const float* __restrict__ wcls = d.w + (size_t)cls * d.cout_pad * d.kpad;
        float acc[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] = 0.f;
        for (int tap = 0; tap < ntap; ++tap) {
            const int ky = tap / KW, kx = tap - ky * KW;
            const int iy = iy0 + ky, ix = ix0 + kx;
            const bool ok = pv && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const float* __restrict__ ap = d.in + ((size_t)(n * H * W + (ok ? iy * W + ix : 0)) * d.in_ld + d.in_coff);
            for (int c4 = sub; c4 < c4n; c4 += G) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 a = ok ? *reinterpret_cast<const f32x4*>(ap + 4 * c4) : z;
                const int ci = 4 * c4;
                const int k = d.korder == 0 ? tap * cin_pad + ci : ((ci >> 5) * ntap + tap) * 32 + (ci & 31);
#pragma unroll
                for (int c = 0; c < CO; ++c) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wcls + (size_t)c * d.kpad + k);
                    acc[c] += a[0] * wv[0] + a[1] * wv[1] + a[2] * wv[2] + a[3] * wv[3];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CO; ++c)
            for (int off = G >> 1; off >= 1; off >>= 1) acc[c] += __shfl_xor(acc[c], off, 64);
        if (pv && sub == 0) {
            const int oy = qy * d.os_y + py, ox = qx * d.os_x + px;
            const size_t opix = ((size_t)n * d.Ho + oy) * d.Wo + ox;
            const int rs = d.res_shift;
            const size_t rpix = ((size_t)n * (d.Ho >> rs) + (oy >> rs)) * (d.Wo >> rs) + (ox >> rs);
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                if (c < d.cout) {
                    float v = acc[c] * (d.scale ? d.scale[c] : 1.f) + (d.shift ? d.shift[c] : 0.f);
                    if (d.res) v += d.res[rpix * d.res_ld + d.res_coff + c];
                    d.out[opix * d.out_ld + d.out_coff + c] = vps_act(v, d.act, d.slope);
                }
            }
        }
    }
}

// The same layers organised for the memory pipe (round 6): in conv_small_kernel a lane has ONE activation load in flight and the 64 lanes
// of a 256-channel pixel pay 6 shuffle steps per output channel for 4 multiply-adds each - the RPN objectness layer (256 -> 3, 1x1,
// 256 x 512) ran at 1.4 TB/s, the 2 -> 2 up-flow layers at four dependent latencies per pixel. Here the (tap, channel slot) pairs of a
// lane are flattened into one index and requested EIGHT at a time before the first is used (buffer-addressed: out-of-image taps, channel
// pads and idle lanes are masked by the offset), G is chosen by the host so that a lane carries about eight loads (256 -> 3: 8 lanes per
// pixel, 8 pixels per wavefront, 3 shuffle steps), and the CO x kpad weights of every parity class sit in LDS (<= 48 KB, else the
// kernel above). Per lane the order of accumulation is the one of conv_small_kernel with the same G.
template <int CO>
__global__ __launch_bounds__(256)
void conv_small_batched_kernel(const vps_conv_desc d, const int M, const int G, const int logG, const int nslot) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];   // [nclass][CO][kpad]
    const int kpad = d.kpad;
    for (int i = threadIdx.x * 4; i < d.nclass * CO * kpad; i += 256 * 4) {
        const int row = i / kpad, k = i - row * kpad;
        const int cls = row / CO, co = row - cls * CO;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(&wsm[i]) = co < d.cout_pad ? *reinterpret_cast<const f32x4*>(d.w + ((size_t)cls * d.cout_pad + co) * kpad + k) : z;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int sub = lane & (G - 1);
    const int ppw = 64 >> logG;
    const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    const long total = (long)d.nclass * M;
    const int H = d.H, W = d.W, KW = d.KW, cin_pad = d.cin_pad, ntap = d.KH * d.KW;
    const int c4n = cin_pad >> 2;
    const int nj = ntap * nslot;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(d.in, (unsigned)((size_t)d.N * H * W * d.in_ld * sizeof(float)));
    const unsigned ld4 = (unsigned)d.in_ld * 4u;
    for (long base = wave_id * ppw; base < total; base += nwaves * ppw) {
        const long idx = base + (lane >> logG);
        const bool pv = idx < total;
        const int cls = pv ? (int)(idx / M) : 0;
        const int m = pv ? (int)(idx - (long)cls * M) : 0;
        const int py = cls / d.os_x, px = cls - py * d.os_x;
        const int qx = m % d.Qw;
        const int tq = m / d.Qw;
        const int qy = tq % d.Qh;
        const int n = tq / d.Qh;
        const int iy0 = qy * d.stride - d.pad_y[py], ix0 = qx * d.stride - d.pad_x[px];
        const float* __restrict__ wcls = wsm + cls * CO * kpad;
        float acc[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] = 0.f;
        int tap = 0, slot = 0;
        for (int j0 = 0; j0 < nj; j0 += 8) {
            f32x4 a[8];
            int kk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ky = tap / KW, kx = tap - ky * KW;
                const int iy = iy0 + ky, ix = ix0 + kx;
                const int c4 = sub + (slot << logG);
                const bool ok = pv && j0 + u < nj && c4 < c4n && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const int ci = ok ? 4 * c4 : 0;
                a[u] = buffer_load16<f32x4>(rsrc, ok ? (unsigned)((n * H + iy) * W + ix) * ld4 + (unsigned)(d.in_coff + ci) * 4u : 0xFFFFFFF0u, 0u);
                kk[u] = ok ? (d.korder == 0 ? tap * cin_pad + ci : ((ci >> 5) * ntap + tap) * 32 + (ci & 31)) : 0;   // a masked load is zeros: any weight will do
                if (++slot == nslot) { slot = 0; ++tap; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int c = 0; c < CO; ++c) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wcls + c * kpad + kk[u]);
                    acc[c] += a[u][0] * wv[0] + a[u][1] * wv[1] + a[u][2] * wv[2] + a[u][3] * wv[3];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CO; ++c)
            for (int off = G >> 1; off >= 1; off >>= 1) acc[c] += __shfl_xor(acc[c], off, 64);
        if (pv && sub == 0) {
            const int oy = qy * d.os_y + py, ox = qx * d.os_x + px;
            const size_t opix = ((size_t)n * d.Ho + oy) * d.Wo + ox;
            const int rs = d.res_shift;
            const size_t rpix = ((size_t)n * (d.Ho >> rs) + (oy >> rs)) * (d.Wo >> rs) + (ox >> rs);
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                if (c < d.cout) {
                    float v = acc[c] * (d.scale ? d.scale[c] : 1.f) + (d.shift ? d.shift[c] : 0.f);
                    if (d.res) v += d.res[rpix * d.res_ld + d.res_coff + c];
                    d.out[opix * d.out_ld + d.out_coff + c] = vps_act(v, d.act, d.slope);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Narrow-output 3x3 stride-1 convolution (the predict_flow layers): same arithmetic as conv_small_kernel, organised for the memory pipe
// AND for its instruction count. The CO x kpad weights sit in LDS (loaded once per workgroup, the CO weights of a k side by side so that
// a 16-byte read holds the CO weights of two consecutive k); every G-lane group walks a horizontal run of RUN output pixels whose RUN+2 input
// columns (float4 channel slices of 3 rows) are all requested up front - each activation is loaded 4.5 times instead of 9 and a run costs
// one memory latency instead of 9 per pixel. Out-of-image taps, channel pads and idle lanes are masked by the ADDRESS (an offset beyond
// the buffer reads zeros). The G-lane reduction is a folding butterfly (every exchange halves the values a lane carries).
// Round 6 rebuilt it around the instruction count: its predecessor issued ~1150 wave instructions per run for 288 multiply-adds
// (64-bit run / row decoding by division ~360; the loads sunk towards their uses by the scheduler - 4 + 2 + 2 ... with vmcnt(0) in
// between; `?:` on the accumulator array turned into indexed extracts = chains of 8 compares + selects per value; G a run-time value;
// scale / shift pointers tested and read per stored value). A SIMD issues one vector instruction per 4 cycles whatever the occupancy:
// `194->2 @256x512` = 32 runs per SIMD x 1150 x 4 cycles = 61 us at 2.4 GHz; it measured 55 us (101 MB at 1.9 TB/s). Now: 32-bit run index
// advanced incrementally (one division per thread, none per run), G a template parameter, one base offset per run + row / column
// strides, a scheduling fence behind the load block, bit selects in the folds, scale / shift of the lane's values read once: ~650
// instructions per run (the library is built without packed FP32, DESIGN.md 3.3: the two-channel FMAs below are two v_fmac each), 55 -> 35 us (`16->2 @1024x2048` 71 -> 42 us), frame +0.45 % (A/B in one call). 3 waves per SIMD (154 VGPRs).
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bit_select(const unsigned m, const float a, const float b) {   // m all ones: a, zero: b
    return __uint_as_float((m & __float_as_uint(a)) | (~m & __float_as_uint(b)));
}

template <int CO, int RUN, int LOGG>
__global__ __launch_bounds__(256, 3)
void conv_small3x3v_kernel(const vps_conv_desc d, const int runs_per_row, const int total_runs) {
    extern __shared__ __attribute__((aligned(16))) float wl[];     // [kpad][CO]
    constexpr int G = 1 << LOGG, NV = RUN * CO, NFOLD = LOGG < 3 ? LOGG : 3, CNT = NV >> NFOLD, CP = CO / 2;
    static_assert(NV == 8 && (CO == 2 || CO == 4), "8 accumulators per lane");
    const int t = threadIdx.x;
    const int kpad = d.kpad;
    for (int i = t * 4; i < CO * kpad; i += 256 * 4) {
        const int co = i / kpad, k = i - co * kpad;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 v = co < d.cout_pad ? *reinterpret_cast<const f32x4*>(d.w + (size_t)co * kpad + k) : z;
#pragma unroll
        for (int e = 0; e < 4; ++e) wl[(k + e) * CO + co] = v[e];
    }
    __syncthreads();
    const f32x4* __restrict__ wl4 = reinterpret_cast<const f32x4*>(wl);

    const int lane = t & 63;
    const int sub = lane & (G - 1);
    constexpr int ppw = 64 >> LOGG;
    const int ngroups = (int)gridDim.x * 4 * ppw;
    const int wave0 = ((int)blockIdx.x * 4 + (t >> 6)) * ppw;
    const int H = d.H, W = d.W, cin_pad = d.cin_pad, c4n = cin_pad >> 2;
    const int nslot = (c4n + G - 1) >> LOGG;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(d.in, (unsigned)((size_t)d.N * H * W * d.in_ld * sizeof(float)));
    const unsigned ld4 = (unsigned)d.in_ld * 4u, rstride = (unsigned)W * ld4;
    const int kstep = d.korder == 0 ? cin_pad : 32;

    // the values this lane holds after the folds: indices mine .. mine + CNT - 1 of [RUN][CO]; their scale / shift once
    int mine = 0;
#pragma unroll
    for (int s = 0; s < NFOLD; ++s) mine += (lane >> s & 1) ? (NV >> 1) >> s : 0;
    float sc[CNT], sh[CNT];
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int c = (mine + i) & (CO - 1);
        sc[i] = (d.scale && c < d.cout) ? d.scale[c] : 1.f;
        sh[i] = (d.shift && c < d.cout) ? d.shift[c] : 0.f;
    }
    const bool writer = (sub >> NFOLD) == 0;

    // run -> (column run, row, image), advanced by ngroups per round
    int run = wave0 + (lane >> LOGG);
    int xr = run % runs_per_row, y, n;
    { const int ty = run / runs_per_row; y = ty % H; n = ty / H; }
    const int dq = ngroups / runs_per_row, dr = ngroups - dq * runs_per_row;

    for (int wrun = wave0; wrun < total_runs; wrun += ngroups) {     // scalar condition: whole wavefronts iterate together (shuffles below)
        const bool rv = run < total_runs;
        const int x0 = xr * RUN;
        f32x2 acc[RUN][CP];
#pragma unroll
        for (int xi = 0; xi < RUN; ++xi)
#pragma unroll
            for (int c = 0; c < CP; ++c) acc[xi][c] = f32x2{0.f, 0.f};
        const unsigned base = (unsigned)((n * H + y) * W + x0) * ld4 + (unsigned)d.in_coff * 4u - rstride - ld4;   // (row y - 1, column x0 - 1); wraps are masked
        const bool rok[3] = {rv && y >= 1, rv, rv && y + 1 < H};
        bool cok[RUN + 2];
#pragma unroll
        for (int j = 0; j < RUN + 2; ++j) cok[j] = (unsigned)(x0 - 1 + j) < (unsigned)W;

        for (int slot = 0; slot < nslot; ++slot) {
            const int c4 = sub + (slot << LOGG);
            const bool cv = c4 < c4n;
            const int ci = cv ? 4 * c4 : 0;
            const int kbase = d.korder == 0 ? ci : ((ci >> 5) * 9) * 32 + (ci & 31);
            const unsigned b0 = base + (unsigned)ci * 4u;
            // all RUN+2 columns of the 3 rows are requested before the first value is used; out-of-image taps, channel pads and idle
            // lanes are masked by the ADDRESS (an offset beyond the buffer reads zeros)
            f32x4 col[RUN + 2][3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const bool rk = rok[ky] && cv;
#pragma unroll
                for (int j = 0; j < RUN + 2; ++j)
                    col[j][ky] = buffer_load16<f32x4>(rsrc, (rk && cok[j]) ? b0 + (unsigned)ky * rstride + (unsigned)j * ld4 : 0xFFFFFFF0u, 0u);
            }
            __builtin_amdgcn_sched_barrier(0);      // the scheduler would sink the loads towards their uses (see the kernel above)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int k = kbase + (ky * 3 + kx) * kstep;           // multiple of 4
                    f32x4 wv[CO];                                            // [4 k][CO] floats = CO 16-byte reads
#pragma unroll
                    for (int q = 0; q < CO; ++q) wv[q] = wl4[((k * CO) >> 2) + q];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int c = 0; c < CP; ++c) {
                            const int f = e * CO + 2 * c;                    // float index of (k + e, channel 2c) in wv
                            const f32x2 w2 = {wv[f >> 2][f & 3], wv[f >> 2][(f & 3) + 1]};
#pragma unroll
                            for (int xi = 0; xi < RUN; ++xi) {
                                const float a = col[xi + kx][ky][e];
                                acc[xi][c] = __builtin_elementwise_fma(f32x2{a, a}, w2, acc[xi][c]);
                            }
                        }
                }
        }
        // reduction over the G lanes of the group: the folding butterfly over the low lane bits (every exchange halves the number of
        // values a lane carries), plain exchanges of what is left over the remaining bits
        float v[NV];
#pragma unroll
        for (int xi = 0; xi < RUN; ++xi)
#pragma unroll
            for (int c = 0; c < CO; ++c) v[xi * CO + c] = acc[xi][c >> 1][c & 1];
#pragma unroll
        for (int s = 0; s < NFOLD; ++s) {
            const int off = 1 << s, half = (NV >> 1) >> s;
            const unsigned up = (lane & off) ? 0xFFFFFFFFu : 0u;        // bit selects (v_bfi): a `?:` on the register array becomes an
#pragma unroll                                                          // indexed extract = a chain of 8 compares + selects per value
            for (int i = 0; i < half; ++i) {
                const float send = bit_select(up, v[i], v[i + half]);
                const float keep = bit_select(up, v[i + half], v[i]);
                v[i] = keep + __shfl_xor(send, off, 64);
            }
        }
#pragma unroll
        for (int off = 1 << NFOLD; off < G; off <<= 1)
#pragma unroll
            for (int i = 0; i < CNT; ++i) v[i] += __shfl_xor(v[i], off, 64);
        if (rv && writer) {
            const int rs = d.res_shift;
#pragma unroll
            for (int i = 0; i < CNT; ++i) {
                const int idx = mine + i, xi = idx / CO, c = idx & (CO - 1);
                const int x = x0 + xi;
                if (x < W && c < d.cout) {
                    const size_t opix = ((size_t)n * d.Ho + y) * d.Wo + x;
                    float o = v[i] * sc[i] + sh[i];
                    if (d.res) o += d.res[(((size_t)n * (d.Ho >> rs) + (y >> rs)) * (d.Wo >> rs) + (x >> rs)) * d.res_ld + d.res_coff + c];
                    d.out[opix * d.out_ld + d.out_coff + c] = vps_act(o, d.act, d.slope);
                }
            }
        }
        run += ngroups; xr += dr; y += dq;
        if (xr >= runs_per_row) { xr -= runs_per_row; ++y; }
        while (y >= H) { y -= H; ++n; }
    }
}

}  // namespace

__attribute__((visibility("hidden")))
void vpsi_launch_conv_small(const vps_conv_desc& d, const int M, hipStream_t s) {
    int G = 1, logG = 0;
    while (G < 64 && G < (d.cin_pad >> 2)) { G <<= 1; ++logG; }
    const size_t wbytes = (size_t)(d.cout <= 2 ? 2 : 4) * d.kpad * sizeof(float);
    if (d.KH == 3 && d.KW == 3 && d.stride == 1 && d.nclass == 1 && d.pad_y[0] == 1 && d.pad_x[0] == 1 && d.Ho == d.H && d.Wo == d.W &&
        wbytes <= 150 * 1024 && (d.cout <= 2 || d.cout_pad >= 4) && (size_t)d.N * d.H * d.W * d.in_ld * sizeof(float) < 0xFFFFFFF0ull &&
        (long)d.N * d.H * d.W < 0x7fffffffL) {               // 32-bit run index
        constexpr int RUN = 4;
        const int run = d.cout <= 2 ? RUN : RUN / 2;
        const int runs_per_row = (d.W + run - 1) / run;
        const long total_runs = (long)d.N * d.H * runs_per_row;
        const int ppw = 64 >> logG;
        long blocks = (total_runs + 4 * ppw - 1) / (4 * ppw);
        if (blocks > 768) blocks = 768;               // 3 blocks of 4 waves per CU (152 VGPRs: all 18 loads of a run in flight): one resident round, the weights are staged once per block
#define VPS_S3V(LG)                                                                                                                            \
        do {                                                                                                                               \
            static bool attr_v = false;                                                                                                    \
            if (!attr_v) {                                                                                                                 \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small3x3v_kernel<2, RUN, LG>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);     \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small3x3v_kernel<4, RUN / 2, LG>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
                attr_v = true;                                                                                                             \
            }                                                                                                                              \
            if (d.cout <= 2) hipLaunchKernelGGL((conv_small3x3v_kernel<2, RUN, LG>), dim3((unsigned)blocks), dim3(256), wbytes, s, d, runs_per_row, (int)total_runs);      \
            else hipLaunchKernelGGL((conv_small3x3v_kernel<4, RUN / 2, LG>), dim3((unsigned)blocks), dim3(256), wbytes, s, d, runs_per_row, (int)total_runs);           \
        } while (0)
        switch (logG) {
            case 0: VPS_S3V(0); break; case 1: VPS_S3V(1); break; case 2: VPS_S3V(2); break; case 3: VPS_S3V(3); break;
            case 4: VPS_S3V(4); break; case 5: VPS_S3V(5); break; default: VPS_S3V(6); break;
        }
#undef VPS_S3V
        return;
    }
    const long total = (long)d.nclass * M;
    {   // eight loads per lane in flight, weights in LDS: conv_small_batched_kernel (VPS_SMALL_BATCHED=0: the one-load-per-step kernel, which also keeps
        // the layers with fewer than eight loads per pixel - the 2 -> 2 up-flow layers measured 31 us there against 36 us here)
        const char* e = getenv("VPS_SMALL_BATCHED");
        const size_t wb = (size_t)d.nclass * (d.cout <= 2 ? 2 : 4) * d.kpad * sizeof(float);
        if (!(e && e[0] == '0') && (d.cin_pad >> 2) * d.KH * d.KW >= 8 && wb <= 48 * 1024 && (d.cout <= 2 || d.cout_pad >= 4) && (size_t)d.N * d.H * d.W * d.in_ld * sizeof(float) < 0xFFFFFFF0ull) {
            const int c4n = d.cin_pad >> 2, ntap = d.KH * d.KW;
            int Gb = 1, logGb = 0;
            while (Gb < 64 && Gb * 8 < c4n * ntap && Gb < c4n) { Gb <<= 1; ++logGb; }
            const int nslot = (c4n + Gb - 1) >> logGb;
            const int ppw = 64 >> logGb;
            const long waves = (total + ppw - 1) / ppw;
            long blocks = (waves + 3) / 4; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;   // 8 blocks per CU: one resident round
            if (d.cout <= 2) hipLaunchKernelGGL((conv_small_batched_kernel<2>), dim3((unsigned)blocks), dim3(256), wb, s, d, M, Gb, logGb, nslot);
            else hipLaunchKernelGGL((conv_small_batched_kernel<4>), dim3((unsigned)blocks), dim3(256), wb, s, d, M, Gb, logGb, nslot);
            return;
        }
    }
    long waves = (total + (64 >> logG) - 1) / (64 >> logG);
    long blocks = (waves + 3) / 4; if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    if (d.cout <= 2) hipLaunchKernelGGL((conv_small_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, s, d, M, G, logG);
    else hipLaunchKernelGGL((conv_small_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, s, d, M, G, logG);
    return;
}
