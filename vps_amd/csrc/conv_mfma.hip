// Implicit-GEMM convolution / transposed convolution / linear / deformable convolution on the gfx950
// matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, 157 TFLOP/s dense peak).
//
// Replaces (reference, relative to /root/reference): every nn.Conv2d/ConvTranspose2d/Linear (+eval
// BatchNorm, +ReLU/LeakyReLU, +residual add) on PanopticFuseTrack.simple_test — SURVEY §8a rows a2, a7,
// a8, a11-a15, a17, a19, a20 — and mmdet/ops/dcn/src/deform_conv_cuda.cpp:151-250 +
// deform_conv_cuda_kernel.cu:83-113,189-241 (DCNv1 forward) WITHOUT the 1.2 GB column buffer: the bilinear
// sampling happens in the A-tile loader of the same GEMM.
//
// Design (MI355X-first, not a translation of the im2col+cuBLAS structure):
//   * NHWC activations, channel stride/offset per tensor -> outputs land directly in concat buffers.
//   * Block tile 128 (pixels) x {128,64,32} (cout) x 32 (k); 4 wavefronts (64 lanes) per block, each wave
//     owns a (TM*32)x(TN*32) sub-tile as TM*TN 32x32 MFMA accumulators (16 VGPR each).
//   * LDS tiles are [row][k] with a 36-float row stride: ds_write_b128 from coalesced float4 global loads,
//     and the MFMA fragments are fetched with ONE conflict-free ds_read_b128 per 4 MFMAs by permuting
//     the k order inside each group of 8 (lanes 0-31 take k0..3, lanes 32-63 take k4..7; MFMA j uses
//     element j of both fragments, so A and B agree on the permutation).
//   * global->register prefetch of tile t+1 is issued before the MFMAs of tile t (loads in flight under
//     4096 cycles of matrix work), written to LDS after.
//   * blockIdx -> tile mapping is XCD-aware (8 XCDs with private L2): consecutive tiles (which share
//     input halos / the same weight panel) stay on one XCD.
//   * split-K for the deep, low-resolution layers (FlowNet conv5/6, ResNet layer4) that would otherwise
//     launch < 256 blocks on a 256-CU chip; partials go to a caller workspace and a reduce kernel applies
//     the epilogue.
#include "conv_common.h"

int vpsi_launch_conv_q(const vps_conv_desc& d, int M, int tiles_m, int tiles_n, int per_split, long nblk, bool tapmajor, hipStream_t s);
int vpsi_launch_conv_thin(const vps_conv_desc& d, hipStream_t s);
void vpsi_launch_conv_small(const vps_conv_desc& d, int M, hipStream_t s);
void vpsi_launch_conv_n16(const vps_conv_desc& d, long tiles2d8, hipStream_t s);
void vpsi_launch_conv_h8(const vps_conv_desc& d, int tiles_m8, int tiles_n, int chunks_per_split, long nblk8, hipStream_t s);
void vpsi_launch_conv_h8s2(const vps_conv_desc& d, int tiles_m8, int tiles_n, int chunks_per_split, long nblk8, int bn, hipStream_t s);


namespace {
template <int TM, int TN, int WAVES_M, int WAVES_N, bool DEFORM>
__global__ __launch_bounds__(256, 2)
void conv_mfma_f32_kernel(const vps_conv_desc d, const int M, const int tiles_m, const int tiles_n,
                          const int ksteps_per_split) {
    constexpr int BN = WAVES_N * TN * 32;
    static_assert(WAVES_M * TM * 32 == BM, "block M tile must be 128");
    static_assert(WAVES_M * WAVES_N == 4, "4 wavefronts per block");
    constexpr int NB = BN / 32;  // float4 B loads per thread per k-step

    __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_LD];

    const int t = threadIdx.x;
    int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    int tile_n, tile_m, cls, split;
    decode_tile(d, swz, tiles_n, tiles_m, tile_n, tile_m, cls, split);

    const int py = cls / d.os_x, px = cls - py * d.os_x;
    const int pad_y = d.pad_y[py], pad_x = d.pad_x[px];
    const float* __restrict__ wcls = d.w + (size_t)cls * d.cout_pad * d.kpad;

    const int H = d.H, W = d.W, KH = d.KH, KW = d.KW, cin_pad = d.cin_pad;
    const int k4 = t & 7;     // which float4 of the 32-wide k slab this thread stages
    const int r0 = t >> 3;    // first tile row this thread stages (then +32, +64, +96)

    // ---- per-thread row bookkeeping for the A (im2col) loader
    RowInfo ri[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = tile_m * BM + r0 + 32 * i;
        if (m < M) {
            const int qx = m % d.Qw;
            const int tq = m / d.Qw;
            const int qy = tq % d.Qh;
            const int n = tq / d.Qh;
            ri[i].iy0 = qy * d.stride - pad_y;
            ri[i].ix0 = qx * d.stride - pad_x;
            ri[i].pixbase = n * H * W;
            ri[i].moff = m;
        } else {
            ri[i].iy0 = -(1 << 24);  // always out of bounds -> zero rows
            ri[i].ix0 = 0;
            ri[i].pixbase = 0;
            ri[i].moff = -1;
        }
    }

    // ---- k bookkeeping: this thread's float4 covers k = kk .. kk+3 (one tap, 4 consecutive channels)
    const int kstep0 = split * ksteps_per_split;
    int nsteps = d.kpad / BK - kstep0;
    if (nsteps > ksteps_per_split) nsteps = ksteps_per_split;
    // k ordering. korder 0 (tap-major): k = tap*cin_pad + ci. korder 1 (chunk-major): k = (chunk*ntap + tap)*32 + c with
    // ci = chunk*32 + c: all taps of one 32-channel slab are consecutive k-steps, so the 9 (25, 49) shifted reads of the same
    // 128-byte activation lines happen within a few steps of each other and hit L1/L2 instead of the Infinity Cache / HBM.
    const int korder = d.korder, ntap = KH * KW;
    int ky, kx, ci;
    if (korder == 0) {
        const int kk = kstep0 * BK + k4 * 4;
        const int tap = kk / cin_pad;
        ci = kk - tap * cin_pad;
        ky = tap / KW;
        kx = tap - ky * KW;
    } else {
        const int chunk = kstep0 / ntap, tap = kstep0 - chunk * ntap;
        ci = chunk * BK + k4 * 4;
        ky = tap / KW;
        kx = tap - ky * KW;
    }
    const float* __restrict__ wrow = wcls + (size_t)(tile_n * BN + r0) * d.kpad + (size_t)kstep0 * BK + k4 * 4;

    f32x4 areg[4];
    unsigned aok = 0;   // bit i: staged row i of the tile in flight is inside the image
    f32x4 breg[NB];
    // deformable: 4 corners + 4 weights per staged row, blended when written to LDS
    f32x4 dcv[DEFORM ? 4 : 1][4];
    float dcw[DEFORM ? 4 : 1][4];

    auto load_tiles = [&](int step) {
        const bool kval = ky < KH && ci < cin_pad;
        if constexpr (!DEFORM) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // branch-free: out-of-image taps read pixel 0 of the image (always mapped) and are zeroed when the tile is
                // written to LDS, so the load is issued unconditionally and its wait sits at the consumer, one tile later
                const int iy = ri[i].iy0 + ky, ix = ri[i].ix0 + kx;
                const bool ok = kval && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const int pix = ri[i].pixbase + (ok ? iy * W + ix : 0);
                areg[i] = *reinterpret_cast<const f32x4*>(d.in + ((size_t)pix * d.in_ld + d.in_coff + (ok ? ci : 0)));
                aok = (aok & ~(1u << i)) | ((ok ? 1u : 0u) << i);
            }
        } else {
            const int tap = ky * KW + kx;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // branch-free: corner weights are zeroed where the reference zeroes the corner value (or skips the whole
                // sample), corner addresses are clamped into the image, and all four float4 loads are always issued
                const bool act = kval && ri[i].moff >= 0;
                const float* op = d.offset + (size_t)(act ? ri[i].moff : 0) * d.off_ld + 2 * min(tap, KH * KW - 1);
                const float h_im = (float)(ri[i].iy0 + ky) + op[0];
                const float w_im = (float)(ri[i].ix0 + kx) + op[1];
                const bool inside = act && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
                const float hf = floorf(h_im), wf = floorf(w_im);
                const int h_low = inside ? (int)hf : 0, w_low = inside ? (int)wf : 0;
                const int h_high = h_low + 1, w_high = w_low + 1;
                const float lh = h_im - hf, lw = w_im - wf;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const bool hl = inside && h_low >= 0, hhv = inside && h_high <= H - 1;
                const bool wl = w_low >= 0, whv = w_high <= W - 1;
                dcw[i][0] = (hl && wl) ? hh * hw : 0.f;
                dcw[i][1] = (hl && whv) ? hh * lw : 0.f;
                dcw[i][2] = (hhv && wl) ? lh * hw : 0.f;
                dcw[i][3] = (hhv && whv) ? lh * lw : 0.f;
                const int hlc = min(max(h_low, 0), H - 1), hhc = min(max(h_high, 0), H - 1);
                const int wlc = min(max(w_low, 0), W - 1), whc = min(max(w_high, 0), W - 1);
                const float* base = d.in + (size_t)ri[i].pixbase * d.in_ld + d.in_coff + ci;
                dcv[i][0] = *reinterpret_cast<const f32x4*>(base + (size_t)(hlc * W + wlc) * d.in_ld);
                dcv[i][1] = *reinterpret_cast<const f32x4*>(base + (size_t)(hlc * W + whc) * d.in_ld);
                dcv[i][2] = *reinterpret_cast<const f32x4*>(base + (size_t)(hhc * W + wlc) * d.in_ld);
                dcv[i][3] = *reinterpret_cast<const f32x4*>(base + (size_t)(hhc * W + whc) * d.in_ld);
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
            breg[j] = *reinterpret_cast<const f32x4*>(wrow + (size_t)(32 * j) * d.kpad + (size_t)step * BK);
        // advance this thread's (ky,kx,ci) by one k-slab
        if (korder == 0) {
            ci += BK;
            while (ci >= cin_pad) {
                ci -= cin_pad;
                if (++kx == KW) { kx = 0; ++ky; }
            }
        } else if (++kx == KW) {
            kx = 0;
            if (++ky == KH) { ky = 0; ci += BK; }
        }
    };

    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v;
            if constexpr (DEFORM) {
                v = dcw[i][0] * dcv[i][0] + dcw[i][1] * dcv[i][1] + dcw[i][2] * dcv[i][2] + dcw[i][3] * dcv[i][3];
            } else {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                v = ((aok >> i) & 1u) ? areg[i] : z;
            }
            *reinterpret_cast<f32x4*>(&As[(r0 + 32 * i) * LDS_LD + k4 * 4]) = v;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
            *reinterpret_cast<f32x4*>(&Bs[(r0 + 32 * j) * LDS_LD + k4 * 4]) = breg[j];
    };

    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int frag_off = (lane & 31) * LDS_LD + (lane >> 5) * 4;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (nsteps > 0) {
        load_tiles(0);
        store_tiles();
    }
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const bool more = step + 1 < nsteps;
        if (more) load_tiles(step + 1);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a)
                af[a] = *reinterpret_cast<const f32x4*>(&As[(wm * TM * 32 + a * 32) * LDS_LD + frag_off + kg * 8]);
#pragma unroll
            for (int b = 0; b < TN; ++b)
                bf[b] = *reinterpret_cast<const f32x4*>(&Bs[(wn * TN * 32 + b * 32) * LDS_LD + frag_off + kg * 8]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[b][j], af[a][j], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_tiles();
            __syncthreads();
        }
    }

    conv_epilogue<TM, TN, BN>(d, acc, M, tile_m, tile_n, cls, split, py, px, wm, wn, lane, (cls * tiles_m + tile_m) * tiles_n + tile_n);
}

// ================================================================================================
// Split-bf16, software-pipelined ("bf16p"): the kernel every non-deformable layer runs in the bf16x3 / bf16x6 modes.
//
// Why it looks the way it does (measurements: tools/pipebench.hip, profiles/r01_pipebench.txt, and phase knock-outs /
// in-kernel phase timers of the kernel above on the 256->256 3x3 @256x512 layer):
//   * on one SIMD, a wave issuing MFMAs and a second wave doing anything else (VALU, ds_read, global loads) take the SUM
//     of their times, not the max: the two-phase "compute | barrier | stage | barrier" loop above therefore runs at
//     matrix time + everything else, whatever the occupancy (measured 1.08 ms against 0.42 ms of pure matrix time);
//   * ds_read_b128 delivers ~64 B/clk/CU: the 24 fragment reads per wave and k-step of the kernel above keep the LDS
//     pipe as busy as the matrix pipe; coalesced 1 KB global loads deliver ~47 B/clk/CU.
// So: (1) the weight operand skips LDS. It is split and packed once on the host in MFMA-fragment order,
//       w_split[plane][class][cout_pad/32][kpad/16][lane 0..63][8 bf16],  lane = 32*(k/8 % 2) + cout % 32,
//     and each wave fetches its B fragments with one coalesced 1 KB global_load_dwordx4 per fragment (L2-resident), one
//     k-step ahead, double-buffered in registers: half the LDS reads, no weight staging writes.
// (2) LDS holds the split activations only, double-buffered, ONE barrier per k-step.
// (3) everything that is not an MFMA - converting and staging the activations of step s+1, the fragment reads of the
//     second slab, issuing the activation loads of step s+2 and the weight loads of step s+1 - is cut into small work
//     items that are interleaved between the MFMAs of step s in program order (sched_barrier-pinned), so that each
//     wave keeps the matrix pipe, the LDS pipe and the memory pipe busy at the same time instead of in turns.
// The loop body is branch-free: loads past the last k-step are clamped / masked rather than skipped.
// ================================================================================================
// TAPMAJOR: k = tap * cin_pad + ci (small channel counts: the first layers) - the (tap, channel) of a staged group differs per
// thread. Otherwise one k-step = one (32-channel chunk, tap), the same for all threads (chunk-major order; a 1x1 layer is its
// one-tap case), and the step's part of the byte offset is scalar.
// DEFORM (round 3): the deformable layers on the same pipeline. The four bilinear corners of a staged (row, tap) group are four
// buffer loads issued one k-step ahead; they are blended with the corner weights of the per-block coefficient table when the row is
// staged (between the MFMAs, like every other work item). Versus the two-barrier kernel of rounds 1-2 (load | MFMAs | barrier | blend +
// stage | barrier, weights through LDS; deleted in round 6 - no layer of the path used it): one barrier per k-step, staging interleaved
// with the matrix work, weights straight to registers.
template <int TM, int TN, int WAVES_M, int WAVES_N, int MODE, bool TAPMAJOR, bool DEFORM = false>
__global__ __launch_bounds__(256, (TN >= 4 ? 1 : 2))            // the 256-column deformable instance: one block per CU, 512 registers per lane
void conv_mfma_bf16p_kernel(const vps_conv_desc d, const int M, const int tiles_m, const int tiles_n,
                            const int ksteps_per_split) {
    constexpr int BN = WAVES_N * TN * 32;
    static_assert(WAVES_M * TM * 32 == BM, "block M tile must be 128");
    static_assert(WAVES_M * WAVES_N == 4, "4 wavefronts per block");
    typedef Split<MODE> SM;
    typedef typename SM::elem elem_t;
    typedef vec8<elem_t> x8;
    typedef vec4<elem_t> x4;
    constexpr int NSA = SM::NSA, NSB = SM::NSB;
    constexpr int ABUF = NSA * BM * LDS_LDH;     // 16-bit elements of one activation buffer (all planes)

    __shared__ __attribute__((aligned(16))) elem_t As[2 * ABUF];
    // DEFORM: corner weights (4 floats) and clamped corner coordinates (4 x u16) of every (tile row, tap), computed once per block
    constexpr int DTAP = 9;
    __shared__ __attribute__((aligned(16))) float cw[DEFORM ? BM * DTAP * 4 : 4];
    __shared__ __attribute__((aligned(8))) unsigned short cc[DEFORM ? BM * DTAP * 4 : 4];
    static_assert(!(DEFORM && TAPMAJOR), "deformable layers use the chunk-major order");

    const int t = threadIdx.x;
    int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    int tile_n, tile_m, cls, split;
    decode_tile(d, swz, tiles_n, tiles_m, tile_n, tile_m, cls, split);

    const int py = cls / d.os_x, px = cls - py * d.os_x;
    const int pad_y = d.pad_y[py], pad_x = d.pad_x[px];
    const int H = d.H, W = d.W, KH = d.KH, KW = d.KW, cin_pad = d.cin_pad;
    const int k4 = t & 7;      // 4-channel group of the 32-wide k-step staged by this thread (8 lanes = one 128-byte line)
    const int r0 = t >> 3;     // rows r0 + 32 i of the tile

    RowInfo ri[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = tile_m * BM + r0 + 32 * i;
        if (m < M) {
            const int qx = m % d.Qw;
            const int tq = m / d.Qw;
            const int qy = tq % d.Qh;
            const int n = tq / d.Qh;
            ri[i].iy0 = qy * d.stride - pad_y;
            ri[i].ix0 = qx * d.stride - pad_x;
            ri[i].pixbase = n * H * W;
        } else {
            ri[i].iy0 = -(1 << 24);
            ri[i].ix0 = 0;
            ri[i].pixbase = 0;
        }
        ri[i].moff = m;
    }

    const int kstep0 = split * ksteps_per_split;
    int nsteps = d.kpad / BK - kstep0;
    if (nsteps > ksteps_per_split) nsteps = ksteps_per_split;
    // k ordering as in the kernels above. korder 1: one k-step = one (32-channel chunk, tap), the same for all threads.
    const int korder = d.korder, ntap = KH * KW;
    int ky = 0, kx = 0, chunk = 0, astep = kstep0;   // state of the next activation tile to load
    if constexpr (!TAPMAJOR) {
        chunk = kstep0 / ntap;
        const int tap = kstep0 - chunk * ntap;
        ky = tap / KW;
        kx = tap - ky * KW;
    }
    (void)korder;

    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    // B fragments of this wave: 32-column blocks nb0 .. nb0+TN-1, k-slab ks, plane p
    const int nbt = d.cout_pad >> 5, kst = d.kpad >> 4;
    // weights: one buffer resource over the whole split-weight tensor; this wave's fragment base is a SCALAR byte offset, the lane
    // adds its 16 bytes -> `buffer_load_dwordx4 v, v_lane, s[rsrc], s_off offen`: the per-load address arithmetic is scalar
    const size_t wplane = (size_t)d.nclass * nbt * kst * 512;
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(d.w_split, (unsigned)(wplane * NSB * sizeof(elem_t)));
    const unsigned wbase = (unsigned)((((size_t)(cls * nbt + tile_n * (BN / 32) + wn * TN) * kst + 2 * (size_t)kstep0) * 512) * sizeof(elem_t));
    const unsigned wlane = (unsigned)lane * 16u;
    // activations: one resource over the input tensor; a staged row contributes pix * in_ld * 4 (+ channel bytes) as a 32-bit offset
    const unsigned in_bytes = (unsigned)((size_t)d.N * H * W * d.in_ld * sizeof(float));
    const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(d.in, in_bytes);
    const unsigned ld4 = (unsigned)d.in_ld * 4u;
    const unsigned acoff = (unsigned)(d.in_coff + k4 * 4) * 4u;      // this thread's 4-channel group inside a 32-channel chunk
    unsigned rowoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rowoff[i] = (unsigned)(ri[i].pixbase + ri[i].iy0 * W + ri[i].ix0) * ld4 + acoff;   // wraps for rows outside: masked

    // activation tiles in flight in registers: tile T waits in slot T & 1 from the step T-3 that requested it until step T-1
    // stages it (two full k-steps: a 3-product k-step is 24 MFMAs = 0.4 us, shorter than a trip to HBM)
    f32x4 areg[DEFORM ? 1 : 2][4];
    f32x4 dcv[DEFORM ? 4 : 1][4];          // DEFORM: the four corner values of the staged rows of the tile in flight (ONE tile ahead)
    float dcw[DEFORM ? 4 : 1][4];
    x8 bnext[2][NSB][TN];
    float amax = 0.f;

    // next activation tile -> registers of `slot` (sequential: every call advances the k state by one step)
    auto load_A = [&](const int slot) {
        int kyc, kxc, chunk_of_step = 0;
        unsigned stepoff;      // byte offset = rowoff[i] (k-invariant, per thread) + the step's (tap, channel) part
        bool kv;
        if constexpr (!TAPMAJOR) {
            // scalar state: the step's offset is an SGPR, one vector add per staged row
            kyc = ky; kxc = kx;
            kv = chunk * BK + k4 * 4 < cin_pad;
            chunk_of_step = chunk;
            stepoff = (unsigned)(kyc * W + kxc) * ld4 + (unsigned)chunk * (BK * 4u);
            if (++kx == KW) {
                kx = 0;
                if (++ky == KH) { ky = 0; ++chunk; }
            }
        } else {
            const int kk = astep * BK + k4 * 4;
            const int tap = kk / cin_pad;
            const int cic = kk - tap * cin_pad;
            kyc = tap / KW;
            kxc = tap - kyc * KW;
            kv = tap < ntap;
            stepoff = (unsigned)(kyc * W + kxc) * ld4 + (unsigned)(cic - k4 * 4) * 4u;
        }
        ++astep;
        if constexpr (DEFORM) {
            // table entry of (row, tap): four corner loads at clamped coordinates; the weights are zero where the reference zeroes the
            // corner or skips the sample; beyond the channel range the offset lies outside the buffer (zeros)
            const int tap = min(kyc * KW + kxc, DTAP - 1);
            const unsigned choff = acoff + (unsigned)chunk_of_step * (BK * 4u);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = ((r0 + 32 * i) * DTAP + tap) * 4;
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(&cw[e]);
                const vec4<unsigned short> c4 = *reinterpret_cast<const vec4<unsigned short>*>(&cc[e]);
                dcw[i][0] = w4[0]; dcw[i][1] = w4[1]; dcw[i][2] = w4[2]; dcw[i][3] = w4[3];
                const unsigned pb = (unsigned)ri[i].pixbase;
                const unsigned o00 = (pb + (unsigned)c4[0] * W + c4[2]) * ld4 + choff, o01 = (pb + (unsigned)c4[0] * W + c4[3]) * ld4 + choff;
                const unsigned o10 = (pb + (unsigned)c4[1] * W + c4[2]) * ld4 + choff, o11 = (pb + (unsigned)c4[1] * W + c4[3]) * ld4 + choff;
                dcv[i][0] = buffer_load16<f32x4>(arsrc, kv ? o00 : 0xFFFFFFF0u, 0u);
                dcv[i][1] = buffer_load16<f32x4>(arsrc, kv ? o01 : 0xFFFFFFF0u, 0u);
                dcv[i][2] = buffer_load16<f32x4>(arsrc, kv ? o10 : 0xFFFFFFF0u, 0u);
                dcv[i][3] = buffer_load16<f32x4>(arsrc, kv ? o11 : 0xFFFFFFF0u, 0u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // out-of-image / out-of-range taps: offset beyond the buffer -> the load returns zeros (no branch, no select on the data)
                const int iy = ri[i].iy0 + kyc, ix = ri[i].ix0 + kxc;
                const bool ok = kv && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                areg[slot][i] = buffer_load16<f32x4>(arsrc, ok ? rowoff[i] + stepoff : 0xFFFFFFF0u, 0u);
            }
        }
    };

    // weight fragments (plane p, slab m) of k-step `step` -> bnext
    auto load_B = [&](int step, int m, int p) {
#pragma unroll
        for (int b = 0; b < TN; ++b)
            bnext[m][p][b] = buffer_load16<x8>(wrsrc, wlane, wbase + (unsigned)(((size_t)p * wplane + ((size_t)b * kst + 2 * step + m) * 512) * sizeof(elem_t)));
    };

    // split staged row i of `slot` and write it into activation buffer `buf`
    auto store_A = [&](int i, int buf, const int slot) {
        x4 sp[NSA];
        if constexpr (DEFORM) split_act<MODE>(dcw[i][0] * dcv[i][0] + dcw[i][1] * dcv[i][1] + dcw[i][2] * dcv[i][2] + dcw[i][3] * dcv[i][3], sp, amax);
        else split_act<MODE>(areg[slot][i], sp, amax);
        const int row = r0 + 32 * i;
#pragma unroll
        for (int p = 0; p < NSA; ++p)
            *reinterpret_cast<x4*>(&As[buf * ABUF + p * (BM * LDS_LDH) + row * LDS_LDH + (((k4 >> 1) ^ lds_swz(row)) << 3) + ((k4 & 1) << 2)]) = sp[p];
    };

    // fragment of slab m: logical 16-byte chunk 2m + (lane>>5) of row (lane&31), swizzled like the writes
    const int frag_row = (wm * TM * 32 + (lane & 31)) * LDS_LDH;
    const int frag_sw = lds_swz(lane & 31);
    const int frag_chunk[2] = {(((lane >> 5)) ^ frag_sw) << 3, ((2 + (lane >> 5)) ^ frag_sw) << 3};
    x8 af[2][NSA][TM];
    auto read_A = [&](int m, int buf) {
#pragma unroll
        for (int p = 0; p < NSA; ++p)
#pragma unroll
            for (int a = 0; a < TM; ++a)
                af[m][p][a] = *reinterpret_cast<const x8*>(&As[buf * ABUF + p * (BM * LDS_LDH) + a * 32 * LDS_LDH + frag_row + frag_chunk[m]]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // prologue: tile 0 staged in buffer 0, tiles 1 and 2 in flight in registers (slots 1, 0), weights of step 0 in flight
    if constexpr (DEFORM) {
        // deform_conv_cuda_kernel.cu:83-113,205-237: sample position = tap position + (dh, dw) of this output pixel; valid iff
        // -1 < h < H and -1 < w < W; a corner outside the image contributes 0
        for (int e = t; e < BM * DTAP; e += 256) {
            const int row = e / DTAP, tap = e - row * DTAP;
            const int m = tile_m * BM + row;
            const bool act = m < M && tap < ntap;
            const int mm = act ? m : 0;
            const int qx = mm % d.Qw, tq = mm / d.Qw, qy = tq % d.Qh;
            const int tky = tap / KW, tkx = tap - tky * KW;
            const float* op = d.offset + (size_t)mm * d.off_ld + 2 * min(tap, ntap - 1);
            const float h_im = (float)(qy * d.stride - pad_y + tky) + op[0];
            const float w_im = (float)(qx * d.stride - pad_x + tkx) + op[1];
            const bool inside = act && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_low = inside ? (int)hf : 0, w_low = inside ? (int)wf : 0;
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - hf, lw = w_im - wf;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const bool hl = inside && h_low >= 0, hhv = inside && h_high <= H - 1;
            const bool wl = w_low >= 0, whv = w_high <= W - 1;
            const f32x4 w4 = {(hl && wl) ? hh * hw : 0.f, (hl && whv) ? hh * lw : 0.f, (hhv && wl) ? lh * hw : 0.f, (hhv && whv) ? lh * lw : 0.f};
            *reinterpret_cast<f32x4*>(&cw[e * 4]) = w4;
            vec4<unsigned short> c4;
            c4[0] = (unsigned short)min(max(h_low, 0), H - 1); c4[1] = (unsigned short)min(max(h_high, 0), H - 1);
            c4[2] = (unsigned short)min(max(w_low, 0), W - 1); c4[3] = (unsigned short)min(max(w_high, 0), W - 1);
            *reinterpret_cast<vec4<unsigned short>*>(&cc[e * 4]) = c4;
        }
        __syncthreads();
    }
    if (nsteps > 0) {
        load_A(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int p = 0; p < SM::NLB; ++p) load_B(0, m, p);
#pragma unroll
        for (int i = 0; i < 4; ++i) store_A(i, 0, 0);
        if constexpr (DEFORM) {
            load_A(0);                 // ONE tile in flight: tile 1
        } else {
            load_A(1);
            load_A(0);
        }
    }
    __syncthreads();

    // work items interleaved between the MFMAs of a k-step (program order; positions are compile-time after unrolling)
    constexpr int NT = SM::NT;
    constexpr int NMF = 2 * NT * TM * TN;            // MFMAs per wave and k-step
    constexpr int NLB = SM::NLB;                     // weight planes loaded; the others are derived in registers
    constexpr int NW = 4 + 1 + 1 + 2 * NLB;          // 4 row stagings, slab-1 fragment reads, next A loads, 2*NLB weight loads

    // one k-step on activation buffer CUR (= parity of the step, compile-time: the loop below is unrolled by two so that the
    // register slots are static)
    auto kstep = [&](const int step, auto cur_tag) {
        constexpr int cur = decltype(cur_tag)::value;
        const int bstep = min(step + 1, nsteps - 1);   // weights of the next step (clamped: the last prefetch is unused)
        x8 bcur[2][NSB][TN];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int p = 0; p < NLB; ++p)
#pragma unroll
                for (int b = 0; b < TN; ++b) bcur[m][p][b] = bnext[m][p][b];
            if constexpr (NLB < NSB) {
#pragma unroll
                for (int b = 0; b < TN; ++b) bcur[m][2][b] = derive_weight_plane<MODE>(bcur[m][0][b]);
            }
        }
        read_A(0, cur);
        __builtin_amdgcn_sched_barrier(0);

        auto work = [&](const int w) {
            constexpr int slot = DEFORM ? 0 : (cur ^ 1);
            if (w < 2) store_A(w, cur ^ 1, slot);                          // stage tile step+1 (its loads are one step old)
            else if (w == 2) read_A(1, cur);                      // fragments of the second slab
            else if (w < 5) store_A(w - 1, cur ^ 1, slot);
            else if (w == 5) load_A(slot);                        // tile step+3 (DEFORM: step+2) -> the slot the stagings above just emptied
            else load_B(bstep, (w - 6) / NLB, (w - 6) % NLB);       // weights of step+1 -> registers
        };

        int mf = 0;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        acc[a][b] = split_mfma<MODE>(bcur[m][SM::PB[q]][b], af[m][SM::PA[q]][a], acc[a][b]);
                        ++mf;
                        // item w runs after MFMA number max(1, (w+1)*NMF/(NW+1)): evenly spread, and the slab-1 fragment
                        // reads (item 2) always land in the first quarter of the stream, well before their consumers
#pragma unroll
                        for (int w = 0; w < NW; ++w) {
                            const int pos = ((w + 1) * NMF) / (NW + 1);
                            if (mf == (pos < 1 ? 1 : pos)) {
                                __builtin_amdgcn_sched_barrier(0);
                                work(w);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
        __syncthreads();
    };
    for (int step = 0; step < nsteps; step += 2) {
        kstep(step, std::integral_constant<int, 0>{});
        if (step + 1 < nsteps) kstep(step + 1, std::integral_constant<int, 1>{});
    }
    report_range<MODE>(d, amax);
    conv_epilogue<TM, TN, BN, false, 4, DEFORM>(d, acc, M, tile_m, tile_n, cls, split, py, px, wm, wn, lane, (cls * tiles_m + tile_m) * tiles_n + tile_n);
}

// ================================================================================================
// Split-bf16, halo-staged ("bf16h"): stride-1 KHxKW layers (3x3 convolutions, the 2x2 parity classes of the transposed
// convolutions) whose output is a whole number of 8x16 patches. Same pipeline as bf16p, different activation path:
// the block's 128 rows are an 8x16 PATCH of output positions, and per 32-channel chunk its (8+KH-1)x(16+KW-1) input
// halo is loaded, split and staged in LDS ONCE; the KH*KW taps of the chunk then read their A fragments from that halo
// tile at a per-tap row offset. Versus bf16p (which re-loads and re-splits the 128x32 activation tile for every tap):
// 6.4x (3x3) / 3.3x (2x2) fewer activation loads, conversions and LDS writes, one barrier per chunk instead of per
// k-step. On this chip a saturated MFMA stream does not overlap with the other instructions of its SIMD (profiles/
// r01_pipebench.txt), so every removed instruction is time. The tap loop is fully unrolled (KH, KW are template
// parameters), so which work item goes between which MFMAs is decided at compile time.
// ================================================================================================
template <int TM, int TN, int WAVES_M, int WAVES_N, int MODE, int KH, int KW>
__global__ __launch_bounds__(256, 2)
void conv_mfma_bf16h_kernel(const vps_conv_desc d, const int tiles_m, const int tiles_n, const int chunks_per_split) {
    constexpr int BN = WAVES_N * TN * 32;
    static_assert(WAVES_M * TM * 32 == BM, "block M tile must be 128");
    static_assert(WAVES_M * WAVES_N == 4, "4 wavefronts per block");
    constexpr int NTAP = KH * KW;
    constexpr int HW = 16 + KW - 1, HH = 8 + KH - 1;   // halo tile
    constexpr int HROWS = HH * HW;                      // <= 180
    constexpr int NLD = (HROWS + 31) / 32;              // staged rows per thread (32 rows per pass of the 256 threads)
    typedef Split<MODE> SM;
    typedef typename SM::elem elem_t;
    typedef vec8<elem_t> x8;
    typedef vec4<elem_t> x4;
    constexpr int NSA = SM::NSA, NSB = SM::NSB;
    constexpr int PLANE = NLD * 32 * LDS_LDH;           // 16-bit elements of one plane of one buffer
    constexpr int ABUF = NSA * PLANE;

    __shared__ __attribute__((aligned(16))) elem_t As[2 * ABUF];

    const int t = threadIdx.x;
    int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    int tile_n, tile_m, cls, split;             // split: split-K over whole 32-channel chunks
    decode_tile(d, swz, tiles_n, tiles_m, tile_n, tile_m, cls, split);

    const int py = cls / d.os_x, px = cls - py * d.os_x;
    const int H = d.H, W = d.W, cin_pad = d.cin_pad;
    const int tiles_x = (d.Qw + 15) >> 4, tiles_y = (d.Qh + 7) >> 3;
    const int tx = tile_m % tiles_x, tq = tile_m / tiles_x;
    const int ty = tq % tiles_y, n = tq / tiles_y;
    const int iy_org = ty * 8 - d.pad_y[py], ix_org = tx * 16 - d.pad_x[px];   // input position of halo row 0, column 0

    const int k4 = t & 7;      // 4-channel group staged by this thread (8 lanes = one 128-byte line)
    const int r0 = t >> 3;     // halo rows r0 + 32 i
    const int chunk0 = split * chunks_per_split;
    const int nchunks = min(chunks_per_split, d.kpad / (BK * NTAP) - chunk0);
    const int nsteps = nchunks * NTAP;

    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int nbt = d.cout_pad >> 5, kst = d.kpad >> 4;
    const size_t wplane = (size_t)d.nclass * nbt * kst * 512;
    // buffer-addressed loads as in the pipelined kernel: weights at a scalar byte offset + the lane's 16 bytes, activations at a
    // 32-bit byte offset that lies beyond the buffer for halo positions outside the image / the channel range (-> zeros)
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(d.w_split, (unsigned)(wplane * NSB * sizeof(elem_t)));
    const unsigned wbase = (unsigned)((((size_t)(cls * nbt + tile_n * (BN / 32) + wn * TN) * kst + 2 * (size_t)chunk0 * NTAP) * 512) * sizeof(elem_t));
    const unsigned wlane = (unsigned)lane * 16u;
    const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(d.in, (unsigned)((size_t)d.N * H * W * d.in_ld * sizeof(float)));
    const unsigned ld4 = (unsigned)d.in_ld * 4u;

    f32x4 areg[NLD];
    int achunk = chunk0;       // next chunk to load
    x8 bnext[2][NSB][TN];
    float amax = 0.f;

    // byte offset of halo position r0 + 32 i (k-invariant), 0xFFFFFFF0 when it lies outside the image
    unsigned hoff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int hp = r0 + 32 * i;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int iy = iy_org + hy, ix = ix_org + hx;
        const bool ok = hp < HROWS && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        hoff[i] = ok ? (unsigned)((n * H + iy) * W + ix) * ld4 + (unsigned)(d.in_coff + k4 * 4) * 4u : 0xFFFFFFF0u;
    }

    auto load_A = [&]() {
        const bool kv = achunk * BK + k4 * 4 < cin_pad;
        const unsigned coff = (unsigned)achunk * (BK * 4u);       // scalar
        ++achunk;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            areg[i] = buffer_load16<f32x4>(arsrc, (kv && hoff[i] != 0xFFFFFFF0u) ? hoff[i] + coff : 0xFFFFFFF0u, 0u);
    };
    auto load_B = [&](int step, int m, int p) {
#pragma unroll
        for (int b = 0; b < TN; ++b)
            bnext[m][p][b] = buffer_load16<x8>(wrsrc, wlane, wbase + (unsigned)(((size_t)p * wplane + ((size_t)b * kst + 2 * step + m) * 512) * sizeof(elem_t)));
    };
    auto store_A = [&](int i, int buf) {
        x4 sp[NSA];
        split_act<MODE>(areg[i], sp, amax);
        const int row = r0 + 32 * i;
#pragma unroll
        for (int p = 0; p < NSA; ++p)
            *reinterpret_cast<x4*>(&As[buf * ABUF + p * PLANE + row * LDS_LDH + (((k4 >> 1) ^ lds_swz(row)) << 3) + ((k4 & 1) << 2)]) = sp[p];
    };

    // halo row of tile row j = wm*TM*32 + a*32 + (lane&31) for tap (0,0); tap (ky,kx) adds ky*HW + kx
    int hbase[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int j = wm * TM * 32 + a * 32 + (lane & 31);
        hbase[a] = (j >> 4) * HW + (j & 15);
    }
    x8 af[2][NSA][TM];
    auto read_A = [&](int m, int buf, int toff) {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int hrow = hbase[a] + toff;
            const int off = buf * ABUF + hrow * LDS_LDH + (((2 * m + (lane >> 5)) ^ lds_swz(hrow)) << 3);
#pragma unroll
            for (int p = 0; p < NSA; ++p) af[m][p][a] = *reinterpret_cast<const x8*>(&As[off + p * PLANE]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // prologue: chunk 0 staged in buffer 0, chunk 1 in flight in registers, weights of step 0 in flight
    load_A();
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int p = 0; p < NSB; ++p) load_B(0, m, p);
#pragma unroll
    for (int i = 0; i < NLD; ++i) store_A(i, 0);
    load_A();
    __syncthreads();

    constexpr int NT = SM::NT;
    constexpr int NMF = 2 * NT * TM * TN;                       // MFMAs per wave and tap
    constexpr int SPT = (NLD + NTAP - 2) / (NTAP - 1);          // halo rows staged per tap (the last tap issues the loads)

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cur = chunk & 1;
#pragma unroll
        for (int tp = 0; tp < NTAP; ++tp) {
            const int step = chunk * NTAP + tp;
            const int bstep = min(step + 1, nsteps - 1);        // weights of the next step (clamped: the last prefetch is unused)
            const int toff = (tp / KW) * HW + (tp % KW);
            constexpr int NLB = NSB;                            // all packed planes loaded (deriving the third one measured slower here: 512->512 3x3 @64x128 0.475 -> 0.545 ms)
            x8 bcur[2][NSB][TN];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int p = 0; p < NLB; ++p)
#pragma unroll
                    for (int b = 0; b < TN; ++b) bcur[m][p][b] = bnext[m][p][b];
                if constexpr (NLB < NSB) {
#pragma unroll
                    for (int b = 0; b < TN; ++b) bcur[m][2][b] = derive_weight_plane<MODE>(bcur[m][0][b]);
                }
            }
            read_A(0, cur, toff);
            __builtin_amdgcn_sched_barrier(0);

            // work items of this tap: SPT halo-row stagings (or, on the last tap, the loads of chunk+2), the fragment reads
            // of the second slab, 2*NSB weight loads of the next step
            const int NW = SPT + 1 + 2 * NLB;
            auto work = [&](const int w) {
                if (w == 1) read_A(1, cur, toff);
                else if (w == 0 || (w >= 2 && w < SPT + 1)) {
                    const int si = (w == 0 ? 0 : w - 1);             // 0 .. SPT-1
                    if (tp < NTAP - 1) {
                        const int row = tp * SPT + si;
                        if (row < NLD) store_A(row, cur ^ 1);
                    } else if (si == 0) load_A();
                } else load_B(bstep, (w - SPT - 1) / NLB, (w - SPT - 1) % NLB);
            };

            int mf = 0;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < NT; ++q)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b) {
                            acc[a][b] = split_mfma<MODE>(bcur[m][SM::PB[q]][b], af[m][SM::PA[q]][a], acc[a][b]);
                            ++mf;
#pragma unroll
                            for (int w = 0; w < NW; ++w) {
                                const int pos = ((w + 1) * NMF) / (NW + 1);
                                if (mf == (pos < 1 ? 1 : pos)) {
                                    __builtin_amdgcn_sched_barrier(0);
                                    work(w);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                        }
        }
        __syncthreads();
    }
    report_range<MODE>(d, amax);
    conv_epilogue<TM, TN, BN, true>(d, acc, tiles_m * BM, tile_m, tile_n, cls, split, py, px, wm, wn, lane, (cls * tiles_m + tile_m) * tiles_n + tile_n);
}

// sum the split-K partials (in split order, like one long accumulation) and apply the epilogue. V channels per thread
// (float4 when cout, the leading dimensions and the channel offsets are multiples of 4); the partials of up to 8 splits are
// requested before the first is added: a load inside the `for (split)` loop waited for itself, ksplit serial latencies.
template <int V>
__global__ __launch_bounds__(256)
void conv_splitk_reduce_kernel(const vps_conv_desc d, const int M) {
    const int cv = d.cout / V;
    const size_t total = (size_t)d.nclass * M * cv;
    const size_t plane = (size_t)d.nclass * M * d.cout_pad;     // floats per split
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(idx % cv) * V;
        const size_t t2 = idx / cv;
        const int m = (int)(t2 % M);
        const int cls = (int)(t2 / M);
        const float* __restrict__ wp = d.ws + ((size_t)cls * M + m) * d.cout_pad + co;
        float s[V];
#pragma unroll
        for (int e = 0; e < V; ++e) s[e] = 0.f;
        for (int sp0 = 0; sp0 < d.ksplit; sp0 += 8) {
            float v[8][V];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float* q = wp + (size_t)min(sp0 + j, d.ksplit - 1) * plane;
                if constexpr (V == 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(q);
                    v[j][0] = t[0]; v[j][1] = t[1]; v[j][2] = t[2]; v[j][3] = t[3];
                } else {
                    v[j][0] = q[0];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < V; ++e) s[e] = (sp0 + j < d.ksplit) ? s[e] + v[j][e] : s[e];
        }
        const int py = cls / d.os_x, px = cls - py * d.os_x;
        const int qx = m % d.Qw;
        const int tq = m / d.Qw;
        const int qy = tq % d.Qh;
        const int n = tq / d.Qh;
        const int oy = qy * d.os_y + py, ox = qx * d.os_x + px;
        const size_t opix = ((size_t)n * d.Ho + oy) * d.Wo + ox;
        const int rs = d.res_shift;
        const size_t rpix = ((size_t)n * (d.Ho >> rs) + (oy >> rs)) * (d.Wo >> rs) + (ox >> rs);
        float r[V];
#pragma unroll
        for (int e = 0; e < V; ++e) r[e] = 0.f;
        if (d.res) {
            if constexpr (V == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(d.res + rpix * d.res_ld + d.res_coff + co);
                r[0] = t[0]; r[1] = t[1]; r[2] = t[2]; r[3] = t[3];
            } else {
                r[0] = d.res[rpix * d.res_ld + d.res_coff + co];
            }
        }
        float o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float v = s[e] * (d.scale ? d.scale[co + e] : 1.f) + (d.shift ? d.shift[co + e] : 0.f);
            if (d.res) v += r[e];
            o[e] = vps_act(v, d.act, d.slope);
        }
        float* op = d.out + opix * d.out_ld + d.out_coff + co;
        if constexpr (V == 4) *reinterpret_cast<f32x4*>(op) = f32x4{o[0], o[1], o[2], o[3]};
        else op[0] = o[0];
    }
}

template <int TM, int TN, int WAVES_M, int WAVES_N>
int launch_conv(const vps_conv_desc& d, int M, hipStream_t s) {
    constexpr int BN = WAVES_N * TN * 32;
    const int tiles_m = cdiv(M, BM);
    const int tiles_n = d.cout_pad / BN;
    const int ksteps = d.kpad / BK;
    // split-K: `ksplit` ranges of ceil(steps / ksplit) k-steps, the last one shorter when the division leaves a rest (every kernel clamps
    // its range). Chunk-major layers (k = (32-channel chunk, tap)) are split over WHOLE chunks when ceil-division of the chunk count
    // reproduces ksplit - the halo-staged kernels need that, and since round 6 the ranges may be uneven: FlowNet's 1026- / 770- / 386-
    // channel decoder layers have 33 / 25 / 13 chunks, no even split exists, and unsplit they filled half of the chip.
    const int ntap_all = d.KH * d.KW;
    const int nch = d.korder == 1 ? ksteps / ntap_all : 0;
    const bool chunk_split = d.ksplit > 1 && d.korder == 1 && nch > 0 && cdiv(nch, cdiv(nch, d.ksplit)) == d.ksplit;
    const int per_split = chunk_split ? cdiv(nch, d.ksplit) * ntap_all : cdiv(ksteps, d.ksplit);
    const int nsplit_eff = cdiv(ksteps, per_split);
    if (nsplit_eff != d.ksplit) return VPS_EARG(20);  // caller must pick ksplit | ceil-consistent
    const long nblk = (long)tiles_m * tiles_n * d.nclass * d.ksplit;
    if (nblk <= 0 || nblk > 0x7fffffffL) return VPS_EARG(21);
    // stride-1 3x3 / 2x2-class layers on whole 8x16 output patches: halo-staged kernel
    // (patches may overhang the right / bottom edge; used when that wastes less than a third of the computed rows)
    const long tiles2d = (long)d.N * ((d.Qh + 7) / 8) * ((d.Qw + 15) / 16);
    // split-K there is over whole 32-channel chunks: the k-steps of a split must be a whole number of chunks
    const int ntap = d.KH * d.KW;
    const bool halo = d.prec != VPS_PREC_F32 && !d.offset && d.stride == 1 && d.korder == 1 && d.KH == d.KW && (d.KH == 3 || d.KH == 2) &&
                      tiles2d * 128 * 2 <= (long)M * 3 && (d.ksplit == 1 || chunk_split);
    // 8-wave variant (256-row tiles, weights through LDS): 128-column layers in the modes whose two activation planes leave room
    // for the weight buffers, when the 8 x 32 patches waste little and there are enough tiles to give every CU one
    const long tiles2d8 = (long)d.N * ((d.Qh + 7) / 8) * ((d.Qw + 31) / 32);
    // VPS_H8_MIN_CHUNKS=n: layers with fewer than n 32-channel chunks stay on the 4-wave halo kernel (two blocks per CU: one
    // block's store drain overlaps the other's k loop; the 8-wave kernel holds a CU alone and its short-K tiles are mostly drain)
    // round 5, measured: `64->128 3x3 @512x1024` 0.303 -> 0.264 ms on the 4-wave kernel, layers with >= 4 chunks unchanged -> 3
    static const int h8_min_chunks = getenv("VPS_H8_MIN_CHUNKS") ? atoi(getenv("VPS_H8_MIN_CHUNKS")) : 3;
    const bool h8 = halo && BN == 128 && (d.prec == VPS_PREC_F16X3 || d.prec == VPS_PREC_BF16X3 || d.prec == VPS_PREC_BF16) &&
                    tiles2d8 * 256 * 2 <= (long)M * 3 && tiles2d8 * tiles_n * d.nclass * d.ksplit >= 256 && ksteps / ntap >= h8_min_chunks;
    // stride-2 3x3 / 5x5 layers on the phase-split 8-wave halo kernel (VPS_S2_HALO=0 in the environment switches it off: A/B runs)
    static const bool s2_enabled = !(getenv("VPS_S2_HALO") && getenv("VPS_S2_HALO")[0] == '0');
    const bool h8s2 = s2_enabled && (BN == 128 || BN == 64) && (d.prec == VPS_PREC_F16X3 || d.prec == VPS_PREC_BF16X3 || d.prec == VPS_PREC_BF16) && !d.offset &&
                      d.stride == 2 && d.nclass == 1 && d.korder == 1 && d.KH == d.KW && (d.KH == 3 || d.KH == 5) &&
                      d.pad_y[0] == d.KH / 2 && d.pad_x[0] == d.KW / 2 && tiles2d8 * 256 * 2 <= (long)M * 3 &&
                      tiles2d8 * tiles_n * d.ksplit >= 256 && (d.ksplit == 1 || chunk_split);
    // 5..16 output channels, stride-1 3x3 / 2x2-class layers with enough 8 x 32 patches: the 16x16x32 kernel (VPS_N16=0 switches it off)
    static const bool n16_enabled = !(getenv("VPS_N16") && getenv("VPS_N16")[0] == '0');
    // VPS_N32=0: layers with 17 .. 32 output channels stay on the 32-column halo kernel (A/B; the two-column-block instance is round 6's)
    const char* const n32_env = getenv("VPS_N32");
    const int n16_max_cout = (n32_env && n32_env[0] == '0') ? 16 : 32;
    const bool n16 = n16_enabled && BN == 32 && d.prec == VPS_PREC_F16X3 && halo && d.cout > 4 && d.cout <= n16_max_cout && d.cout_pad == 32 && d.ksplit == 1 &&
                     !d.gn_stats && tiles2d8 * 256 * 2 <= (long)M * 3 && tiles2d8 * d.nclass >= 256;
    if (n16) {
        vpsi_launch_conv_n16(d, tiles2d8, s);
    } else if (h8s2) {
        vpsi_launch_conv_h8s2(d, (int)tiles2d8, tiles_n, per_split / ntap, (long)tiles2d8 * tiles_n * d.ksplit, BN == 64 ? 64 : 128, s);
    } else if (h8) {
        vpsi_launch_conv_h8(d, (int)tiles2d8, tiles_n, per_split / ntap, (long)tiles2d8 * tiles_n * d.nclass * d.ksplit, s);
    } else if (halo) {
        const int tiles_m2 = (int)tiles2d;
        const long nblk2 = (long)tiles_m2 * tiles_n * d.nclass * d.ksplit;
        if (nblk2 > 0x7fffffffL) return VPS_EARG(21);
#define VPS_HALO_LAUNCH(MODE, K)                                                                                                  \
    hipLaunchKernelGGL((conv_mfma_bf16h_kernel<TM, TN, WAVES_M, WAVES_N, MODE, K, K>), dim3((unsigned)nblk2), dim3(256), 0, s, d, tiles_m2, \
                       tiles_n, per_split / ntap)
        if (d.prec == VPS_PREC_BF16) { if (d.KH == 3) VPS_HALO_LAUNCH(VPS_PREC_BF16, 3); else VPS_HALO_LAUNCH(VPS_PREC_BF16, 2); }
        else if (d.prec == VPS_PREC_BF16X3) { if (d.KH == 3) VPS_HALO_LAUNCH(VPS_PREC_BF16X3, 3); else VPS_HALO_LAUNCH(VPS_PREC_BF16X3, 2); }
        else if (d.prec == VPS_PREC_F16X3) { if (d.KH == 3) VPS_HALO_LAUNCH(VPS_PREC_F16X3, 3); else VPS_HALO_LAUNCH(VPS_PREC_F16X3, 2); }
        else { if (d.KH == 3) VPS_HALO_LAUNCH(VPS_PREC_BF16X6, 3); else VPS_HALO_LAUNCH(VPS_PREC_BF16X6, 2); }
#undef VPS_HALO_LAUNCH
    } else {
    // deformable layers: the pipelined kernel when the k order is chunk-major (every layer of the path; weights in fragment order),
    // the two-barrier kernel for the tap-major order (row-major weights: vps_hip.h)
    const bool dcn_pipe = true;                            // (tap-major deformable launches are refused by vps_conv2d)
    const bool tapmajor = d.korder == 0 && ntap > 1;      // small channel counts; a 1x1 layer is the one-tap case of the chunk-major order
#define VPS_CONV_LAUNCH(KERNEL)                                                                              \
    hipLaunchKernelGGL((KERNEL), dim3((unsigned)nblk), dim3(256), 0, s, d, M, tiles_m, tiles_n, per_split)
    // uniform-lead kernel (conv_q.hip: both operands two k-steps ahead, weights through LDS): every non-deformable layer of the
    // modes with two activation planes on 64- / 128-column tiles. VPS_UNIFORM_LEAD=0 in the environment switches back to the
    // pipelined kernel (A/B runs): bit 0 = chunk-major layers, bit 1 = tap-major layers.
    const bool q_done = BN >= 64 && vpsi_launch_conv_q(d, M, tiles_m, tiles_n, per_split, nblk, tapmajor, s);
    if (q_done) {
    } else if (d.prec == VPS_PREC_F32) {
        if (d.offset) VPS_CONV_LAUNCH((conv_mfma_f32_kernel<TM, TN, WAVES_M, WAVES_N, true>));
        else VPS_CONV_LAUNCH((conv_mfma_f32_kernel<TM, TN, WAVES_M, WAVES_N, false>));
    } else if (d.prec == VPS_PREC_BF16) {
        if (d.offset && dcn_pipe) VPS_CONV_LAUNCH((conv_mfma_bf16p_kernel<TM, TN, WAVES_M, WAVES_N, VPS_PREC_BF16, false, true>));
        else if (tapmajor) VPS_CONV_LAUNCH((conv_mfma_bf16p_kernel<TM, TN, WAVES_M, WAVES_N, VPS_PREC_BF16, true>));
        else VPS_CONV_LAUNCH((conv_mfma_bf16p_kernel<TM, TN, WAVES_M, WAVES_N, VPS_PREC_BF16, false>));
    } else if (d.prec == VPS_PREC_BF16X3) {
        if (d.offset && dcn_pipe) VPS_CONV_LAUNCH((conv_mfma_bf16p_kernel<TM, TN, WAVES_M, WAVES_N, VPS_PREC_BF16X3, false, true>));
        else if (tapmajor) VPS_CONV_LAUNCH((conv_mfma_bf16p_kernel<TM, TN, WAVES_M, WAVES_N, VPS_PREC_BF16X3, true>));
        else VPS_CONV_LAUNCH((conv_mfma_bf16p_kernel<TM, TN, WAVES_M, WAVES_N, VPS_PREC_BF16X3, false>));
    } else if (d.prec == VPS_PREC_F16X3) {
        if (d.offset && dcn_pipe) VPS_CONV_LAUNCH((conv_mfma_bf16p_kernel<TM, TN, WAVES_M, WAVES_N, VPS_PREC_F16X3, false, true>));
        else if (tapmajor) VPS_CONV_LAUNCH((conv_mfma_bf16p_kernel<TM, TN, WAVES_M, WAVES_N, VPS_PREC_F16X3, true>));
        else VPS_CONV_LAUNCH((conv_mfma_bf16p_kernel<TM, TN, WAVES_M, WAVES_N, VPS_PREC_F16X3, false>));
    } else {
        if (d.offset && dcn_pipe) VPS_CONV_LAUNCH((conv_mfma_bf16p_kernel<TM, TN, WAVES_M, WAVES_N, VPS_PREC_BF16X6, false, true>));
        else if (tapmajor) VPS_CONV_LAUNCH((conv_mfma_bf16p_kernel<TM, TN, WAVES_M, WAVES_N, VPS_PREC_BF16X6, true>));
        else VPS_CONV_LAUNCH((conv_mfma_bf16p_kernel<TM, TN, WAVES_M, WAVES_N, VPS_PREC_BF16X6, false>));
    }
#undef VPS_CONV_LAUNCH
    }
    int st = vps_launch_status();
    if (st) return st;
    if (d.ksplit > 1 && !d.tile_counter) {
        const size_t total = (size_t)d.nclass * M * d.cout;
        const bool vec = !((d.cout | d.cout_pad | d.out_ld | d.out_coff) & 3) && !((uintptr_t)d.out & 15) &&
                         (!d.res || (!((d.res_ld | d.res_coff) & 3) && !((uintptr_t)d.res & 15)));
        if (vec) hipLaunchKernelGGL(conv_splitk_reduce_kernel<4>, dim3(stream_grid((long)(total / 4), 256)), dim3(256), 0, s, d, M);
        else hipLaunchKernelGGL(conv_splitk_reduce_kernel<1>, dim3(stream_grid((long)total, 256)), dim3(256), 0, s, d, M);
        st = vps_launch_status();
    }
    return st;
}

}  // namespace

extern "C" int vps_conv2d(const vps_conv_desc* dp, void* stream) {
    if (!dp) return VPS_EARG(1);
    const vps_conv_desc& d = *dp;
    if (!d.in || !d.out) return VPS_EARG(2);
    if (d.prec < VPS_PREC_F32 || d.prec > VPS_PREC_F16X3) return VPS_EARG(12);
    if (d.prec == VPS_PREC_F32 ? !d.w : !d.w_split) return VPS_EARG(13);
    if ((d.in_ld & 3) || (d.in_coff & 3) || (d.cin_pad & 3) || d.cin_pad <= 0) return VPS_EARG(3);
    if ((d.kpad % BK) || d.kpad < d.KH * d.KW * d.cin_pad || (d.korder != 0 && d.korder != 1)) return VPS_EARG(4);
    if (d.korder == 1 && d.kpad != d.KH * d.KW * ((d.cin_pad + BK - 1) / BK) * BK) return VPS_EARG(14);
    if (d.tile_n != 32 && d.tile_n != 64 && d.tile_n != 128 && !(d.tile_n == 256 && d.offset && d.prec == VPS_PREC_F16X3 && d.korder == 1)) return VPS_EARG(5);
    if (d.cout_pad % d.tile_n || d.cout > d.cout_pad || d.cout <= 0) return VPS_EARG(6);
    if (d.nclass != d.os_y * d.os_x || d.nclass < 1 || d.os_y > 2 || d.os_x > 2) return VPS_EARG(7);
    if (d.ksplit < 1 || (d.ksplit > 1 && !d.ws)) return VPS_EARG(8);
    if (d.offset && (d.nclass != 1 || d.off_ld < 2 * d.KH * d.KW || d.KH * d.KW > 9 || d.H > 65535 || d.W > 65535)) return VPS_EARG(9);
    // deformable layers of the split-operand modes: chunk-major k order only (every layer of the path; the tap-major two-barrier kernel of
    // rounds 1-2 is gone) - a tap-major deformable layer runs in VPS_PREC_F32 (vps_amd/nhwc.py packs it that way)
    if (d.offset && d.prec != VPS_PREC_F32 && d.korder != 1) return VPS_EARG(17);
    if (((uintptr_t)d.in & 15) || ((uintptr_t)d.w & 15) || ((uintptr_t)d.w_split & 15)) return VPS_EARG(10);
    // the split-operand kernels address the input and the weights through 32-bit buffer offsets
    if (d.prec != VPS_PREC_F32 && (size_t)d.N * d.H * d.W * d.in_ld * sizeof(float) >= 0xFFFFFFF0ull) return VPS_EARG(16);
    // GroupNorm sums in the epilogue: the deformable kernel of the split-operand modes only, unsplit, float4 stores, groups of 4 | 8 | 16 ...
    if (d.gn_stats && (!d.offset || d.prec == VPS_PREC_F32 || d.ksplit != 1 || d.gn_rep < 1 || (d.gn_rep & (d.gn_rep - 1)) || (d.gn_cpg != 4 && (d.gn_cpg < 8 || (d.gn_cpg & 7))) || d.cout % d.gn_cpg ||
                       ((d.cout | d.out_ld | d.out_coff) & 3) || ((uintptr_t)d.out & 15) || d.res || ((uintptr_t)d.gn_stats & 7)))
        return VPS_EARG(15);
    const long Ml = (long)d.N * d.Qh * d.Qw;
    if (Ml <= 0 || Ml > 0x7fffffffL) return VPS_EARG(11);
    const int M = (int)Ml;
    hipStream_t s = (hipStream_t)stream;
    // narrow outputs (cout <= 4) in exact fp32 on the vector ALU: conv_small.hip
    if (d.cout <= 4 && d.prec == VPS_PREC_F32 && !d.offset && d.ksplit == 1 && d.tile_n == 32) {
        vpsi_launch_conv_small(d, M, s);
        return vps_launch_status();
    }
    // thin-input layers at full resolution (3 / 6 / 11 / 12 -> 64 channels) with their own weight packing: conv_thin.hip
    if (d.w_thin && vpsi_launch_conv_thin(d, s)) return vps_launch_status();
    // wave arrangement <TM, TN, WAVES_M, WAVES_N> of the 4 waves of a block. Weight fragments come from global memory (one 1 KB
    // load per fragment = 64 cycles of the CU's vector-memory pipe, tools/gapbench.hip), activation fragments from LDS (two
    // conflict-free 1 KB reads per 32 cycles are free). Measured per layer (profiles/r02_wave_arrangement_ab.txt): for 64-column
    // tiles 2x2 waves of 64 rows x 32 columns beat 4x1 waves of 32 x 64 (half the weight loads: +8..18 %); for 128-column tiles
    // the 64 x 64 wave tile stays: 1x4 waves of 128 x 32 halve the weight loads again but double the fragment reads of the halo
    // tile, whose 18-row pitch costs a 2-way bank conflict (-7..13 %).
    if (d.tile_n == 256) {
        // deformable layers with >= 256 output channels (round 5): ONE block computes all 256 columns of its 128 pixels - the bilinear
        // loader (4 corner loads + 16 multiply-adds + the fp16 split per staged float4: 8.3 VALU per MFMA with 128 columns) runs once
        // instead of twice. 2 x 2 waves of 64 x 128, accumulators in AGPRs, one block per CU.
        const int tiles_m = cdiv(M, BM), tiles_n = d.cout_pad / 256, ksteps = d.kpad / BK;
        const int per_split = cdiv(ksteps, d.ksplit);
        if (cdiv(ksteps, per_split) != d.ksplit) return VPS_EARG(20);
        const long nblk = (long)tiles_m * tiles_n * d.ksplit;
        hipLaunchKernelGGL((conv_mfma_bf16p_kernel<2, 4, 2, 2, VPS_PREC_F16X3, false, true>), dim3((unsigned)nblk), dim3(256), 0, s, d, M, tiles_m, tiles_n, per_split);
        int st = vps_launch_status();
        if (st || d.ksplit == 1 || d.tile_counter) return st;
        const size_t total = (size_t)M * d.cout;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel<4>, dim3(stream_grid((long)(total / 4), 256)), dim3(256), 0, s, d, M);
        return vps_launch_status();
    }
    switch (d.tile_n) {
        case 128: return launch_conv<2, 2, 2, 2>(d, M, s);
        case 64: return launch_conv<2, 1, 2, 2>(d, M, s);
        default: return launch_conv<1, 1, 4, 1>(d, M, s);
    }
}
