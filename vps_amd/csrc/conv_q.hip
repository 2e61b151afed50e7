// Uniform-lead MFMA convolution kernels (round 4): see the block comment below. Own translation unit (conv_mfma.hip takes ~5 minutes
// to compile); entered through vpsi_launch_conv_q from launch_conv in conv_mfma.hip.
#include "conv_common.h"
#include <cstdio>

namespace {

// ================================================================================================
// Split-operand, UNIFORM-LEAD pipeline ("bf16q", round 4): what the pipelined kernel above runs into, and the fix.
//
// vmcnt retires loads IN ORDER. In the kernel above a k-step issues [activation loads of tile s+3] and then [weight fragments of
// step s+1]; the next step opens with the wait for its first weight fragment (`s_waitcnt vmcnt(9)` in the ISA), and that wait
// also retires every load issued BEFORE the fragment - the activation tile that was meant to fly for two more k-steps. Its real
// flight time is half a k-step (~0.2 us) against a loaded HBM latency of 1-3 us: every k-step of every layer exposes one memory
// latency (the "chain of memory latencies" of the 1x1 layers: 8 k-steps = 28 us per block for 6 us of matrix work).
// Any operand that is waited for every step bounds the flight time of everything issued before it, so here BOTH operands have
// the same lead: the weight fragments of a k-step go global -> registers -> LDS exactly like the activation tile (the fragment
// order makes that a linear 16-byte copy), requested two full k-steps before they are staged, and the step's fragments are read
// from LDS (ds_read_b128, lgkmcnt - a counter of its own). A step now waits only for loads that are two steps old.
// Side effects: the block's weight traffic on the vector-memory pipe halves (each fragment is loaded once per block, not once per
// column wave pair), 80 KB of LDS per block (two blocks per CU, exactly the 160 KB), modes with two activation planes only.
// ================================================================================================
template <int TM, int TN, int WAVES_M, int WAVES_N, int MODE, bool TAPMAJOR>
__global__ __launch_bounds__(256, 2)
void conv_mfma_bf16q_kernel(const vps_conv_desc d, const int M, const int tiles_m, const int tiles_n,
                            const int ksteps_per_split) {
    constexpr int BN = WAVES_N * TN * 32;
    static_assert(WAVES_M * TM * 32 == BM, "block M tile must be 128");
    static_assert(WAVES_M * WAVES_N == 4, "4 wavefronts per block");
    typedef Split<MODE> SM;
    typedef typename SM::elem elem_t;
    typedef vec8<elem_t> x8;
    typedef vec4<elem_t> x4;
    constexpr int NSA = SM::NSA, NSB = SM::NSB;
    static_assert(NSA <= 2, "two activation planes: 2 x (16 KB + 24 KB) = 80 KB of LDS per block, two blocks per CU");
    constexpr int ABUF = NSA * BM * LDS_LDH;     // 16-bit elements of one activation buffer (all planes)
    constexpr int NLB = SM::NLB;                 // weight planes loaded and staged; the others are derived in registers
    constexpr int NFRAG = NLB * 2 * (BN / 32);   // 1 KB weight fragments of one k-step of the block tile: (plane, slab, column block)
    constexpr int BBUF = NFRAG * 512;
    static_assert((NFRAG * 64) % 256 == 0, "whole 16-byte chunks per thread");
    constexpr int NBL = NFRAG * 64 / 256;        // 16-byte weight chunks per thread and k-step
    static_assert(2 * (ABUF + BBUF) * 2 <= 80 * 1024, "two blocks per CU");

    __shared__ __attribute__((aligned(16))) elem_t As[2 * ABUF];
    __shared__ __attribute__((aligned(16))) elem_t Bs[2 * BBUF];

    const int t = threadIdx.x;
    int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    int tile_n, tile_m, cls, split;
    decode_tile(d, swz, tiles_n, tiles_m, tile_n, tile_m, cls, split);

    const int py = cls / d.os_x, px = cls - py * d.os_x;
    const int pad_y = d.pad_y[py], pad_x = d.pad_x[px];
    const int H = d.H, W = d.W, KH = d.KH, KW = d.KW, cin_pad = d.cin_pad;
    const int k4 = t & 7;      // 4-channel group of the 32-wide k-step staged by this thread (8 lanes = one 128-byte line)
    const int r0 = t >> 3;     // rows r0 + 32 i of the tile

    const int kstep0 = split * ksteps_per_split;
    int nsteps = d.kpad / BK - kstep0;
    if (nsteps > ksteps_per_split) nsteps = ksteps_per_split;
    const int ntap = KH * KW;
    int ky = 0, kx = 0, chunk = 0, astep = kstep0;   // state of the next activation tile to load
    if constexpr (!TAPMAJOR) {
        chunk = kstep0 / ntap;
        const int tap = kstep0 - chunk * ntap;
        ky = tap / KW;
        kx = tap - ky * KW;
    }

    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int nbt = d.cout_pad >> 5, kst = d.kpad >> 4;
    const size_t wplane = (size_t)d.nclass * nbt * kst * 512;
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(d.w_split, (unsigned)(wplane * NSB * sizeof(elem_t)));
    const unsigned wbase = (unsigned)((((size_t)(cls * nbt + tile_n * (BN / 32)) * kst + 2 * (size_t)kstep0) * 512) * sizeof(elem_t));
    const unsigned wlane = (unsigned)lane * 16u;
    const unsigned in_bytes = (unsigned)((size_t)d.N * H * W * d.in_ld * sizeof(float));
    const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(d.in, in_bytes);
    const unsigned ld4 = (unsigned)d.in_ld * 4u;
    const unsigned acoff = (unsigned)(d.in_coff + k4 * 4) * 4u;      // this thread's 4-channel group inside a 32-channel chunk
    int riy0[4], rix0[4];
    unsigned rowoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = tile_m * BM + r0 + 32 * i;
        const int mm = m < M ? m : 0;
        const int qx = mm % d.Qw;
        const int tq = mm / d.Qw;
        const int qy = tq % d.Qh;
        const int n = tq / d.Qh;
        const int iy0 = qy * d.stride - pad_y;
        rix0[i] = qx * d.stride - pad_x;
        rowoff[i] = (unsigned)(n * H * W + iy0 * W + rix0[i]) * ld4 + acoff;   // wraps for rows outside: masked
        riy0[i] = m < M ? iy0 : -(1 << 24);                                        // rows past the end: every tap out of the image
    }

    // operands in flight in registers. Loads are issued at the END of a k-step, weights first: ... W(s+1), A(s+2), W(s+2), A(s+3) ...
    // Activation tile T waits in slot T & 1 from the end of step T-3 until the second half of step T-1 stages it (two k-steps of
    // flight: HBM); the weight fragments of step T wait in their single slot from the end of step T-2 until the end of step T-1 (one
    // k-step: they come from L2). Because vmcnt retires in order, the wait for A(s+1) leaves W(s+1) and A(s+2) in flight and the wait
    // for W(s+1) leaves A(s+2): nothing is retired before it is needed.
    f32x4 areg[2][4];
    x8 breg[NBL];
    float amax = 0.f;

    auto load_A = [&](const int slot) {
        int kyc, kxc;
        unsigned stepoff, kmask;      // byte offset = rowoff[i] (k-invariant, per thread) + the step's (tap, channel) part
        if constexpr (!TAPMAJOR) {
            kyc = ky; kxc = kx;
            // channel groups beyond cin_pad: all offset bits set -> beyond the buffer -> zeros (OR, not a select: no divergent branch)
            kmask = chunk * BK + k4 * 4 < cin_pad ? 0u : 0xFFFFFFF0u;
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
            kmask = tap < ntap ? 0u : 0xFFFFFFF0u;
            stepoff = (unsigned)(kyc * W + kxc) * ld4 + (unsigned)(cic - k4 * 4) * 4u;
        }
        ++astep;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = riy0[i] + kyc, ix = rix0[i] + kxc;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            areg[slot][i] = buffer_load16<f32x4>(arsrc, (ok ? rowoff[i] + stepoff : 0xFFFFFFF0u) | kmask, 0u);
        }
    };
    // this thread's 16-byte chunks c = t + 256 j of fragment f = c / 64 = wave + 4 j = (plane * 2 + slab) * (BN/32) + column block:
    // wave-uniform for a given j -> scalar offset + lane * 16. Steps past the end are clamped (loaded, staged, never read).
    auto load_B = [&](const int step) {
        const int sc = min(step, nsteps - 1);
#pragma unroll
        for (int j = 0; j < NBL; ++j) {
            const int f = wave + 4 * j, bcol = f % (BN / 32), pm = f / (BN / 32);
            breg[j] = buffer_load16<x8>(wrsrc, wlane, wbase + (unsigned)(((size_t)(pm >> 1) * wplane + ((size_t)bcol * kst + 2 * sc + (pm & 1)) * 512) * sizeof(elem_t)));
        }
    };
    auto store_A = [&](const int i, const int buf, const int slot) {
        x4 sp[NSA];
        split_act<MODE>(areg[slot][i], sp, amax);
        const int row = r0 + 32 * i;
#pragma unroll
        for (int p = 0; p < NSA; ++p)
            *reinterpret_cast<x4*>(&As[buf * ABUF + p * (BM * LDS_LDH) + row * LDS_LDH + (((k4 >> 1) ^ lds_swz(row)) << 3) + ((k4 & 1) << 2)]) = sp[p];
    };
    auto store_B = [&](const int j, const int buf) {
        *reinterpret_cast<x8*>(&Bs[buf * BBUF + (t + 256 * j) * 8]) = breg[j];
    };

    const int frag_row = (wm * TM * 32 + (lane & 31)) * LDS_LDH;
    const int frag_sw = lds_swz(lane & 31);
    const int frag_chunk[2] = {(((lane >> 5)) ^ frag_sw) << 3, ((2 + (lane >> 5)) ^ frag_sw) << 3};
    // fragments: activations of both slabs of the k-step (read when the step opens), weights of ONE slab - the planes of slab 1
    // replace those of slab 0 one by one, each right after the last product of slab 0 that multiplies with it (24 instead of 48 VGPRs;
    // with both operands' slabs resident the kernel sat at the 256-VGPR limit and the allocator's copies waited for loads in flight)
    x8 af[2][NSA][TM];
    x8 bf[NSB][TN];
    auto read_A = [&](const int m, const int buf) {
#pragma unroll
        for (int p = 0; p < NSA; ++p)
#pragma unroll
            for (int a = 0; a < TM; ++a)
                af[m][p][a] = *reinterpret_cast<const x8*>(&As[buf * ABUF + p * (BM * LDS_LDH) + a * 32 * LDS_LDH + frag_row + frag_chunk[m]]);
    };
    auto read_B = [&](const int m, const int p, const int buf) {
#pragma unroll
        for (int b = 0; b < TN; ++b)
            bf[p][b] = *reinterpret_cast<const x8*>(&Bs[buf * BBUF + (((p * 2 + m) * (BN / 32)) + wn * TN + b) * 512 + lane * 8]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // prologue: tile 0 staged in buffers 0; then A(1), W(1), A(2) - the order in which a step of the loop finds its loads in flight
    // (the compiler merges the wait counts of all paths into the loop: a different order here costs the loop its counted waits)
    if (nsteps > 0) {
        load_B(0);
        load_A(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) store_A(i, 0, 0);
#pragma unroll
        for (int j = 0; j < NBL; ++j) store_B(j, 0);
        load_A(1);
        load_B(1);
        load_A(0);
    }
    __syncthreads();

    constexpr int NT = SM::NT;
    constexpr int NM1 = NT * TM * TN;                // MFMAs per wave and slab
    constexpr int NW = 8;                            // work items of slab 1: 4 row stagings, the weight staging in two halves, the two requests

    auto kstep = [&](const int step, auto cur_tag) {
        constexpr int cur = decltype(cur_tag)::value;
        constexpr int slot = cur ^ 1;                  // tile step+1 waits there; tile step+3 goes there
        read_A(0, cur);
#pragma unroll
        for (int p = 0; p < NLB; ++p) read_B(0, p, cur);
        read_A(1, cur);
        __builtin_amdgcn_sched_barrier(0);
        // a derived weight plane (f16x3: plane 2 = 2^-11 * plane 0) is made from plane 0 right before the product that uses it: plane 0
        // then holds the current slab's fragments (slab 1's arrive behind the last product of slab 0, a product group = 4 MFMAs earlier)
        auto prep = [&](const int q) {
            if constexpr (NLB < NSB) {
                if (SM::PB[q] >= NLB) {
#pragma unroll
                    for (int b = 0; b < TN; ++b) bf[SM::PB[q]][b] = derive_weight_plane<MODE>(bf[0][b]);
                }
            }
        };

        // ---- slab 0: MFMAs, and after the last product that uses weight plane p, that plane's fragments of slab 1
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            prep(q);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = split_mfma<MODE>(bf[SM::PB[q]][b], af[0][SM::PA[q]][a], acc[a][b]);
            bool last_use = SM::PB[q] < NLB;               // derived planes are not refilled
#pragma unroll
            for (int q2 = q + 1; q2 < NT; ++q2) last_use = last_use && SM::PB[q2] != SM::PB[q];
            if constexpr (NLB < NSB) {
                // plane 0 is also the source of the derived plane: it has to outlive the product that multiplies with that one
#pragma unroll
                for (int q2 = q + 1; q2 < NT; ++q2) last_use = last_use && !(SM::PB[q] == 0 && SM::PB[q2] >= NLB);
            }
            if (last_use) {
                __builtin_amdgcn_sched_barrier(0);
                read_B(1, SM::PB[q], cur);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- slab 1: MFMAs with the step's non-MFMA work between them (program order pinned): stage tile step+1 (activations, then
        // weights), then request W(step+2) and A(step+3)
        auto work = [&](const int w) {
            if (w < 4) store_A(w, cur ^ 1, slot);
            else if (w == 4) {
#pragma unroll
                for (int j = 0; j < (NBL + 1) / 2; ++j) store_B(j, cur ^ 1);
            } else if (w == 5) {
#pragma unroll
                for (int j = (NBL + 1) / 2; j < NBL; ++j) store_B(j, cur ^ 1);
            } else if (w == 6) load_B(step + 2);
            else load_A(slot);
        };
        int mf = 0;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            prep(q);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = split_mfma<MODE>(bf[SM::PB[q]][b], af[1][SM::PA[q]][a], acc[a][b]);
                    ++mf;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        // item w after MFMA number ceil((w + 1) * NM1 / (NW + 1)) of the slab (several items share a gap when NM1 < NW)
                        const int pos = ((w + 1) * NM1 + NW) / (NW + 1);
                        if (mf == (pos < 1 ? 1 : (pos > NM1 ? NM1 : pos))) {
                            __builtin_amdgcn_sched_barrier(0);
                            work(w);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
        }
        __syncthreads();
    };
    // pairs of steps, then the odd one: with `if (step + 1 < nsteps)` inside the loop the compiler has to assume a path from the even
    // step straight back to the even step, on which the slot it stages would hold the NEWEST loads - its vmcnt waits then retire
    // everything in flight (vmcnt(9) instead of vmcnt(19) in the ISA of the first version) and the lead collapses
    int step = 0;
    for (; step + 1 < nsteps; step += 2) {
        kstep(step, std::integral_constant<int, 0>{});
        kstep(step + 1, std::integral_constant<int, 1>{});
    }
    if (step < nsteps) kstep(step, std::integral_constant<int, 0>{});
    report_range<MODE>(d, amax);
    conv_epilogue<TM, TN, BN, false, 4, false, true>(d, acc, M, tile_m, tile_n, cls, split, py, px, wm, wn, lane, (cls * tiles_m + tile_m) * tiles_n + tile_n,
                                                   reinterpret_cast<int*>(As));     // the k loop is over: the activation buffers are free
}

}  // namespace

int vpsi_launch_conv_pw(const vps_conv_desc& d, int M, int tiles_m, int tiles_n, hipStream_t s);

// -> 1 if a uniform-lead instance exists for this launch and was enqueued, 0 if the caller has to use another kernel
__attribute__((visibility("hidden")))
int vpsi_launch_conv_q(const vps_conv_desc& d_in, int M, int tiles_m, int tiles_n, int per_split, long nblk, bool tapmajor, hipStream_t s) {
    const vps_conv_desc& d = d_in;
    // 1x1 layers with at least two rounds of resident blocks: the persistent pointwise kernel (conv_pw.hip)
    if (!tapmajor && vpsi_launch_conv_pw(d, M, tiles_m, tiles_n, s)) return 1;
    // VPS_UNIFORM_LEAD in the environment (A/B runs): bit 0 = chunk-major layers (default on), bit 1 = tap-major layers (default off:
    // the thin first layers measured 12 .. 23 % slower here than on the pipelined kernel - 3 .. 7 k-steps per tile, the longer prologue
    // is not paid back)
    static const int q_mode = getenv("VPS_UNIFORM_LEAD") ? atoi(getenv("VPS_UNIFORM_LEAD")) : 1;
    if (!(q_mode & (tapmajor ? 2 : 1))) return 0;
    if (d.offset || !(d.prec == VPS_PREC_F16X3 || d.prec == VPS_PREC_BF16X3 || d.prec == VPS_PREC_BF16)) return 0;
    if (d.tile_n != 128 && d.tile_n != 64) return 0;
    // VPS_DEBUG_OCC=1: resident blocks per CU of the two f16x3 instances, once, on stderr (80 KB of LDS per block: two blocks need all 160 KB)
    static bool occ_done = false;
    if (!occ_done && getenv("VPS_DEBUG_OCC")) {
        occ_done = true;
        int n128 = -1, n64 = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n128, conv_mfma_bf16q_kernel<2, 2, 2, 2, VPS_PREC_F16X3, false>, 256, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n64, conv_mfma_bf16q_kernel<2, 1, 2, 2, VPS_PREC_F16X3, false>, 256, 0);
        fprintf(stderr, "[vps] uniform-lead kernel occupancy: %d blocks/CU (128-column tiles), %d (64-column tiles)\n", n128, n64);
    }
#define VPS_Q_LAUNCH(TN, MODE, TAP)                                                                                          \
    hipLaunchKernelGGL((conv_mfma_bf16q_kernel<2, TN, 2, 2, MODE, TAP>), dim3((unsigned)nblk), dim3(256), 0, s, d, M, tiles_m, tiles_n, per_split)
#define VPS_Q_MODE(MODE)                                                                                                     \
    do {                                                                                                                     \
        if (d.tile_n == 128) { if (tapmajor) VPS_Q_LAUNCH(2, MODE, true); else VPS_Q_LAUNCH(2, MODE, false); }               \
        else { if (tapmajor) VPS_Q_LAUNCH(1, MODE, true); else VPS_Q_LAUNCH(1, MODE, false); }                               \
    } while (0)
    if (d.prec == VPS_PREC_BF16) VPS_Q_MODE(VPS_PREC_BF16);
    else if (d.prec == VPS_PREC_BF16X3) VPS_Q_MODE(VPS_PREC_BF16X3);
    else VPS_Q_MODE(VPS_PREC_F16X3);
#undef VPS_Q_MODE
#undef VPS_Q_LAUNCH
    return 1;
}
