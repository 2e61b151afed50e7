// "h8p" (round 6): the 8-wave halo-staged stride-1 3x3 / 2x2 kernel of conv_h8.hip with every operand ONE STAGE FURTHER AHEAD.
// Knock-outs of conv_mfma_h8_kernel on `256->256 3x3 @256x512` (tools/gpu_calls_r06/call_h8ko.sh): without the activation loads -14 %,
// without the weight loads -17 %, without both and their staging -30 %, without the per-tap barrier only -4 % - and the bare core
// (fragment reads + MFMAs) at 0.56 of the matrix pipe. The kernel was not short of bandwidth anywhere; each operand simply arrived late:
//   * the weights of tap s+1 were requested at the start of tap s and waited for at its end (an L2 round trip per tap);
//   * the activation rows of chunk c+2 were requested in the last tap of chunk c and the first of them staged half a tap later (an HBM
//     round trip per chunk);
//   * a tap's first fragments were read from LDS right behind the barrier, the matrix pipe idle until they arrived.
// Here:
//   * weights: requested THREE taps ahead into one of three register sets (index = tap % 3, static in the unrolled tap loop: 9 % 3 = 0),
//     staged two taps ahead into a ring of three LDS slots; only the two packed planes are loaded, the f16x3 mode's third plane
//     (2^-11 * plane 0) is made by the staging thread (a third less weight traffic through the vector-memory pipe);
//   * activations: the rows of chunk c+1 are staged in the FIRST half of chunk c's taps, the loads of chunk c+2 go out right behind
//     them - half a chunk (4-5 taps) of flight instead of half a tap;
//   * fragments: always one 16-k slab ahead of the MFMAs that use them, ACROSS the barrier - the second slab of a tap is read under
//     the MFMAs of the first, the first slab of the next tap under the MFMAs of the second (its weights sit in the ring since the
//     tap before, its activations in the chunk buffer); reads and MFMAs alternate one to one in program order.
// Accumulation order per output element is conv_mfma_h8_kernel's: results are bitwise equal (tests/test_hip_ops.py).
// LDS: activations 2 buffers x 2 planes x 340 rows x 64 B = 85 KB (rows beyond the halo tile are not kept any more), weights 3 slots x
// 24 KB = 72 KB: 157 KB of the CU's 160.
#include "conv_common.h"

namespace {

template <int N, typename F>
__device__ __forceinline__ void static_for_p(F&& f) {
    if constexpr (N > 0) {
        static_for_p<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

template <int MODE, int KH, int KW>
__global__ __launch_bounds__(512, 2)
void conv_mfma_h8p_kernel(const vps_conv_desc d, const int tiles_m, const int tiles_n, const int chunks_per_split) {
    constexpr int TM = 2, TN = 2, WAVES_N = 2, BN = 128;
    constexpr int NTAP = KH * KW;
    static_assert(NTAP % 3 == 0 || NTAP == 4, "ring / register-set index of a tap must be static: 9 taps (9 % 3 == 0) or 4 taps (12-step cycle handled by the caller)");
    constexpr int HW = 32 + KW - 1, HH = 8 + KH - 1;   // halo tile of an 8 x 32 patch
    constexpr int HROWS = HH * HW;                      // 340 (3x3), 297 (2x2)
    constexpr int NLD = (HROWS + 63) / 64;              // staged rows per thread (64 rows per pass of the 512 threads)
    typedef Split<MODE> SM;
    typedef typename SM::elem elem_t;
    typedef vec8<elem_t> x8;
    typedef vec4<elem_t> x4;
    constexpr int NSA = SM::NSA, NSB = SM::NSB, NLB = SM::NLB, NT = SM::NT;
    constexpr int PLANE = HROWS * LDS_LDH;              // 16-bit elements of one plane of one activation buffer
    constexpr int ABUF = NSA * PLANE;
    constexpr int NFRAG = NSB * 2 * (BN / 32);          // 1 KB weight fragments of one tap of the block tile, all planes
    constexpr int BBUF = NFRAG * 512;
    static_assert(NFRAG * 64 == NSB * 512, "thread t stages chunk t of every plane");
    static_assert((2 * ABUF + 3 * BBUF) * 2 <= 160 * 1024, "LDS budget of the CU");

    __shared__ __attribute__((aligned(16))) elem_t As[2 * ABUF];
    __shared__ __attribute__((aligned(16))) elem_t Bs[3 * BBUF];

    const int t = threadIdx.x;
    int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    int tile_n, tile_m, cls, split;             // split: split-K over whole 32-channel chunks
    decode_tile(d, swz, tiles_n, tiles_m, tile_n, tile_m, cls, split);

    const int py = cls / d.os_x, px = cls - py * d.os_x;
    const int H = d.H, W = d.W, cin_pad = d.cin_pad;
    const int tiles_x = (d.Qw + 31) >> 5, tiles_y = (d.Qh + 7) >> 3;
    const int tx = tile_m % tiles_x, tq = tile_m / tiles_x;
    const int ty = tq % tiles_y, n = tq / tiles_y;
    const int iy_org = ty * 8 - d.pad_y[py], ix_org = tx * 32 - d.pad_x[px];   // input position of halo row 0, column 0

    const int k4 = t & 7;      // 4-channel group staged by this thread (8 lanes = one 128-byte line)
    const int r0 = t >> 3;     // halo rows r0 + 64 i
    const int chunk0 = split * chunks_per_split;
    const int nchunks = min(chunks_per_split, d.kpad / (BK * NTAP) - chunk0);
    const int nsteps = nchunks * NTAP;

    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;     // wm 0..3: output rows 2 wm, 2 wm + 1 of the patch
    const int nbt = d.cout_pad >> 5, kst = d.kpad >> 4;
    const size_t wplane = (size_t)d.nclass * nbt * kst * 512;
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(d.w_split, (unsigned)(wplane * NSB * sizeof(elem_t)));
    const unsigned wbase = (unsigned)((((size_t)(cls * nbt + tile_n * (BN / 32)) * kst + 2 * (size_t)chunk0 * NTAP) * 512) * sizeof(elem_t));
    const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(d.in, (unsigned)((size_t)d.N * H * W * d.in_ld * sizeof(float)));
    const unsigned ld4 = (unsigned)d.in_ld * 4u;

    f32x4 areg[NLD];
    int achunk = chunk0;       // next chunk to load
    x8 breg[3][NLB];           // weights in flight: set s % 3 holds the loaded planes of tap s + 3 from the start of tap s to the end of tap s + 1
    float amax = 0.f;

    // byte offset of halo position r0 + 64 i (k-invariant), 0xFFFFFFF0 when it lies outside the halo / the image
    unsigned hoff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int hp = r0 + 64 * i;
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
    auto store_A = [&](const int i, const int buf) {
        x4 sp[NSA];
        split_act<MODE>(areg[i], sp, amax);
        const int row = r0 + 64 * i;
        if (row < HROWS) {
#pragma unroll
            for (int p = 0; p < NSA; ++p)
                *reinterpret_cast<x4*>(&As[buf * ABUF + p * PLANE + row * LDS_LDH + (((k4 >> 1) ^ lds_swz(row)) << 3) + ((k4 & 1) << 2)]) = sp[p];
        }
    };
    // this thread's 16-byte chunk t of fragment f = t / 64 = slab * (BN/32) + column block of EVERY plane (fragment index wave-uniform:
    // scalar offset + lane * 16). Steps past the end are clamped (loaded, staged, never read).
    auto load_B = [&](const int step, const int set) {
        const int sc = min(step, nsteps - 1);
        const int f = wave, bcol = f % (BN / 32), slab = f / (BN / 32);
#pragma unroll
        for (int p = 0; p < NLB; ++p)
            breg[set][p] = buffer_load16<x8>(wrsrc, (unsigned)lane * 16u,
                                             wbase + (unsigned)(((size_t)p * wplane + ((size_t)bcol * kst + 2 * sc + slab) * 512) * sizeof(elem_t)));
    };
    auto store_B = [&](const int slot, const int set) {
#pragma unroll
        for (int p = 0; p < NLB; ++p) *reinterpret_cast<x8*>(&Bs[slot * BBUF + (t + 512 * p) * 8]) = breg[set][p];
        if constexpr (NLB < NSB) *reinterpret_cast<x8*>(&Bs[slot * BBUF + (t + 512 * NLB) * 8]) = derive_weight_plane<MODE>(breg[set][0]);
    };

    // halo row of tile row j = wm*64 + a*32 + (lane&31) = patch row 2 wm + a, column lane&31, for tap (0,0); tap (ky,kx) adds ky*HW + kx
    int hbase[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) hbase[a] = (wm * TM + a) * HW + (lane & 31);
    x8 af[2][NSA][TM];
    x8 bcur[2][NSB][TN];
    // fragment reads, ONE at a time (they are dealt out between the MFMAs): index j < NRD of slab m
    constexpr int NRD = NSA * TM + NSB * TN;
    auto read_frag = [&](const int j, const int m, const int abuf, const int toff, const int slot) {
        if (j < NSA * TM) {
            const int p = j / TM, a = j - p * TM;
            const int hrow = hbase[a] + toff;
            af[m][p][a] = *reinterpret_cast<const x8*>(&As[abuf * ABUF + p * PLANE + hrow * LDS_LDH + (((2 * m + (lane >> 5)) ^ lds_swz(hrow)) << 3)]);
        } else {
            const int jj = j - NSA * TM, p = jj / TN, b = jj - p * TN;
            bcur[m][p][b] = *reinterpret_cast<const x8*>(&Bs[slot * BBUF + (((p * 2 + m) * (BN / 32)) + wn * TN + b) * 512 + lane * 8]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // prologue: chunk 0 staged in activation buffer 0, chunk 1 in flight in registers; weights of taps 0 and 1 staged in slots 0 and 1,
    // those of tap 2 in flight in set 2 (tap 0 stages them), the first slab of tap 0 read
    load_A();
    load_B(0, 0);
    load_B(1, 1);
#pragma unroll
    for (int i = 0; i < NLD; ++i) store_A(i, 0);
    store_B(0, 0);
    store_B(1, 1);
    load_B(2, 2);
    load_A();
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NRD; ++j) read_frag(j, 0, 0, 0, 0);

    constexpr int NMS = NT * TM * TN;                           // MFMAs per wave and slab
    constexpr int STG = NTAP / 2;                               // the next chunk's rows are staged in taps 0 .. STG-1, the chunk after it requested in tap STG
    constexpr int SPT = (NLD + STG - 1) / STG;                  // rows staged per tap
    auto mfma = [&](const int m, const int i) {
        const int q = i / (TM * TN), r = i - q * (TM * TN), a = r / TN, b = r - a * TN;
        acc[a][b] = split_mfma<MODE>(bcur[m][SM::PB[q]][b], af[m][SM::PA[q]][a], acc[a][b]);
    };

    // one tap. R3: tap index mod 3 (ring slot / register set), static.
    auto tap = [&](const int chunk, auto tp_tag, auto r3_tag) {
        constexpr int tp = decltype(tp_tag)::value;
        constexpr int R3 = decltype(r3_tag)::value;
        const int cur = chunk & 1;
        const int step = chunk * NTAP + tp;
        constexpr int toff = (tp / KW) * HW + (tp % KW);
        constexpr int ntp = (tp + 1) % NTAP;
        constexpr int ntoff = (ntp / KW) * HW + (ntp % KW);
        const int nbuf = tp == NTAP - 1 ? cur ^ 1 : cur;
        // ---- slab 0: its MFMAs, the request for the weights of tap + 3, the fragment reads of slab 1
#pragma unroll
        for (int i = 0; i < NMS; ++i) {
            mfma(0, i);
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0) load_B(step + 3, R3);
#pragma unroll
            for (int j = 0; j < NRD; ++j)
                if ((j * (NMS - 1)) / NRD + 1 == i || (NMS == 1 && i == 0)) read_frag(j, 1, cur, toff, R3);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- slab 1: its MFMAs, the staging work (activation rows of the next chunk in the first taps, then the request for the chunk
        // after it; the weights of tap + 2, one tap old), the fragment reads of slab 0 of the NEXT tap
#pragma unroll
        for (int i = 0; i < NMS; ++i) {
            mfma(1, i);
            __builtin_amdgcn_sched_barrier(0);
            if (tp < STG) {
#pragma unroll
                for (int si = 0; si < SPT; ++si)
                    if (i == (si * NMS) / (SPT + 1) && tp * SPT + si < NLD) store_A(tp * SPT + si, cur ^ 1);
            }
            if (i == (SPT * NMS) / (SPT + 1)) {
                store_B((R3 + 2) % 3, (R3 + 2) % 3);                       // tap + 2: requested during tap - 1
                if (tp == STG) load_A();                                     // behind the weight staging: its wait leaves these in flight
            }
#pragma unroll
            for (int j = 0; j < NRD; ++j)
                if ((j * (NMS - 1)) / NRD + 1 == i || (NMS == 1 && i == 0)) read_frag(j, 0, nbuf, ntoff, (R3 + 1) % 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };

    if constexpr (NTAP % 3 == 0) {
        for (int chunk = 0; chunk < nchunks; ++chunk)
            static_for_p<NTAP>([&](auto tpc) {                   // unrolled over the taps: every tap has static offsets and ring indices
                constexpr int TP = decltype(tpc)::value;
                tap(chunk, std::integral_constant<int, TP>{}, std::integral_constant<int, TP % 3>{});
            });
    } else {
        // 4 taps per chunk: the ring index of a tap repeats every 3 chunks (12 steps)
        int chunk = 0;
        for (; chunk + 2 < nchunks; chunk += 3)
            static_for_p<3 * NTAP>([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                tap(chunk + S / NTAP, std::integral_constant<int, S % NTAP>{}, std::integral_constant<int, S % 3>{});
            });
        // tail of one or two chunks: chunk % 3 == 0 here, the ring phase of its first tap is 0
        if (chunk < nchunks) {
            static_for_p<NTAP>([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                tap(chunk, std::integral_constant<int, S>{}, std::integral_constant<int, S % 3>{});
            });
            ++chunk;
            if (chunk < nchunks)
                static_for_p<NTAP>([&](auto sc) {
                    constexpr int S = decltype(sc)::value;
                    tap(chunk, std::integral_constant<int, S>{}, std::integral_constant<int, (NTAP + S) % 3>{});
                });
        }
    }
    report_range<MODE>(d, amax);
    conv_epilogue<TM, TN, BN, true, 5>(d, acc, tiles_m * 256, tile_m, tile_n, cls, split, py, px, wm, wn, lane, (cls * tiles_m + tile_m) * tiles_n + tile_n);
}

}  // namespace

// -> 1 if the pipelined instance takes this launch and was enqueued, 0: the caller launches conv_mfma_h8_kernel.
// VPS_H8P=0 in the environment switches it off (A/B runs; read per call: tests compare the two kernels in one process)
__attribute__((visibility("hidden")))
int vpsi_launch_conv_h8p(const vps_conv_desc& d, int tiles_m8, int tiles_n, int chunks_per_split, long nblk8, hipStream_t s) {
    const char* const e = getenv("VPS_H8P");
    if (e && atoi(e) == 0) return 0;
    if (d.prec != VPS_PREC_F16X3 || d.KH != 3) return 0;       // (the 2x2 instance - three chunks of 4 taps unrolled - spills: conv_mfma_h8_kernel keeps those layers)
    hipLaunchKernelGGL((conv_mfma_h8p_kernel<VPS_PREC_F16X3, 3, 3>), dim3((unsigned)nblk8), dim3(512), 0, s, d, tiles_m8, tiles_n, chunks_per_split);
    return 1;
}
