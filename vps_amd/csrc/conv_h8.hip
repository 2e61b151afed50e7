// The 8-wave halo-staged convolution kernels ("h8": stride-1 3x3 / 2x2-class layers; "h8s2": stride-2 3x3 / 5x5 layers, phase-split) -
// the largest items of the frame (FPN / TCEA / FlowNet 3x3 layers at 256x512 and 128x256, the FlowNet 5x5 stride-2 layers). Own
// translation unit since round 6 (conv_mfma.hip takes minutes to compile); entered through vpsi_launch_conv_h8 / vpsi_launch_conv_h8s2
// from launch_conv in conv_mfma.hip, which decides WHEN a layer comes here.
#include "conv_common.h"

namespace {

// ================================================================================================
// Halo-staged, 8 wavefronts, weights shared through LDS ("h8"): the 128-column stride-1 3x3 / 2x2 layers in the 3-product
// modes (f16x3, bf16x3) and plain bf16, when the layer fills the chip with 256-row tiles.
// tools/gapbench.hip: a 1 KB global load occupies the CU's vector-memory pipe for 64 cycles, a 1 KB LDS read for 4-16. In the
// 4-wave kernel above every wave loads the 12 fragments (12 KB) of its 64 columns per tap from global memory: 2 blocks x 4 waves
// x 12 KB = 1536 vector-memory cycles per tap and CU - exactly the 2 x 24 x 32 MFMA cycles a SIMD spends on the tap in the
// 3-product modes (measured: matrix pipe 0.45 busy). Here ONE block of 8 waves (256 rows = an 8 x 32 patch, 128 columns) owns
// the CU: the 24 weight fragments of a tap are loaded once (3 x 16 bytes per thread) into a double-buffered LDS region and read
// by all 8 waves (lane-contiguous, conflict-free); a quarter of the global weight traffic. The 32-wide patch rows also make
// the activation-fragment reads conflict-free (32 consecutive halo rows per sub-tile instead of 2 x 16 rows 18 apart).
// Price: one barrier per tap (the weight buffers alternate per tap) instead of one per chunk.
// ================================================================================================
template <int MODE, int KH, int KW>
__global__ __launch_bounds__(512, 2)
void conv_mfma_h8_kernel(const vps_conv_desc d, const int tiles_m, const int tiles_n, const int chunks_per_split) {
    constexpr int TM = 2, TN = 2, WAVES_N = 2, BN = 128;
    constexpr int NTAP = KH * KW;
    constexpr int HW = 32 + KW - 1, HH = 8 + KH - 1;   // halo tile of an 8 x 32 patch
    constexpr int HROWS = HH * HW;                      // <= 340
    constexpr int NLD = (HROWS + 63) / 64;              // staged rows per thread (64 rows per pass of the 512 threads)
    typedef Split<MODE> SM;
    typedef typename SM::elem elem_t;
    typedef vec8<elem_t> x8;
    typedef vec4<elem_t> x4;
    constexpr int NSA = SM::NSA, NSB = SM::NSB;
    constexpr int PLANE = NLD * 64 * LDS_LDH;           // 16-bit elements of one plane of one activation buffer
    constexpr int ABUF = NSA * PLANE;
    // all packed planes are loaded and staged: deriving the f16x3 mode's third plane in registers (derive_weight_plane) saves a third
    // of the weight traffic but measured slower here (256->256 3x3 @256x512: 2.81 -> 2.88 ms; the tap loop is not short of load slots)
    constexpr int NLB = NSB;
    constexpr int NFRAG = NLB * 2 * (BN / 32);          // 1 KB weight fragments of one tap of the block tile
    constexpr int BBUF = NFRAG * 512;
    static_assert((NFRAG * 64) % 512 == 0, "whole 16-byte chunks per thread");
    constexpr int NBL = NFRAG * 64 / 512;               // 16-byte weight chunks per thread and tap
    static_assert(2 * (ABUF + BBUF) * 2 <= 160 * 1024, "LDS budget of the CU");

    __shared__ __attribute__((aligned(16))) elem_t As[2 * ABUF];
    __shared__ __attribute__((aligned(16))) elem_t Bs[2 * BBUF];

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
    // buffer-addressed loads (see the pipelined kernel): scalar weight offsets, zero-fill of the halo by an out-of-range offset
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(d.w_split, (unsigned)(wplane * NSB * sizeof(elem_t)));
    const unsigned wbase = (unsigned)((((size_t)(cls * nbt + tile_n * (BN / 32)) * kst + 2 * (size_t)chunk0 * NTAP) * 512) * sizeof(elem_t));
    const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(d.in, (unsigned)((size_t)d.N * H * W * d.in_ld * sizeof(float)));
    const unsigned ld4 = (unsigned)d.in_ld * 4u;

    f32x4 areg[NLD];
    int achunk = chunk0;       // next chunk to load
    x8 breg[NBL];
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
    auto store_A = [&](int i, int buf) {
        x4 sp[NSA];
        split_act<MODE>(areg[i], sp, amax);
        const int row = r0 + 64 * i;
#pragma unroll
        for (int p = 0; p < NSA; ++p)
            *reinterpret_cast<x4*>(&As[buf * ABUF + p * PLANE + row * LDS_LDH + (((k4 >> 1) ^ lds_swz(row)) << 3) + ((k4 & 1) << 2)]) = sp[p];
    };
    // this thread's 16-byte chunks c = t + 512 j of fragment f = c / 64 = (plane * 2 + slab) * (BN/32) + column block.
    // The fragment index is wave-uniform for a given j (64 chunks per fragment, 64 lanes per wave): scalar offset + lane * 16
    auto load_B = [&](int step) {
#pragma unroll
        for (int j = 0; j < NBL; ++j) {
            const int f = __builtin_amdgcn_readfirstlane((t + 512 * j) >> 6), bcol = f % (BN / 32), pm = f / (BN / 32);
            breg[j] = buffer_load16<x8>(wrsrc, (unsigned)lane * 16u,
                                        wbase + (unsigned)(((size_t)(pm >> 1) * wplane + ((size_t)bcol * kst + 2 * step + (pm & 1)) * 512) * sizeof(elem_t)));
        }
    };
    auto store_B = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NBL; ++j) *reinterpret_cast<x8*>(&Bs[buf * BBUF + (t + 512 * j) * 8]) = breg[j];
    };

    // halo row of tile row j = wm*64 + a*32 + (lane&31) = patch row 2 wm + a, column lane&31, for tap (0,0); tap (ky,kx) adds ky*HW + kx
    int hbase[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) hbase[a] = (wm * TM + a) * HW + (lane & 31);
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
    x8 bcur[2][NSB][TN];
    auto read_B = [&](int m, int buf) {
#pragma unroll
        for (int p = 0; p < NLB; ++p)
#pragma unroll
            for (int b = 0; b < TN; ++b)
                bcur[m][p][b] = *reinterpret_cast<const x8*>(&Bs[buf * BBUF + (((p * 2 + m) * (BN / 32)) + wn * TN + b) * 512 + lane * 8]);
        if constexpr (NLB < NSB) {
#pragma unroll
            for (int b = 0; b < TN; ++b) bcur[m][2][b] = derive_weight_plane<MODE>(bcur[m][0][b]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // prologue: chunk 0 staged in activation buffer 0, chunk 1 in flight in registers, weights of tap 0 staged in weight buffer 0
    load_A();
    load_B(0);
#pragma unroll
    for (int i = 0; i < NLD; ++i) store_A(i, 0);
    store_B(0);
    load_A();
    __syncthreads();

    constexpr int NT = SM::NT;
    constexpr int NMF = 2 * NT * TM * TN;                       // MFMAs per wave and tap
    constexpr int SPT = (NLD + NTAP - 2) / (NTAP - 1);          // halo rows staged per tap (the last tap issues the loads)
    constexpr int NW = SPT + 3;                                 // weight loads | slab-1 fragment reads | SPT stagings | weight stores

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cur = chunk & 1;
#pragma unroll
        for (int tp = 0; tp < NTAP; ++tp) {
            const int step = chunk * NTAP + tp;
            const int bb = step & 1;                            // weight buffer of this tap
            const int bstep = min(step + 1, nsteps - 1);        // weights of the next tap (clamped: the last prefetch is unused)
            const int toff = (tp / KW) * HW + (tp % KW);
            read_A(0, cur, toff);
            read_B(0, bb);
            __builtin_amdgcn_sched_barrier(0);

            auto work = [&](const int w) {
                if (w == 0) load_B(bstep);                               // next tap's weights -> registers (issued first)
                else if (w == 1) { read_A(1, cur, toff); read_B(1, bb); }   // fragments of the second slab
                else if (w < SPT + 2) {
                    const int si = w - 2;                                // 0 .. SPT-1
                    if (tp < NTAP - 1) {
                        const int row = tp * SPT + si;
                        if (row < NLD) store_A(row, cur ^ 1);
                    } else if (si == 0) load_A();
                } else store_B(bb ^ 1);                                  // next tap's weights -> LDS (their loads are most of a tap old)
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
            __syncthreads();                                    // weight buffers alternate per tap (and, after the last tap, the chunk's)
        }
    }
    report_range<MODE>(d, amax);
    conv_epilogue<TM, TN, BN, true, 5>(d, acc, tiles_m * 256, tile_m, tile_n, cls, split, py, px, wm, wn, lane, (cls * tiles_m + tile_m) * tiles_n + tile_n);
}

// ================================================================================================
// Stride-2 K x K layers (K = 3, 5) on the 8-wave halo structure ("h8s2"), phase-split staging. EXPERIMENTAL: dispatched only when
// the environment variable VPS_S2_HALO is set (launch_conv); not part of the measured configuration of round 2.
// A stride-2 conv is four stride-1 convs on the (row, column)-parity sub-images of the input: tap (ky, kx) = (2j + a, 2i + b) of
// output pixel (oy, ox) reads sub-image (a, b) at (oy + j, ox + i) (origin shifted by the padding). One stage of the k loop =
// (32-channel chunk, phase (a, b)): the 8 x 32 output patch's sub-image patch ((8 + J_a - 1) x (32 + I_b - 1) pixels, J_0 = I_0 =
// ceil(K/2), J_1 = I_1 = floor(K/2)) is staged ONCE in LDS like the stride-1 halo tile, then the J_a x I_b taps of the phase read
// their activation fragments from it at the row offset j * HW + i. 5x5: 9 + 6 + 6 + 4 taps on <= 340 staged rows each (54 rows
// per tap; the pipelined kernel stages 256 rows per tap and re-fetches every input element 6.25 times); 3x3: 4 + 2 + 2 + 1.
// Everything else - weights through LDS per tap, fragment layouts, interleaving, epilogue - is conv_mfma_h8_kernel's.
// ================================================================================================
struct S2Tap { int ph, j, i, idx, t, nt; };    // phase, tap (j, i) inside it, k index ky*K + kx, position t of nt taps of the phase
constexpr int s2_taps_1d(int K, int a) { return (K - a + 1) / 2; }
constexpr S2Tap s2_tap(int K, int ts) {
    int base = 0;
    for (int ph = 0; ph < 4; ++ph) {
        const int a = ph >> 1, b = ph & 1, J = s2_taps_1d(K, a), I = s2_taps_1d(K, b);
        if (ts < base + J * I) {
            const int t = ts - base, j = t / I, i = t - j * I;
            return S2Tap{ph, j, i, (2 * j + a) * K + 2 * i + b, t, J * I};
        }
        base += J * I;
    }
    return S2Tap{0, 0, 0, 0, 0, 1};
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

template <int MODE, int K, int BN = 128>
__global__ __launch_bounds__(512, 2)
void conv_mfma_h8s2_kernel(const vps_conv_desc d, const int tiles_m, const int tiles_n, const int chunks_per_split) {
    static_assert(BN == 128 || BN == 64, "two column waves of 64 or 32 columns");
    constexpr int TM = 2, TN = BN / 64, WAVES_N = 2;
    constexpr int NTAP = K * K;
    constexpr int J0 = s2_taps_1d(K, 0);                // taps per axis of the even phase (the larger one)
    constexpr int HW = 32 + J0 - 1, HH = 8 + J0 - 1;    // sub-image patch of an 8 x 32 output patch: 10 x 34 (K = 5), 9 x 33 (K = 3)
    constexpr int HROWS = HH * HW;
    constexpr int NLD = (HROWS + 63) / 64;
    typedef Split<MODE> SM;
    typedef typename SM::elem elem_t;
    typedef vec8<elem_t> x8;
    typedef vec4<elem_t> x4;
    constexpr int NSA = SM::NSA, NSB = SM::NSB;
    constexpr int PLANE = NLD * 64 * LDS_LDH;
    constexpr int ABUF = NSA * PLANE;
    constexpr int NLB = NSB;                            // all packed planes loaded (see the stride-1 kernel above)
    constexpr int NFRAG = NLB * 2 * (BN / 32);
    constexpr int NBL = (NFRAG * 64 + 511) / 512;       // 16-byte weight chunks per thread
    constexpr bool WHOLE = (NFRAG * 64) % 512 == 0;
    constexpr int BBUF = NBL * 512 * 8;                 // whole rounds of the 512 threads (>= NFRAG * 512)
    static_assert(2 * (ABUF + BBUF) * 2 <= 160 * 1024, "LDS budget of the CU");

    __shared__ __attribute__((aligned(16))) elem_t As[2 * ABUF];
    __shared__ __attribute__((aligned(16))) elem_t Bs[2 * BBUF];

    const int t = threadIdx.x;
    const int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    int tile_n, tile_m, cls, split;
    decode_tile(d, swz, tiles_n, tiles_m, tile_n, tile_m, cls, split);

    const int H = d.H, W = d.W, cin_pad = d.cin_pad;
    const int tiles_x = (d.Qw + 31) >> 5, tiles_y = (d.Qh + 7) >> 3;
    const int tx = tile_m % tiles_x, tq = tile_m / tiles_x;
    const int ty = tq % tiles_y, n = tq / tiles_y;
    const int iy_org = ty * 16 - d.pad_y[0], ix_org = tx * 64 - d.pad_x[0];   // input position of tap (0, 0) of the patch's first output

    const int k4 = t & 7;
    const int r0 = t >> 3;
    const int chunk0 = split * chunks_per_split;
    const int nchunks = min(chunks_per_split, d.kpad / (BK * NTAP) - chunk0);

    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int nbt = d.cout_pad >> 5, kst = d.kpad >> 4;
    const size_t wplane = (size_t)nbt * kst * 512;
    // buffer-addressed loads (see the pipelined kernel): scalar weight offsets, zero-fill of the patch by an out-of-range offset
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(d.w_split, (unsigned)(wplane * NSB * sizeof(elem_t)));
    const unsigned wbase = (unsigned)((((size_t)(tile_n * (BN / 32)) * kst + 2 * (size_t)chunk0 * NTAP) * 512) * sizeof(elem_t));
    const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(d.in, (unsigned)((size_t)d.N * H * W * d.in_ld * sizeof(float)));
    const unsigned ld4 = (unsigned)d.in_ld * 4u;

    f32x4 areg[NLD];
    int achunk = chunk0, aph = 0;      // next (chunk, phase) stage to load
    x8 breg[NBL];
    float amax = 0.f;

    // sub-image patch of the next stage -> registers (sequential: every call advances (chunk, phase))
    auto load_A = [&]() {
        const int a = aph >> 1, b = aph & 1;
        const int rows = 8 + ((K - a + 1) >> 1) - 1, cols = 32 + ((K - b + 1) >> 1) - 1;
        const int cic = achunk * BK + k4 * 4;
        const bool kv = cic < cin_pad;
        const unsigned coff = (unsigned)(d.in_coff + cic) * 4u;
        if (++aph == 4) { aph = 0; ++achunk; }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int hp = r0 + 64 * i;
            const int hy = hp / HW, hx = hp - hy * HW;
            const int iy = iy_org + a + 2 * hy, ix = ix_org + b + 2 * hx;
            const bool ok = kv && hy < rows && hx < cols && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            areg[i] = buffer_load16<f32x4>(arsrc, ok ? (unsigned)((n * H + iy) * W + ix) * ld4 + coff : 0xFFFFFFF0u, 0u);
        }
    };
    auto store_A = [&](int i, int buf) {
        x4 sp[NSA];
        split_act<MODE>(areg[i], sp, amax);
        const int row = r0 + 64 * i;
#pragma unroll
        for (int p = 0; p < NSA; ++p)
            *reinterpret_cast<x4*>(&As[buf * ABUF + p * PLANE + row * LDS_LDH + (((k4 >> 1) ^ lds_swz(row)) << 3) + ((k4 & 1) << 2)]) = sp[p];
    };
    // wstep = k-step pair index relative to this split: (chunk - chunk0) * NTAP + ky * K + kx
    auto load_B = [&](int wstep) {
#pragma unroll
        for (int j = 0; j < NBL; ++j) {
            // no branch in here (it would split the interleaved MFMA stream): when the fragments are not a whole number of rounds
            // (64 columns, 3 weight planes) the surplus waves re-load the last fragment into the padding of the buffer
            const int fr = __builtin_amdgcn_readfirstlane((t + 512 * j) >> 6), f = WHOLE ? fr : min(fr, NFRAG - 1);
            const int bcol = f % (BN / 32), pm = f / (BN / 32);
            breg[j] = buffer_load16<x8>(wrsrc, (unsigned)lane * 16u,
                                        wbase + (unsigned)(((size_t)(pm >> 1) * wplane + ((size_t)bcol * kst + 2 * wstep + (pm & 1)) * 512) * sizeof(elem_t)));
        }
    };
    auto store_B = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NBL; ++j)
            *reinterpret_cast<x8*>(&Bs[buf * BBUF + (t + 512 * j) * 8]) = breg[j];
    };

    int hbase[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) hbase[a] = (wm * TM + a) * HW + (lane & 31);
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
    x8 bcur[2][NSB][TN];
    auto read_B = [&](int m, int buf) {
#pragma unroll
        for (int p = 0; p < NLB; ++p)
#pragma unroll
            for (int b = 0; b < TN; ++b)
                bcur[m][p][b] = *reinterpret_cast<const x8*>(&Bs[buf * BBUF + (((p * 2 + m) * (BN / 32)) + wn * TN + b) * 512 + lane * 8]);
        if constexpr (NLB < NSB) {
#pragma unroll
            for (int b = 0; b < TN; ++b) bcur[m][2][b] = derive_weight_plane<MODE>(bcur[m][0][b]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // prologue: stage (chunk 0, phase 0) in activation buffer 0, stage (chunk 0, phase 1) in flight in registers, weights of the
    // first tap in weight buffer 0
    load_A();
    load_B(s2_tap(K, 0).idx);
#pragma unroll
    for (int i = 0; i < NLD; ++i) store_A(i, 0);
    store_B(0);
    load_A();
    __syncthreads();

    constexpr int NT = SM::NT;
    constexpr int NMF = 2 * NT * TM * TN;                       // MFMAs per wave and tap

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        static_for<NTAP>([&](auto ts_tag) {
            constexpr int TS = decltype(ts_tag)::value;         // position of the tap in the chunk's phase-major tap sequence
            constexpr S2Tap tp = s2_tap(K, TS);
            constexpr S2Tap nx = s2_tap(K, (TS + 1) % NTAP);
            constexpr int cur = tp.ph & 1;                      // four stages per chunk: the activation buffer is the phase's parity
            constexpr bool last_of_phase = tp.t == tp.nt - 1;
            constexpr int SPT = (NLD + tp.nt - 1) / tp.nt;      // rows of the next stage staged per tap of this phase
            constexpr int NW = 3 + SPT + (last_of_phase ? 1 : 0);   // weight loads | slab-1 fragment reads | SPT stagings | [next loads] | weight stores
            const int step = chunk * NTAP + TS;
            const int bb = step & 1;                            // weight buffer of this tap
            const int wnext = (TS + 1 < NTAP ? chunk : min(chunk + 1, nchunks - 1)) * NTAP + nx.idx;   // clamped: the last prefetch is unused
            constexpr int toff = tp.j * HW + tp.i;
            read_A(0, cur, toff);
            read_B(0, bb);
            __builtin_amdgcn_sched_barrier(0);

            auto work = [&](const int w) {
                if (w == 0) load_B(wnext);
                else if (w == 1) { read_A(1, cur, toff); read_B(1, bb); }
                else if (w < SPT + 2) {
                    const int row = tp.t * SPT + (w - 2);
                    if (row < NLD) store_A(row, cur ^ 1);
                } else if (last_of_phase && w == SPT + 2) load_A();          // the stage after next, into the registers just staged
                else store_B(bb ^ 1);
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
            __syncthreads();
        });
    }
    report_range<MODE>(d, amax);
    conv_epilogue<TM, TN, BN, true, 5>(d, acc, tiles_m * 256, tile_m, tile_n, cls, split, 0, 0, wm, wn, lane, (cls * tiles_m + tile_m) * tiles_n + tile_n);
}

}  // namespace

int vpsi_launch_conv_h8p(const vps_conv_desc& d, int tiles_m8, int tiles_n, int chunks_per_split, long nblk8, hipStream_t s);

// stride-1 K x K (K = 3, 2) layers with 128-column tiles: nblk8 blocks of 512 threads
__attribute__((visibility("hidden")))
void vpsi_launch_conv_h8(const vps_conv_desc& d, int tiles_m8, int tiles_n, int chunks_per_split, long nblk8, hipStream_t s) {
    // f16x3: the pipelined instance (conv_h8p.hip, round 6)
    if (vpsi_launch_conv_h8p(d, tiles_m8, tiles_n, chunks_per_split, nblk8, s)) return;
#define VPS_H8_LAUNCH(MODE, K)                                                                                                   \
    hipLaunchKernelGGL((conv_mfma_h8_kernel<MODE, K, K>), dim3((unsigned)nblk8), dim3(512), 0, s, d, tiles_m8, tiles_n, chunks_per_split)
    if (d.prec == VPS_PREC_BF16) { if (d.KH == 3) VPS_H8_LAUNCH(VPS_PREC_BF16, 3); else VPS_H8_LAUNCH(VPS_PREC_BF16, 2); }
    else if (d.prec == VPS_PREC_BF16X3) { if (d.KH == 3) VPS_H8_LAUNCH(VPS_PREC_BF16X3, 3); else VPS_H8_LAUNCH(VPS_PREC_BF16X3, 2); }
    else { if (d.KH == 3) VPS_H8_LAUNCH(VPS_PREC_F16X3, 3); else VPS_H8_LAUNCH(VPS_PREC_F16X3, 2); }
#undef VPS_H8_LAUNCH
}

// stride-2 K x K (K = 3, 5) layers, 128- or 64-column tiles
__attribute__((visibility("hidden")))
void vpsi_launch_conv_h8s2(const vps_conv_desc& d, int tiles_m8, int tiles_n, int chunks_per_split, long nblk8, int bn, hipStream_t s) {
#define VPS_H8S2_LAUNCH(MODE, K)                                                                                                 \
    do {                                                                                                                         \
        if (bn == 64) hipLaunchKernelGGL((conv_mfma_h8s2_kernel<MODE, K, 64>), dim3((unsigned)nblk8), dim3(512), 0, s, d, tiles_m8, tiles_n, chunks_per_split); \
        else hipLaunchKernelGGL((conv_mfma_h8s2_kernel<MODE, K, 128>), dim3((unsigned)nblk8), dim3(512), 0, s, d, tiles_m8, tiles_n, chunks_per_split);         \
    } while (0)
    if (d.prec == VPS_PREC_BF16) { if (d.KH == 3) VPS_H8S2_LAUNCH(VPS_PREC_BF16, 3); else VPS_H8S2_LAUNCH(VPS_PREC_BF16, 5); }
    else if (d.prec == VPS_PREC_BF16X3) { if (d.KH == 3) VPS_H8S2_LAUNCH(VPS_PREC_BF16X3, 3); else VPS_H8S2_LAUNCH(VPS_PREC_BF16X3, 5); }
    else { if (d.KH == 3) VPS_H8S2_LAUNCH(VPS_PREC_F16X3, 3); else VPS_H8S2_LAUNCH(VPS_PREC_F16X3, 5); }
#undef VPS_H8S2_LAUNCH
}
