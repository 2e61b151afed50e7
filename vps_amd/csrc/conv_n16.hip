// The 16-column kernels of vps_conv2d (narrow-output layers at full resolution: the FlowNetFusion / FlowNetSD decoder). Own translation
// unit since round 6; entered through vpsi_launch_conv_n16 from launch_conv (conv_mfma.hip), which decides WHEN a layer comes here.
#include "conv_common.h"

namespace {

// ================================================================================================
// Narrow-output layers (5 <= cout <= 16: the full-resolution FlowNetFusion / FlowNetSD interconv and deconv layers) on the
// 16x16x32 MFMA shape. A 32-column tile of the 32x32x16 shape spends half of every MFMA and of every weight fragment on zero
// columns; here the WEIGHT fragment (16 output channels x 32 k) is the A operand and 16 consecutive pixels of a patch row
// x one whole 32-channel chunk are the B operand, so one MFMA is one (tap, chunk) of 16 pixels with no padding, and the
// accumulator (col = lane & 15 = pixel, rows 4 (lane >> 4) + r = 4 consecutive channels) is stored as one float4 per lane.
// Structure of conv_mfma_h8_kernel: 8 waves, an 8 x 32 output patch per block (wave w = patch row w, two 16-pixel groups), the
// halo tile of the patch staged once per 32-channel chunk (split into the planes of the arithmetic), but the weights of a WHOLE
// chunk (KH*KW taps x NSB planes x 1 KB) are staged per chunk: barriers per chunk, not per tap (a tap is only 6 MFMAs of 16 cycles
// per wave). A chunk of this kernel is short (54 MFMAs = 0.9k cycles against ~2k cycles of HBM latency) and a patch has only 3..6
// chunks, so the latency is hidden by a SECOND BLOCK on the CU rather than by a deeper software pipeline: ONE activation and ONE
// weight buffer in LDS (76 KB), the next chunk in flight in registers while this one is multiplied, <= 128 VGPRs (the
// double-buffered first version kept one block per CU and ran at the memory latency: 0.43 ms for 82 -> 16 @1024x2048). The 16-byte chunk index of an LDS row is XOR-ed with (row >> 1) & 3: the 16 lanes of a
// k-group read 16 consecutive rows, 8 of them cover the 8 distinct 16-byte slots of 128 bytes.
// The weight fragments come from the SAME packed layout as every other kernel ([plane][32-column block][16-k step][lane][8]):
// lane l of the 16x16x32 A fragment (channel l & 15, k-group g = l >> 4) is lane (l & 15) + 32 (g & 1) of step 2 s + (g >> 1).
// ================================================================================================
__device__ __forceinline__ int lds_swz16(int row) { return (row >> 1) & 3; }

template <int MODE, int KH, int KW>
__global__ __launch_bounds__(512, 4)
void conv_mfma_n16_kernel(const vps_conv_desc d, const int tiles_m) {
    constexpr int NTAP = KH * KW;
    constexpr int HW = 32 + KW - 1, HH = 8 + KH - 1;   // halo tile of an 8 x 32 patch
    constexpr int HROWS = HH * HW;
    constexpr int NLD = (HROWS + 63) / 64;              // staged rows per thread
    typedef Split<MODE> SM;
    typedef typename SM::elem elem_t;
    typedef vec8<elem_t> x8;
    typedef vec4<elem_t> x4;
    constexpr int NSA = SM::NSA, NSB = SM::NSB, NT = SM::NT;
    constexpr int PLANE = NLD * 64 * LDS_LDH;
    constexpr int ABUF = NSA * PLANE;
    constexpr int NWF = NTAP * NSB;                     // 1 KB weight fragments of one chunk
    constexpr int WBUF = NWF * 512;
    constexpr int NBL = (NWF * 64 + 511) / 512;         // 16-byte weight chunks per thread and chunk
    static_assert((ABUF + WBUF) * 2 <= 80 * 1024, "two blocks per CU");

    __shared__ __attribute__((aligned(16))) elem_t As[ABUF];
    __shared__ __attribute__((aligned(16))) elem_t Ws[WBUF];

    const int t = threadIdx.x;
    int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    int tile_n, tile_m, cls, split;
    decode_tile(d, swz, 1, tiles_m, tile_n, tile_m, cls, split);

    const int py = cls / d.os_x, px = cls - py * d.os_x;
    const int H = d.H, W = d.W, cin_pad = d.cin_pad;
    const int tiles_x = (d.Qw + 31) >> 5, tiles_y = (d.Qh + 7) >> 3;
    const int tx = tile_m % tiles_x, tq = tile_m / tiles_x;
    const int ty = tq % tiles_y, n = tq / tiles_y;
    const int iy_org = ty * 8 - d.pad_y[py], ix_org = tx * 32 - d.pad_x[px];

    const int k4 = t & 7;
    const int r0 = t >> 3;
    const int nchunks = d.kpad / (BK * NTAP);

    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // patch row
    const int nbt = d.cout_pad >> 5, kst = d.kpad >> 4;
    const size_t wplane = (size_t)d.nclass * nbt * kst * 512;
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(d.w_split, (unsigned)(wplane * NSB * sizeof(elem_t)));
    const unsigned wbase = (unsigned)(((size_t)(cls * nbt) * kst * 512) * sizeof(elem_t));
    // source of this lane's 16 bytes inside the two 32x16 fragments of a 32-k step
    const unsigned wlane = (unsigned)((((lane >> 5) & 1) * 512 + ((lane & 15) + 32 * ((lane >> 4) & 1)) * 8) * sizeof(elem_t));
    const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(d.in, (unsigned)((size_t)d.N * H * W * d.in_ld * sizeof(float)));
    const unsigned ld4 = (unsigned)d.in_ld * 4u;

    f32x4 areg[NLD];
    x8 wreg[NBL];
    int achunk = 0;
    float amax = 0.f;

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
        const unsigned coff = (unsigned)achunk * (BK * 4u);
        ++achunk;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            areg[i] = buffer_load16<f32x4>(arsrc, (kv && hoff[i] != 0xFFFFFFF0u) ? hoff[i] + coff : 0xFFFFFFF0u, 0u);
    };
    auto store_A = [&](int i) {
        x4 sp[NSA];
        split_act<MODE>(areg[i], sp, amax);
        const int row = r0 + 64 * i;
#pragma unroll
        for (int p = 0; p < NSA; ++p)
            *reinterpret_cast<x4*>(&As[p * PLANE + row * LDS_LDH + (((k4 >> 1) ^ lds_swz16(row)) << 3) + ((k4 & 1) << 2)]) = sp[p];
    };
    // fragment f = tap * NSB + plane of the chunk: wave-uniform per j
    auto load_W = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NBL; ++j) {
            const int f = __builtin_amdgcn_readfirstlane((t + 512 * j) >> 6);
            if (f < NWF) {
                const int tap = f / NSB, pl = f - tap * NSB;
                wreg[j] = buffer_load16<x8>(wrsrc, wlane, wbase + (unsigned)(((size_t)pl * wplane + (size_t)(2 * (chunk * NTAP + tap)) * 512) * sizeof(elem_t)));
            }
        }
    };
    auto store_W = [&](int j) {
        if (__builtin_amdgcn_readfirstlane((t + 512 * j) >> 6) < NWF) *reinterpret_cast<x8*>(&Ws[(t + 512 * j) * 8]) = wreg[j];
    };

    // B operand: pixel x = 16 h + (lane & 15) of patch row `wave`, k-group lane >> 4 = 16-byte chunk of the LDS row
    const int hbase = wave * HW + (lane & 15);
    const int kg = lane >> 4;
    f32x4 acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};

    // LDS element offsets of this lane's activation fragments, per (tap, pixel group): loop-invariant (the swizzle depends on the row)
    int aoff[NTAP][2];
#pragma unroll
    for (int tp = 0; tp < NTAP; ++tp)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int hrow = hbase + 16 * h + (tp / KW) * HW + (tp % KW);
            aoff[tp][h] = hrow * LDS_LDH + ((kg ^ lds_swz16(hrow)) << 3);
        }

    load_A();
    load_W(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        // the chunk in the registers -> LDS (the previous chunk's fragment reads are behind the barrier that ended its loop body)
#pragma unroll
        for (int i = 0; i < NLD; ++i) store_A(i);
#pragma unroll
        for (int j = 0; j < NBL; ++j) store_W(j);
        if (chunk + 1 < nchunks) {                               // next chunk in flight while this one is multiplied
            load_A();
            load_W(chunk + 1);
        }
        __syncthreads();
#pragma unroll
        for (int tp = 0; tp < NTAP; ++tp) {
            x8 wf[NSB], af[2][NSA];
#pragma unroll
            for (int p = 0; p < NSB; ++p) wf[p] = *reinterpret_cast<const x8*>(&Ws[(tp * NSB + p) * 512 + lane * 8]);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int p = 0; p < NSA; ++p) af[h][p] = *reinterpret_cast<const x8*>(&As[aoff[tp][h] + p * PLANE]);
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if constexpr (MODE == VPS_PREC_F16X3) acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[SM::PB[q]], af[h][SM::PA[q]], acc[h], 0, 0, 0);
                    else acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[SM::PB[q]], af[h][SM::PA[q]], acc[h], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    report_range<MODE>(d, amax);

    // epilogue: lane = pixel (lane & 15) of the group, output channels 4 (lane >> 4) .. + 3
    const int co = 4 * kg;
    const int qy = ty * 8 + wave;
    const bool v4 = !((d.out_ld | d.out_coff | d.cout) & 3) && !((uintptr_t)d.out & 15);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (co + e < d.cout) {
            if (d.scale) sc[e] = d.scale[co + e];
            if (d.shift) sh[e] = d.shift[co + e];
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int qx = tx * 32 + 16 * h + (lane & 15);
        if (qy < d.Qh && qx < d.Qw && co < d.cout) {
            const int oy = qy * d.os_y + py, ox = qx * d.os_x + px;
            const size_t opix = ((size_t)n * d.Ho + oy) * d.Wo + ox;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc[h][e] * sc[e] + sh[e];
            if (d.res) {
                const int rs = d.res_shift;
                const float* rp = d.res + (((size_t)n * (d.Ho >> rs) + (oy >> rs)) * (d.Wo >> rs) + (ox >> rs)) * d.res_ld + d.res_coff + co;
#pragma unroll
                for (int e = 0; e < 4; ++e) if (co + e < d.cout) o[e] += rp[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = vps_act(o[e], d.act, d.slope);
            float* op = d.out + opix * d.out_ld + d.out_coff + co;
            if (v4) *reinterpret_cast<f32x4*>(op) = o;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (co + e < d.cout) op[e] = o[e];
            }
        }
    }
}


// ================================================================================================
// The same for a stride-2 TRANSPOSED layer (4 x 4 kernel = four parity classes of 2 x 2 taps: FlowNetFusion deconv0 162->16 @512x1024,
// flow_modules/FlowNetFusion.py), ALL FOUR CLASSES IN ONE BLOCK (round 6). As four separate class launches of the kernel above the layer
// read, split and staged every input patch four times: 1.6 GB through the L2 for a 0.34 GB input, 0.33 ms - it ran at the memory rate.
// The taps of class (py, px) are input rows qy - pad_y[py] + {0, 1}: with pads (1, 0) the four classes together touch the 3 x 3
// neighbourhood of a position, i.e. the halo tile of a 3 x 3 layer. It is staged ONCE per 32-channel chunk; the 16 (class, tap) weight
// fragments of the chunk (two planes loaded, the third derived by the staging thread) sit beside it; every wave multiplies its patch
// row against all of them: 4 classes x 4 taps x 6 MFMAs, four accumulator pairs. One activation and one weight buffer (76 KB: two blocks
// per CU; the weights' third plane is derived when a fragment is read), the next chunk in flight in registers, like the kernel above. Per output element the accumulation order is that kernel's
// (chunk, tap, product): bitwise the same results.
// ================================================================================================
template <int MODE>
__global__ __launch_bounds__(512, 4)
void conv_mfma_n16t_kernel(const vps_conv_desc d, const int tiles_m) {
    constexpr int NTAP = 4, NCLS = 4;
    constexpr int HW = 32 + 2, HH = 8 + 2;              // the union of the four classes' 2 x 2 windows: a 3 x 3 halo
    constexpr int HROWS = HH * HW;
    constexpr int NLD = (HROWS + 63) / 64;
    typedef Split<MODE> SM;
    typedef typename SM::elem elem_t;
    typedef vec8<elem_t> x8;
    typedef vec4<elem_t> x4;
    constexpr int NSA = SM::NSA, NSB = SM::NSB, NLB = SM::NLB, NT = SM::NT;
    constexpr int PLANE = HROWS * LDS_LDH;
    constexpr int ABUF = NSA * PLANE;
    constexpr int NWL = NCLS * NTAP * NLB;              // 1 KB weight fragments LOADED per chunk: (class, tap, plane)
    constexpr int WBUF = NWL * 512;                     // ... and staged: the derived plane is made from plane 0 when a fragment is read
    static_assert(NWL * 64 % 512 == 0, "whole 16-byte chunks per thread");
    constexpr int NBL = NWL * 64 / 512;                 // 16-byte weight chunks per thread and chunk
    static_assert((ABUF + WBUF) * 2 <= 80 * 1024, "two blocks per CU");

    __shared__ __attribute__((aligned(16))) elem_t As[ABUF];
    __shared__ __attribute__((aligned(16))) elem_t Ws[WBUF];

    const int t = threadIdx.x;
    const int tile_m = xcd_swizzle(blockIdx.x, gridDim.x);
    const int H = d.H, W = d.W, cin_pad = d.cin_pad;
    const int tiles_x = (d.Qw + 31) >> 5, tiles_y = (d.Qh + 7) >> 3;
    const int tx = tile_m % tiles_x, tq = tile_m / tiles_x;
    const int ty = tq % tiles_y, n = tq / tiles_y;
    const int iy_org = ty * 8 - 1, ix_org = tx * 32 - 1;

    const int k4 = t & 7;
    const int r0 = t >> 3;
    const int nchunks = d.kpad / (BK * NTAP);

    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // patch row
    const int nbt = d.cout_pad >> 5, kst = d.kpad >> 4;
    const size_t wplane = (size_t)d.nclass * nbt * kst * 512;
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(d.w_split, (unsigned)(wplane * NSB * sizeof(elem_t)));
    // source of this lane's 16 bytes inside the two 32x16 fragments of a 32-k step (see the kernel above)
    const unsigned wlane = (unsigned)((((lane >> 5) & 1) * 512 + ((lane & 15) + 32 * ((lane >> 4) & 1)) * 8) * sizeof(elem_t));
    const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(d.in, (unsigned)((size_t)d.N * H * W * d.in_ld * sizeof(float)));
    const unsigned ld4 = (unsigned)d.in_ld * 4u;

    f32x4 areg[NLD];
    x8 wreg[NBL];
    int achunk = 0;
    float amax = 0.f;

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
        const unsigned coff = (unsigned)achunk * (BK * 4u);
        ++achunk;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            areg[i] = buffer_load16<f32x4>(arsrc, (kv && hoff[i] != 0xFFFFFFF0u) ? hoff[i] + coff : 0xFFFFFFF0u, 0u);
    };
    auto store_A = [&](int i) {
        x4 sp[NSA];
        split_act<MODE>(areg[i], sp, amax);
        const int row = r0 + 64 * i;
        if (row < HROWS) {
#pragma unroll
            for (int p = 0; p < NSA; ++p)
                *reinterpret_cast<x4*>(&As[p * PLANE + row * LDS_LDH + (((k4 >> 1) ^ lds_swz16(row)) << 3) + ((k4 & 1) << 2)]) = sp[p];
        }
    };
    // fragment g = (class * NTAP + tap) * NLB + plane of the chunk: wave-uniform per j
    auto load_W = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NBL; ++j) {
            const int g = __builtin_amdgcn_readfirstlane((t + 512 * j) >> 6);
            const int ct = g / NLB, pl = g - ct * NLB, cls = ct / NTAP, tap = ct - cls * NTAP;
            wreg[j] = buffer_load16<x8>(wrsrc, wlane, (unsigned)(((size_t)pl * wplane + ((size_t)(cls * nbt) * kst + (size_t)(2 * (chunk * NTAP + tap))) * 512) * sizeof(elem_t)));
        }
    };
    auto store_W = [&](int j) { *reinterpret_cast<x8*>(&Ws[(t + 512 * j) * 8]) = wreg[j]; };

    const int hbase = wave * HW + (lane & 15);
    const int kg = lane >> 4;
    f32x4 acc[NCLS][2];
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[c][h] = f32x4{0.f, 0.f, 0.f, 0.f};

    // class (py, px), tap (ky, kx) reads the input at q - pad[p] + k: window position 1 - pad[p] + k of the 3 x 3 neighbourhood
    // (pads are 0 or 1: checked by the launcher). The row offset of a (class, tap) is wave-uniform (scalar registers); the LDS offset
    // of a fragment is made from it when it is read (a table of the 32 offsets cost the second block per CU its registers)
    const int dy0 = 1 - d.pad_y[0], dy1 = 1 - d.pad_y[1], dx0 = 1 - d.pad_x[0], dx1 = 1 - d.pad_x[1];

    load_A();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        // the chunk's weights are requested HERE (L2-resident, 32 KB per block) and land while the activation rows are split and
        // staged: held in registers through the MFMA phase they cost the second block per CU (128 registers per lane)
        load_W(chunk);
#pragma unroll
        for (int i = 0; i < NLD; ++i) store_A(i);
#pragma unroll
        for (int j = 0; j < NBL; ++j) store_W(j);
        if (chunk + 1 < nchunks) load_A();                       // next chunk's activations in flight while this one is multiplied
        __syncthreads();
#pragma unroll
        for (int cls = 0; cls < NCLS; ++cls)
#pragma unroll
            for (int tp = 0; tp < NTAP; ++tp) {
                x8 wf[NSB], af[2][NSA];
#pragma unroll
                for (int p = 0; p < NLB; ++p) wf[p] = *reinterpret_cast<const x8*>(&Ws[(((cls * NTAP + tp) * NLB + p) * 64 + lane) * 8]);
                if constexpr (NLB < NSB) wf[NLB] = derive_weight_plane<MODE>(wf[0]);
                const int rdelta = (((cls >> 1) ? dy1 : dy0) + (tp >> 1)) * HW + ((cls & 1) ? dx1 : dx0) + (tp & 1);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int hrow = hbase + 16 * h + rdelta;
                    const int off = hrow * LDS_LDH + ((kg ^ lds_swz16(hrow)) << 3);
#pragma unroll
                    for (int p = 0; p < NSA; ++p) af[h][p] = *reinterpret_cast<const x8*>(&As[off + p * PLANE]);
                }
#pragma unroll
                for (int q = 0; q < NT; ++q)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        acc[cls][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[SM::PB[q]], af[h][SM::PA[q]], acc[cls][h], 0, 0, 0);
            }
        __syncthreads();
    }
    report_range<MODE>(d, amax);

    // epilogue: lane = pixel (lane & 15) of the group, output channels 4 (lane >> 4) .. + 3, for each of the four parity classes
    const int co = 4 * kg;
    const int qy = ty * 8 + wave;
    const bool v4 = !((d.out_ld | d.out_coff | d.cout) & 3) && !((uintptr_t)d.out & 15);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (co + e < d.cout) {
            if (d.scale) sc[e] = d.scale[co + e];
            if (d.shift) sh[e] = d.shift[co + e];
        }
    }
#pragma unroll
    for (int cls = 0; cls < NCLS; ++cls) {
        const int py = cls >> 1, px = cls & 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int qx = tx * 32 + 16 * h + (lane & 15);
            if (qy < d.Qh && qx < d.Qw && co < d.cout) {
                const int oy = qy * 2 + py, ox = qx * 2 + px;
                const size_t opix = ((size_t)n * d.Ho + oy) * d.Wo + ox;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[cls][h][e] * sc[e] + sh[e];
                if (d.res) {
                    const int rs = d.res_shift;
                    const float* rp = d.res + (((size_t)n * (d.Ho >> rs) + (oy >> rs)) * (d.Wo >> rs) + (ox >> rs)) * d.res_ld + d.res_coff + co;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (co + e < d.cout) o[e] += rp[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = vps_act(o[e], d.act, d.slope);
                float* op = d.out + opix * d.out_ld + d.out_coff + co;
                if (v4) *reinterpret_cast<f32x4*>(op) = o;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (co + e < d.cout) op[e] = o[e];
                }
            }
        }
    }
}


// ================================================================================================
// 17 .. 32 output channels (FlowNetFusion's 162->32 3x3 @512x1024; round 6): the 16-column structure with TWO 16-channel column blocks
// per wave - the activation fragments of a tap are read once and multiplied with both weight fragments. On the 32-column instance of
// the 4-wave halo kernel (8 x 16 patches: 1.41x halo traffic, one 32x32 tile per wave with every weight fragment loaded per wave from
// global memory) the layer took 0.27 ms against 0.065 ms of HBM time. One activation buffer (halo rows only) + the two loaded weight
// planes of a chunk (the third is derived when a fragment is read): 80 KB, two blocks per CU; the chunk's weights are requested at
// staging time (registers), the next chunk's activations fly during the MFMA phase.
// ================================================================================================
template <int MODE, int KH, int KW>
__global__ __launch_bounds__(512, 4)
void conv_mfma_n32_kernel(const vps_conv_desc d, const int tiles_m) {
    constexpr int NTAP = KH * KW, NCB = 2;
    constexpr int HW = 32 + KW - 1, HH = 8 + KH - 1;
    constexpr int HROWS = HH * HW;
    constexpr int NLD = (HROWS + 63) / 64;
    typedef Split<MODE> SM;
    typedef typename SM::elem elem_t;
    typedef vec8<elem_t> x8;
    typedef vec4<elem_t> x4;
    constexpr int NSA = SM::NSA, NSB = SM::NSB, NLB = SM::NLB, NT = SM::NT;
    constexpr int PLANE = HROWS * LDS_LDH;
    constexpr int ABUF = NSA * PLANE;
    constexpr int NWL = NTAP * NLB * NCB;               // 1 KB weight fragments of one chunk: (tap, plane, column block)
    constexpr int WBUF = NWL * 512;
    constexpr int NBL = (NWL * 64 + 511) / 512;
    static_assert((ABUF + WBUF) * 2 <= 80 * 1024, "two blocks per CU");

    __shared__ __attribute__((aligned(16))) elem_t As[ABUF];
    __shared__ __attribute__((aligned(16))) elem_t Ws[WBUF];

    const int t = threadIdx.x;
    int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    int tile_n, tile_m, cls, split;
    decode_tile(d, swz, 1, tiles_m, tile_n, tile_m, cls, split);

    const int py = cls / d.os_x, px = cls - py * d.os_x;
    const int H = d.H, W = d.W, cin_pad = d.cin_pad;
    const int tiles_x = (d.Qw + 31) >> 5, tiles_y = (d.Qh + 7) >> 3;
    const int tx = tile_m % tiles_x, tq = tile_m / tiles_x;
    const int ty = tq % tiles_y, n = tq / tiles_y;
    const int iy_org = ty * 8 - d.pad_y[py], ix_org = tx * 32 - d.pad_x[px];

    const int k4 = t & 7;
    const int r0 = t >> 3;
    const int nchunks = d.kpad / (BK * NTAP);

    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // patch row
    const int nbt = d.cout_pad >> 5, kst = d.kpad >> 4;
    const size_t wplane = (size_t)d.nclass * nbt * kst * 512;
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(d.w_split, (unsigned)(wplane * NSB * sizeof(elem_t)));
    const unsigned wbase = (unsigned)(((size_t)(cls * nbt) * kst * 512) * sizeof(elem_t));
    // source of this lane's 16 bytes inside the two 32x16 fragments of a 32-k step, column block 0 (block 1: 16 lanes = 256 bytes further)
    const unsigned wlane = (unsigned)((((lane >> 5) & 1) * 512 + ((lane & 15) + 32 * ((lane >> 4) & 1)) * 8) * sizeof(elem_t));
    const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(d.in, (unsigned)((size_t)d.N * H * W * d.in_ld * sizeof(float)));
    const unsigned ld4 = (unsigned)d.in_ld * 4u;

    f32x4 areg[NLD];
    x8 wreg[NBL];
    int achunk = 0;
    float amax = 0.f;

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
        const unsigned coff = (unsigned)achunk * (BK * 4u);
        ++achunk;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            areg[i] = buffer_load16<f32x4>(arsrc, (kv && hoff[i] != 0xFFFFFFF0u) ? hoff[i] + coff : 0xFFFFFFF0u, 0u);
    };
    auto store_A = [&](int i) {
        x4 sp[NSA];
        split_act<MODE>(areg[i], sp, amax);
        const int row = r0 + 64 * i;
        if (row < HROWS) {
#pragma unroll
            for (int p = 0; p < NSA; ++p)
                *reinterpret_cast<x4*>(&As[p * PLANE + row * LDS_LDH + (((k4 >> 1) ^ lds_swz16(row)) << 3) + ((k4 & 1) << 2)]) = sp[p];
        }
    };
    // fragment f = (tap * NLB + plane) * NCB + column block of the chunk: wave-uniform per j (surplus waves re-load the last one)
    auto load_W = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NBL; ++j) {
            const int f = min(__builtin_amdgcn_readfirstlane((t + 512 * j) >> 6), NWL - 1);
            const int cb = f % NCB, tpl = f / NCB, pl = tpl % NLB, tap = tpl / NLB;
            wreg[j] = buffer_load16<x8>(wrsrc, wlane + (unsigned)cb * 256u,
                                        wbase + (unsigned)(((size_t)pl * wplane + (size_t)(2 * (chunk * NTAP + tap)) * 512) * sizeof(elem_t)));
        }
    };
    auto store_W = [&](int j) {
        if (t + 512 * j < NWL * 64) *reinterpret_cast<x8*>(&Ws[(t + 512 * j) * 8]) = wreg[j];        // (the last round is half empty: 36 fragments)
    };

    const int hbase = wave * HW + (lane & 15);
    const int kg = lane >> 4;
    f32x4 acc[NCB][2];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[c][h] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_A();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        load_W(chunk);
#pragma unroll
        for (int i = 0; i < NLD; ++i) store_A(i);
#pragma unroll
        for (int j = 0; j < NBL; ++j) store_W(j);
        if (chunk + 1 < nchunks) load_A();
        __syncthreads();
#pragma unroll
        for (int tp = 0; tp < NTAP; ++tp) {
            __builtin_amdgcn_sched_barrier(0);                    // (hoisting every tap's fragment reads to the top costs the second block per CU its registers)
            x8 af[2][NSA];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int hrow = hbase + 16 * h + (tp / KW) * HW + (tp % KW);
                const int off = hrow * LDS_LDH + ((kg ^ lds_swz16(hrow)) << 3);
#pragma unroll
                for (int p = 0; p < NSA; ++p) af[h][p] = *reinterpret_cast<const x8*>(&As[off + p * PLANE]);
            }
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                x8 wf[NSB];                                       // one column block at a time: 128 registers per lane, two blocks per CU
#pragma unroll
                for (int p = 0; p < NLB; ++p) wf[p] = *reinterpret_cast<const x8*>(&Ws[(((tp * NLB + p) * NCB + c) * 64 + lane) * 8]);
                if constexpr (NLB < NSB) wf[NLB] = derive_weight_plane<MODE>(wf[0]);
#pragma unroll
                for (int q = 0; q < NT; ++q)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        acc[c][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[SM::PB[q]], af[h][SM::PA[q]], acc[c][h], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    report_range<MODE>(d, amax);

    // epilogue: lane = pixel (lane & 15) of the group, output channels 16 c + 4 (lane >> 4) .. + 3
    const int qy = ty * 8 + wave;
    const bool v4 = !((d.out_ld | d.out_coff | d.cout) & 3) && !((uintptr_t)d.out & 15);
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const int co = 16 * c + 4 * kg;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (co + e < d.cout) {
                if (d.scale) sc[e] = d.scale[co + e];
                if (d.shift) sh[e] = d.shift[co + e];
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int qx = tx * 32 + 16 * h + (lane & 15);
            if (qy < d.Qh && qx < d.Qw && co < d.cout) {
                const int oy = qy * d.os_y + py, ox = qx * d.os_x + px;
                const size_t opix = ((size_t)n * d.Ho + oy) * d.Wo + ox;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[c][h][e] * sc[e] + sh[e];
                if (d.res) {
                    const int rs = d.res_shift;
                    const float* rp = d.res + (((size_t)n * (d.Ho >> rs) + (oy >> rs)) * (d.Wo >> rs) + (ox >> rs)) * d.res_ld + d.res_coff + co;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (co + e < d.cout) o[e] += rp[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = vps_act(o[e], d.act, d.slope);
                float* op = d.out + opix * d.out_ld + d.out_coff + co;
                if (v4) *reinterpret_cast<f32x4*>(op) = o;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (co + e < d.cout) op[e] = o[e];
                }
            }
        }
    }
}

}  // namespace

// 5..16 output channels, stride-1 3x3 / 2x2-class layers with enough 8 x 32 patches (launch_conv decides). VPS_N16T=0 switches the
// class-fused instance for transposed layers off (A/B; read per call)
__attribute__((visibility("hidden")))
void vpsi_launch_conv_n16(const vps_conv_desc& d, long tiles2d8, hipStream_t s) {
    const char* const e = getenv("VPS_N16T");
    const bool fused = !(e && atoi(e) == 0) && d.KH == 2 && d.KW == 2 && d.nclass == 4 && d.os_y == 2 && d.os_x == 2 &&
                       d.pad_y[0] >= 0 && d.pad_y[0] <= 1 && d.pad_y[1] >= 0 && d.pad_y[1] <= 1 &&
                       d.pad_x[0] >= 0 && d.pad_x[0] <= 1 && d.pad_x[1] >= 0 && d.pad_x[1] <= 1;
    if (fused && d.cout <= 16) {
        hipLaunchKernelGGL((conv_mfma_n16t_kernel<VPS_PREC_F16X3>), dim3((unsigned)tiles2d8), dim3(512), 0, s, d, (int)tiles2d8);
        return;
    }
    const long nblk8 = tiles2d8 * d.nclass;
    if (d.cout > 16) {                                   // 17 .. 32 output channels: two column blocks per wave
        if (d.KH == 3) hipLaunchKernelGGL((conv_mfma_n32_kernel<VPS_PREC_F16X3, 3, 3>), dim3((unsigned)nblk8), dim3(512), 0, s, d, (int)tiles2d8);
        else hipLaunchKernelGGL((conv_mfma_n32_kernel<VPS_PREC_F16X3, 2, 2>), dim3((unsigned)nblk8), dim3(512), 0, s, d, (int)tiles2d8);
        return;
    }
    if (d.KH == 3) hipLaunchKernelGGL((conv_mfma_n16_kernel<VPS_PREC_F16X3, 3, 3>), dim3((unsigned)nblk8), dim3(512), 0, s, d, (int)tiles2d8);
    else hipLaunchKernelGGL((conv_mfma_n16_kernel<VPS_PREC_F16X3, 2, 2>), dim3((unsigned)nblk8), dim3(512), 0, s, d, (int)tiles2d8);
}
