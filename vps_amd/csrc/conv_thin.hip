// Thin-input convolutions on the fp16 matrix cores (f16x3 arithmetic): the first layers of the FlowNet2 sub-networks and of the
// ResNet stem - 3 / 6 / 11 / 12 input channels, 64 output channels, 3x3 stride 1 or 7x7 stride 2 at full resolution
// (mmdet/models/flow_modules/FlowNetS.py:20 conv1 12->64 7x7 s2, FlowNetC.py:20 conv1 3->64 7x7 s2, FlowNetSD.py:16 conv0 6->64 3x3,
// FlowNetFusion.py:16 conv0 11->64 3x3, mmdet/models/backbones/resnet.py:454 conv1 3->64 7x7 s2).
//
// On the pipelined kernels these layers are 3..19 k-steps of a tap-major gather per 128-pixel tile: every k-step re-loads, re-splits and
// re-stages 32 (tap, channel) values per pixel, and a tile is one chain of memory latencies (0.3 ms for 0.1 ms of HBM traffic).
// Here the k order is made for the hardware instead: with the channels of a pixel padded to C4 = 4 / 8 / 12, the KS taps of one kernel
// ROW are KS * C4 CONSECUTIVE values of the NHWC input, so
//   * the block stages the input patch of its PH x 32 output positions ONCE (split into the two fp16 planes), pixel-major, in LDS;
//   * k runs over (kernel row j, position within the row's KS * C4 values, padded to a multiple of 16): the MFMA fragment of a lane
//     - 8 consecutive k of one output position - is 16 contiguous bytes of the staged patch at (row + j, column * stride), whatever
//     taps / channels they are. No per-tap gather, no per-k-step staging; the padding k of a row read the next pixels' (finite)
//     values against zero weights;
//   * the weights (both loaded planes, fragment order) stay in LDS for the life of a persistent block (24..56 KB; the 12-channel 7x7
//     layer streams them per kernel row, the next row in flight in registers); plane 2 = 2^-11 * plane 0 is derived in registers.
// The weights come in their own packing (vps_conv_desc.w_thin, nhwc.py).
#include "conv_common.h"

namespace {

constexpr int TMODE = VPS_PREC_F16X3;
typedef _Float16 h16;
typedef vec8<h16> h16x8;
typedef vec4<h16> h16x4;

// C4: channels per staged pixel (cin_pad); KS: kernel size; S: stride; PH: output rows per block (4 waves x PH/4 rows x 32 columns)
// WRES: the weights of ALL kernel rows stay in LDS (loaded once per block); else one row at a time, the next one in flight in registers.
//
// PERSISTENT blocks with nothing in the tile loop that waits for a store. Memory operations retire in order (one vmcnt queue for
// loads and stores), so any load a tile waits for drags every older store with it; a 64-channel tile at full resolution is 64 KB of
// stores and a store round trip under load is ~6 us (tools/storebench: 32 MB in flight at 5 TB/s), against ~2 us of work per tile.
// First versions, measured on 6->64 @1024x2048 (0.300 ms on the pipelined kernel; the same 537 MB written by a bare store loop with
// the epilogue's access pattern: 0.108 ms, tools/storebench): one tile per block, weights streamed per kernel row 0.265 ms (an L2 round
// trip per row); persistent with resident weights and the shared epilogue 0.236 ms - stores alone 0.209 ms of it (knock-outs): that
// epilogue loads scale / shift per tile (a vmcnt(0) behind the previous tile's stores) and its stores sit behind `if (inside)` branches,
// which leaves the compiler no lower bound for the stores in flight, so the next tile's staging waited with vmcnt(0) as well. Here:
//   * the patch of tile t + 2 is requested before the MFMAs of tile t (two register sets, tile loop unrolled by two; t + 1 with one
//     set in the streamed-weight instance, whose registers also hold a kernel row of weights in flight);
//   * scale / shift live in LDS (read with lgkmcnt, not vmcnt);
//   * stores are unconditional buffer stores - a lane outside the map / beyond cout stores to an offset past the buffer, which the
//     hardware drops - so the staging wait is an exact vmcnt(2 tiles of stores + one patch) and two tiles of stores stay in flight.
template <int C4, int KS, int S, int PH, int OCC, bool WRES>
__global__ __launch_bounds__(256, OCC)
void conv_thin_kernel(const vps_conv_desc d, const int tiles_x, const int tiles_y, const int ntiles) {
    typedef Split<TMODE> SM;
    constexpr int TM = PH / 4;
    constexpr int PR = (PH - 1) * S + KS, PC = 31 * S + KS, NPOS = PR * PC;
    constexpr int RL = KS * C4, NK16 = (RL + 15) / 16;
    constexpr int NV = C4 / 4;                                   // float4 per staged position
    constexpr int NLOAD = (NPOS * NV + 255) / 256;
    constexpr int APLANE = NPOS * C4 + 16;                       // + slack: the last positions' padding k read past the patch
    constexpr int WROW = 2 * NK16 * 2 * 512;                     // halfs of one kernel row's weights: [plane][k16][n block][lane][8]
    constexpr int WBUF = (WRES ? KS : 1) * WROW;
    constexpr int JU = KS <= 3 ? KS : 1;                         // the 7 kernel rows of the stems stay a loop (registers)
    static_assert((2 * APLANE) % 8 == 0, "the weight buffer behind the two activation planes stays 16-byte aligned");
    extern __shared__ __attribute__((aligned(16))) unsigned char thin_smem[];       // (2 * APLANE + WBUF) halfs + 128 floats
    h16* const As = reinterpret_cast<h16*>(thin_smem);
    h16* const Ws = As + 2 * APLANE;
    float* const scsh = reinterpret_cast<float*>(Ws + WBUF);      // [scale 64][shift 64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p32 = lane & 31, kh = lane >> 5;
    const int G = gridDim.x;
    const int bsw = xcd_swizzle(blockIdx.x, G);                   // an XCD's blocks walk neighbouring tiles (they share halo columns)
    const __amdgpu_buffer_rsrc_t in_rsrc = make_rsrc(d.in, (unsigned)((size_t)d.N * d.H * d.W * d.in_ld * sizeof(float)));
    const __amdgpu_buffer_rsrc_t out_rsrc = make_rsrc(d.out, (unsigned)((size_t)d.N * d.Ho * d.Wo * d.out_ld * sizeof(float)));
    const u32x4* __restrict__ wg = reinterpret_cast<const u32x4*>(d.w_thin);

    // all loads of a tile's patch are requested at once; a tile index past the end requests nothing (offsets beyond the buffer return 0)
    auto issue_patch = [&](const int tile, f32x4 (&v)[NLOAD]) {
        // the thread index is made opaque per call: the ~10 index values per load below depend on nothing that changes from tile to tile,
        // and hoisted out of the tile loop they cost 300 spilled registers
        int t_ = threadIdx.x;
        asm volatile("" : "+v"(t_));
        const int tx = tile % tiles_x, tq = tile / tiles_x;
        const int ty = tq % tiles_y, n = tq / tiles_y;
        const int iy0 = ty * PH * S - d.pad_y[0], ix0 = tx * 32 * S - d.pad_x[0];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int idx = t_ + i * 256;
            const int pos = idx / NV, q = idx - pos * NV;
            const int r = pos / PC, c = pos - r * PC;
            const int iy = iy0 + r, ix = ix0 + c;
            const bool ok = tile < ntiles && idx < NPOS * NV && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
            const unsigned off = (unsigned)((((size_t)n * d.H + iy) * d.W + ix) * d.in_ld + d.in_coff + 4 * q) * 4u;
            v[i] = buffer_load16<f32x4>(in_rsrc, ok ? off : 0xFFFFFFF0u, 0);
        }
    };
    float amax = 0.f;
    auto stage_patch = [&](const f32x4 (&v)[NLOAD]) {
        int t_ = threadIdx.x;
        asm volatile("" : "+v"(t_));
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int idx = t_ + i * 256;
            h16x4 pl[SM::NSA];
            split_act<TMODE>(v[i], pl, amax);
            if (idx < NPOS * NV) {
#pragma unroll
                for (int p = 0; p < 2; ++p) *reinterpret_cast<h16x4*>(&As[p * APLANE + idx * 4]) = pl[p];
            }
        }
    };

    // LEAD = 2 where the registers allow it (the streamed-weight instance keeps a row of weights in flight too: one tile ahead)
    constexpr int LEAD = WRES ? 2 : 1;
    f32x4 v0[NLOAD], v1[LEAD == 2 ? NLOAD : 1];
    issue_patch(bsw, v0);
    if constexpr (LEAD == 2) issue_patch(bsw + G, v1);
    u32x4 wreg[NK16];
    if constexpr (WRES) {
#pragma unroll
        for (int j = 0; j < KS; ++j) {
#pragma unroll
            for (int u = 0; u < NK16; ++u) wreg[u] = wg[j * (WROW / 8) + tid + u * 256];
#pragma unroll
            for (int u = 0; u < NK16; ++u) reinterpret_cast<u32x4*>(Ws)[j * (WROW / 8) + tid + u * 256] = wreg[u];
        }
    } else {
#pragma unroll
        for (int u = 0; u < NK16; ++u) wreg[u] = wg[tid + u * 256];
#pragma unroll
        for (int u = 0; u < NK16; ++u) reinterpret_cast<u32x4*>(Ws)[tid + u * 256] = wreg[u];
    }
    if (tid < 8) {
        const h16x4 z = {(h16)0, (h16)0, (h16)0, (h16)0};
        *reinterpret_cast<h16x4*>(&As[(tid >> 2) * APLANE + NPOS * C4 + (tid & 3) * 4]) = z;
    }
    if (tid < 128) {
        const int c = tid & 63;
        float val = tid < 64 ? 1.f : 0.f;
        if (c < d.cout) {
            if (tid < 64 && d.scale) val = d.scale[c];
            if (tid >= 64 && d.shift) val = d.shift[c];
        }
        scsh[tid] = val;
    }

    // one tile: stage its patch, request the patch two tiles ahead into the registers just freed, MFMAs, stores
    auto body = [&](const int tile, f32x4 (&v)[NLOAD]) {
        stage_patch(v);                                           // As is free: the barrier behind the previous tile's MFMAs
        __syncthreads();
        issue_patch(tile + LEAD * G, v);

        f32x16 acc[TM][2];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll JU
        for (int j = 0; j < KS; ++j) {
            if constexpr (!WRES) {
                // the next row's weights - after the last row, row 0 for the next tile
#pragma unroll
                for (int u = 0; u < NK16; ++u) wreg[u] = wg[((j + 1) % KS) * (WROW / 8) + tid + u * 256];
            }
            const h16* const Wj = Ws + (WRES ? j * WROW : 0);
#pragma unroll
            for (int s = 0; s < NK16; ++s) {
                h16x8 af[2][TM], wf[3][2];
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int base = (((wave * TM + a) * S + j) * PC + p32 * S) * C4 + s * 16 + kh * 8;
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const h16x4 lo = *reinterpret_cast<const h16x4*>(&As[p * APLANE + base]);
                        const h16x4 hi = *reinterpret_cast<const h16x4*>(&As[p * APLANE + base + 4]);
                        af[p][a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) wf[p][b] = *reinterpret_cast<const h16x8*>(&Wj[((p * NK16 + s) * 2 + b) * 512 + lane * 8]);
                    wf[2][b] = derive_weight_plane<TMODE>(wf[0][b]);
                }
#pragma unroll
                for (int q = 0; q < SM::NT; ++q)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc[a][b] = split_mfma<TMODE>(wf[SM::PB[q]][b], af[SM::PA[q]][a], acc[a][b]);
            }
            if constexpr (!WRES) {
                __syncthreads();                                  // every wave is done with row j's weights (and, after the last row, with As)
#pragma unroll
                for (int u = 0; u < NK16; ++u) reinterpret_cast<u32x4*>(Ws)[tid + u * 256] = wreg[u];
                if (j + 1 < KS) __syncthreads();                  // after the last row the barrier behind the next tile's staging does it
            }
        }
        if constexpr (WRES) __syncthreads();                      // every wave is done with As

        // ---- epilogue on the transposed accumulators (conv_common.h): a lane owns pixel p32 of its TM rows and channels
        // 32 b + 8 g + 4 kh .. + 3; y = act(acc * scale + shift), float4 buffer stores, masked by the offset
        int l_ = threadIdx.x;
        asm volatile("" : "+v"(l_));
        const int prow = l_ & 31, cq = 4 * ((l_ >> 5) & 1), wv = l_ >> 6;
        const int tx = tile % tiles_x, tq = tile / tiles_x;
        const int ty = tq % tiles_y, n = tq / tiles_y;
        const int ox = tx * 32 + prow;
        // the activation without a branch per element: y = max(t, 0) + ns * min(t, 0), ns = 0 (ReLU) | slope (leaky) | 1 (none) - the same values
        const float ns = d.act == VPS_ACT_RELU ? 0.f : (d.act == VPS_ACT_LEAKY ? d.slope : 1.f);
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int oy = ty * PH + wv * TM + a;
            const bool inside = tile < ntiles && oy < d.Qh && ox < d.Qw;
            const unsigned pixoff = (unsigned)((((size_t)n * d.Ho + oy) * d.Wo + ox) * d.out_ld + d.out_coff + cq) * 4u;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = b * 32 + 8 * g + cq;
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(&scsh[co]);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(&scsh[64 + co]);
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = acc[a][b][4 * g + e] * sc[e] + sh[e];
                        o[e] = fmaxf(t, 0.f) + ns * fminf(t, 0.f);
                    }
                    const unsigned off = inside && co < d.cout ? pixoff + (unsigned)(b * 32 + 8 * g) * 4u : 0xFFFFFFF0u;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), out_rsrc, (int)off, 0, 0);
                }
        }
    };
    if constexpr (LEAD == 2) {
        // two tiles per trip: the register sets are static. A trailing odd tile runs as a masked tile (no loads, no stores)
        for (int tile = bsw; tile < ntiles; tile += 2 * G) {
            body(tile, v0);
            body(tile + G, v1);
        }
    } else {
        for (int tile = bsw; tile < ntiles; tile += G) body(tile, v0);
    }
    report_range<TMODE>(d, amax);
}

template <int C4, int KS, int S, int PH, int OCC, bool WRES>
int launch_thin(const vps_conv_desc& d, hipStream_t s) {
    const int tiles_x = (d.Qw + 31) / 32, tiles_y = (d.Qh + PH - 1) / PH;
    const long ntiles = (long)d.N * tiles_x * tiles_y;
    if (ntiles < 512 || ntiles > 0x7fffffffL) return 0;          // small maps: the pipelined kernels (split-K, fewer idle lanes)
    constexpr int PR = (PH - 1) * S + KS, PC = 31 * S + KS, NK16 = (KS * C4 + 15) / 16;
    constexpr size_t smem = (size_t)(2 * (PR * PC * C4 + 16) + (WRES ? KS : 1) * 2 * NK16 * 2 * 512) * 2 + 128 * sizeof(float);
    static int resident = 0;                                      // blocks the chip holds at once
    if (!resident) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_thin_kernel<C4, KS, S, PH, OCC, WRES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        int per_cu = 0, dev = 0, cus = 256;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_thin_kernel<C4, KS, S, PH, OCC, WRES>, 256, smem);
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        resident = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 256);
        if (getenv("VPS_DEBUG_OCC")) fprintf(stderr, "[vps] thin-input kernel <%d,%d,%d,%d>: %d blocks/CU, %zu B LDS\n", C4, KS, S, PH, per_cu, smem);
    }
    // whole rounds: every block walks the same number of tiles (+-1)
    const long rounds = (ntiles + resident - 1) / resident;
    const long grid = (ntiles + rounds - 1) / rounds;
    hipLaunchKernelGGL((conv_thin_kernel<C4, KS, S, PH, OCC, WRES>), dim3((unsigned)grid), dim3(256), smem, s, d, tiles_x, tiles_y, (int)ntiles);
    return 1;
}

}  // namespace

// -> 1 if a thin-input instance exists for this launch and was enqueued, 0 if the caller has to use another kernel.
// VPS_THIN=0 in the environment switches the family off (A/B runs).
__attribute__((visibility("hidden")))
int vpsi_launch_conv_thin(const vps_conv_desc& d, hipStream_t s) {
    static const int on = getenv("VPS_THIN") ? atoi(getenv("VPS_THIN")) : 1;
    if (!on || !d.w_thin || d.prec != VPS_PREC_F16X3 || d.offset || d.res || d.nclass != 1 || d.ksplit != 1 || d.korder != 0) return 0;
    if (d.KH != d.KW || d.cout_pad != 64 || d.tile_n != 64 || (d.cout & 3) || ((uintptr_t)d.w_thin & 15)) return 0;
    if (d.pad_y[0] != d.KH / 2 || d.pad_x[0] != d.KH / 2 || d.pad_y[1] != d.KH / 2 || d.pad_x[1] != d.KH / 2) return 0;
    // the input is addressed through a 32-bit buffer resource: a narrow window inside a buffer of >= 4 GiB would wrap (loads beyond the
    // truncated size return zeros, silently) -> the pipelined kernel takes such a launch
    if ((size_t)d.N * d.H * d.W * d.in_ld * sizeof(float) >= 0xFFFFFFF0ull) return 0;
    // float4 buffer stores: 16-byte aligned channel windows of an output below 4 GiB
    if (((d.out_ld | d.out_coff) & 3) || ((uintptr_t)d.out & 15) || (size_t)d.N * d.Ho * d.Wo * d.out_ld * sizeof(float) >= 0xFFFFFFF0ull) return 0;
    if (d.KH == 3 && d.stride == 1 && d.cin_pad == 8) return launch_thin<8, 3, 1, 8, 2, true>(d, s);
    if (d.KH == 3 && d.stride == 1 && d.cin_pad == 12) return launch_thin<12, 3, 1, 8, 2, true>(d, s);
    if (d.KH == 7 && d.stride == 2 && d.cin_pad == 4) return launch_thin<4, 7, 2, 8, 2, true>(d, s);
    if (d.KH == 7 && d.stride == 2 && d.cin_pad == 12) return launch_thin<12, 7, 2, 4, 2, false>(d, s);
    return 0;
}
