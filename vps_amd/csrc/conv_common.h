// Shared pieces of the MFMA convolution kernels (conv_mfma.hip, conv_q.hip): tile constants, XCD-aware tile order, the
// transposed-accumulator epilogue, buffer-addressed loads and the split-operand arithmetics. Everything lives in an
// anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "common.h"
#include <type_traits>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {


constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDS_LD = 36;  // floats per LDS row (32 + 4 pad): 16B aligned, conflict-free b128 reads

struct RowInfo {
    int iy0, ix0;   // top-left input coordinate of the receptive field (can be negative)
    int pixbase;    // n*H*W
    int moff;       // output pixel index (for the deformable offsets)
};

__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    // bijective remap so that XCD x (blocks bid%8==x) works on a contiguous chunk of tiles
    const int xcd = bid & 7;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// work-item order inside an XCD's contiguous chunk: column tile fastest; the parity classes of a transposed conv come next when the
// input outweighs the weights (the classes of one pixel tile read the same input patch: adjacent in the dispatch order, they share it
// through the XCD's L2 instead of streaming the input once per class), else the pixel tiles come next (class-major: an XCD then
// needs the weights of ~one class instead of all of them)
__device__ __forceinline__ void decode_tile(const vps_conv_desc& d, int swz, const int tiles_n, const int tiles_m, int& tile_n, int& tile_m, int& cls,
                                            int& split) {
    tile_n = swz % tiles_n; swz /= tiles_n;
    const bool cls_inner = d.nclass > 1 && (size_t)d.kpad * d.cout_pad * 8 < (size_t)d.N * d.H * d.W * d.cin_pad;
    if (cls_inner) {
        cls = swz % d.nclass; swz /= d.nclass;
        tile_m = swz % tiles_m;
        split = swz / tiles_m;
    } else {
        tile_m = swz % tiles_m; swz /= tiles_m;
        cls = swz % d.nclass;
        split = swz / d.nclass;
    }
}

// ---- shared epilogue.
// The kernels feed the WEIGHT fragment as the MFMA's A operand and the ACTIVATION fragment as its B operand (the two
// fragment layouts of the 32x32 shapes are mirror images, so this is an argument swap), i.e. every accumulator holds the
// TRANSPOSED 32x32 tile: C/D layout row = (r&3) + 8*(r>>2) + 4*(lane>>5) = output channel, col = lane&31 = pixel. A lane
// therefore owns ONE pixel per 32-row sub-tile and, per register group g = r>>2, FOUR CONSECUTIVE channels
//     co = tile_n*BN + wn*TN*32 + b*32 + 8*g + 4*(lane>>5) + (r&3)
// -> the output position / residual position is computed once per sub-tile (not once per register), and outputs, residuals,
// scale and shift move as float4 (16 B per lane: a quarter of the store instructions of the row-per-register layout; the
// epilogue of the short-K layers - 1x1 bottleneck convolutions - was store-issue bound).
// TILE2D: the block's 128 rows are an 8x16 patch of output positions (halo kernel), tile_m = (n*Qh/8 + ty)*Qw/16 + tx
// PWL = log2 of the patch width of a TILE2D block (8 x 16 patches of the 4-wave halo kernel, 8 x 32 of the 8-wave one)
// GN: the sums the GroupNorm behind this conv needs (vps_conv_desc.gn_stats) are taken from the values as they are stored
// EXTFLAG: the last-block flag of the split-K reduction lives in LDS the CALLER owns (`ext_flag`; a kernel that uses the whole LDS
// budget of its occupancy target cannot afford the 4 static bytes)
// PHL = log2 of the patch height of a TILE2D block (8 rows everywhere but in the 4-row instance of the thin-input kernel)
// RES = false: the caller guarantees d.res == NULL (the residual registers are not allocated)
template <int TM, int TN, int BN, bool TILE2D = false, int PWL = 4, bool GN = false, bool EXTFLAG = false, int PHL = 3, bool RES = true>
__device__ __forceinline__ void conv_epilogue(const vps_conv_desc& d, f32x16 (&acc)[TM][TN], const int M, const int tile_m,
                                              const int tile_n, const int cls, const int split, const int py, const int px,
                                              const int wm, const int wn, const int lane, const int tile_lin, int* ext_flag = nullptr) {
    const int prow = lane & 31;
    const int cq = 4 * (lane >> 5);
    const int cbase = tile_n * BN + wn * TN * 32 + cq;       // + b*32 + 8*g: first of this lane's 4 channels
    const float* const d_res = RES ? d.res : nullptr;

    // ---- one output position per sub-tile a. Rows past the end are clamped to a valid position and not stored.
    size_t opix[TM], rpix[TM];
    int mlin[TM];
    bool inside[TM];
    {
        int t2_n = 0, t2_y0 = 0, t2_x0 = 0;
        if constexpr (TILE2D) {
            const int tiles_x = (d.Qw + (1 << PWL) - 1) >> PWL, tiles_y = (d.Qh + (1 << PHL) - 1) >> PHL;
            const int tx = tile_m % tiles_x, tq = tile_m / tiles_x;
            t2_x0 = tx << PWL; t2_y0 = (tq % tiles_y) << PHL; t2_n = tq / tiles_y;
        }
        const bool simple_pix = !TILE2D && (d.nclass == 1 && d.os_y == 1 && d.os_x == 1 && d.res_shift == 0);
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int jl = wm * TM * 32 + a * 32 + prow;       // row of the block tile
            int qx, qy, n;
            if constexpr (TILE2D) {
                // patches overhang the right / bottom edge when Qw % 16 or Qh % 8: those rows are computed and dropped
                qx = t2_x0 + (jl & ((1 << PWL) - 1)); qy = t2_y0 + (jl >> PWL); n = t2_n;
                inside[a] = qx < d.Qw && qy < d.Qh;
                qx = min(qx, d.Qw - 1); qy = min(qy, d.Qh - 1);
                mlin[a] = (n * d.Qh + qy) * d.Qw + qx;
            } else {
                const int m_raw = tile_m * BM + jl;
                inside[a] = m_raw < M;
                mlin[a] = min(m_raw, M - 1);
                if (simple_pix) {
                    opix[a] = rpix[a] = (size_t)mlin[a];
                    continue;
                }
                qx = mlin[a] % d.Qw;
                const int tq = mlin[a] / d.Qw;
                qy = tq % d.Qh;
                n = tq / d.Qh;
            }
            const int oy = qy * d.os_y + py, ox = qx * d.os_x + px;
            opix[a] = ((size_t)n * d.Ho + oy) * d.Wo + ox;
            const int rs = d.res_shift;
            rpix[a] = ((size_t)n * (d.Ho >> rs) + (oy >> rs)) * (d.Wo >> rs) + (ox >> rs);
        }
    }

    if (d.ksplit > 1) {
        // partial sums [split][class][pixel][cout_pad]; pixel = linear (n, qy, qx) index whatever the tiling. cout_pad is a
        // multiple of 32 and the scratch buffer is 16-byte aligned: always float4
        const int Mpix = TILE2D ? d.N * d.Qh * d.Qw : M;
        const size_t splane = (size_t)d.nclass * Mpix * d.cout_pad;              // floats per split
        float* __restrict__ ws = d.ws + ((size_t)(split * d.nclass + cls) * Mpix) * d.cout_pad;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            if (!inside[a]) continue;
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(&ws[(size_t)mlin[a] * d.cout_pad + cbase + b * 32 + 8 * g]) = v;
                }
        }
        if (!d.tile_counter) return;             // reduced by conv_splitk_reduce_kernel
        // LAST-BLOCK reduction: the block that finishes a tile's last split adds the partial sums up (in split order 0, 1, ...:
        // the result does not depend on which block came last) and runs the epilogue itself - no reduce launch, the partials are
        // read while they are still in the cache hierarchy. Release / acquire at device scope around the tile's ticket counter.
        int* flag;
        if constexpr (EXTFLAG) flag = ext_flag;
        else {
            __shared__ int is_last;
            flag = &is_last;
        }
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            const int ticket = atomicAdd(&d.tile_counter[tile_lin], 1);
            *flag = ticket == d.ksplit - 1;
            if (*flag) d.tile_counter[tile_lin] = 0;        // ready for the next launch that uses this scratch buffer
        }
        __syncthreads();
        if (!*flag) return;
        __threadfence();
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        const float* __restrict__ w0 = d.ws + ((size_t)cls * Mpix) * d.cout_pad;
        for (int sp = 0; sp < d.ksplit; ++sp) {
            f32x4 v[TM][TN][4];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        v[a][b][g] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(&w0[(size_t)sp * splane + (size_t)mlin[a] * d.cout_pad + cbase + b * 32 + 8 * g]));
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[a][b][4 * g + e] += v[a][b][g][e];
        }
        // fall through: the regular epilogue on the summed accumulators
    }

    const bool vec = !((d.cout | d.out_ld | d.out_coff) & 3) && !((uintptr_t)d.out & 15) &&
                     (!d_res || (!((d.res_ld | d.res_coff) & 3) && !((uintptr_t)d_res & 15)));
    if (vec) {
        // residual: all TM*TN*4 float4 values are requested (branch-free, clamped) before the first one is used. A load that
        // sits behind `if (row valid) if (column valid)` gets an s_waitcnt vmcnt(0) of its own.
        f32x4 rv[TM][TN][4];
        if (d_res) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cc = min(cbase + b * 32 + 8 * g, d.cout - 4);
                        rv[a][b][g] = *reinterpret_cast<const f32x4*>(d_res + rpix[a] * d.res_ld + d.res_coff + cc);
                    }
        }
        // scale / shift of ALL channel groups are requested here too, before the first store. Loaded group by group between the stores
        // (as this loop did until round 4) every group's `s_waitcnt vmcnt(0)` for its two loads also waited for the stores of the group
        // before it - memory operations retire in order - i.e. 4 * TN serialised store round trips per tile: the thin-input layers
        // spent 15 of 17 us per tile there (found in the ISA of conv_thin.hip; tools/storebench: the same store pattern alone runs at 4.9 TB/s)
        f32x4 scv[TN][4], shv[TN][4];
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cc = min(cbase + b * 32 + 8 * g, d.cout - 4);
                scv[b][g] = f32x4{1.f, 1.f, 1.f, 1.f}; shv[b][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (d.scale) scv[b][g] = *reinterpret_cast<const f32x4*>(d.scale + cc);
                if (d.shift) shv[b][g] = *reinterpret_cast<const f32x4*>(d.shift + cc);
            }
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = cbase + b * 32 + 8 * g;
                const bool cok = co < d.cout;                    // cout % 4 == 0: a group is valid or invalid as a whole
                const f32x4 sc = scv[b][g], sh = shv[b][g];
                float gs = 0.f, gq = 0.f;
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    if (!inside[a] || !cok) continue;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[a][b][4 * g + e] * sc[e] + sh[e];
                        if (d_res) t += rv[a][b][g][e];
                        v[e] = vps_act(t, d.act, d.slope);
                        if constexpr (GN) { gs += v[e]; gq += v[e] * v[e]; }
                    }
                    *reinterpret_cast<f32x4*>(d.out + opix[a] * d.out_ld + d.out_coff + co) = v;
                }
                if constexpr (GN) {
                    // this lane's <= TM * 4 values belong to ONE group (4 | gn_cpg): channels co .. co+3 of group co / gn_cpg. The 32 lanes
                    // of a half wave hold the same channels of 32 positions; with gn_cpg == 8 the other half holds the group's other
                    // four channels. Lane sums in fp32 (8 values), everything above in double, one atomic pair per wave (and half).
                    if (d.gn_stats) {
                        double ds = (double)gs, dq = (double)gq;
#pragma unroll
                        for (int off = 1; off < 32; off <<= 1) { ds += __shfl_xor(ds, off, 64); dq += __shfl_xor(dq, off, 64); }
                        if (d.gn_cpg >= 8) { ds += __shfl_xor(ds, 32, 64); dq += __shfl_xor(dq, 32, 64); }
                        const bool writer = d.gn_cpg >= 8 ? lane == 0 : (lane & 31) == 0;
                        if (writer && cok) {
                            // gn_rep copies of the 2 G sums, chosen by block: a thousand blocks on 4 cache lines would queue up
                            double* __restrict__ st = d.gn_stats + (size_t)(blockIdx.x & (d.gn_rep - 1)) * 2 * (d.cout / d.gn_cpg);
                            const int grp = co / d.gn_cpg;
                            atomicAdd(&st[2 * grp], ds);
                            atomicAdd(&st[2 * grp + 1], dq);
                        }
                    }
                }
            }
        return;
    }
    // scalar path: a channel count / window that is not a multiple of 4 (19-, 18-, 9-channel heads, odd concat offsets)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = cbase + b * 32 + 8 * g + e;
                const bool cok = co < d.cout;
                const int cc = min(co, d.cout - 1);
                const float sc = d.scale ? d.scale[cc] : 1.f;
                const float sh = d.shift ? d.shift[cc] : 0.f;
                float rvs[TM];
#pragma unroll
                for (int a = 0; a < TM; ++a) rvs[a] = d_res ? d_res[rpix[a] * d.res_ld + d.res_coff + cc] : 0.f;
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    if (!inside[a] || !cok) continue;
                    const float t = acc[a][b][4 * g + e] * sc + sh + rvs[a];
                    d.out[opix[a] * d.out_ld + d.out_coff + co] = vps_act(t, d.act, d.slope);
                }
            }
}

// ================================================================================================
// Split-bf16 variant: the same implicit GEMM on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA
// rate) with fp32 operands decomposed into NS bf16 terms (x = x0 + x1 (+ x2), each term the bf16 RNE of the
// remaining residual) and fp32 accumulation:
//   NS = 2: products x0*w0 + x0*w1 + x1*w0          (3 MFMAs, relative error ~2^-16 per product, "bf16x3")
//   NS = 3: x0w0 + x0w1 + x1w0 + x1w1 + x0w2 + x2w0 (6 MFMAs, relative error ~2^-23: fp32-grade,  "bf16x6")
// Weights are split once on the host (planes [NS][class][cout_pad][kpad] bf16); activations stay fp32 in HBM and are
// split by the thread that stages them into LDS (once per block, not once per consuming wave). LDS rows are
// [row][32 k] bf16 (64 bytes) with XOR-swizzled 16-byte chunks (conflict-free staging writes and fragment reads).
// ================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <class T> using vec8 = T __attribute__((ext_vector_type(8)));
template <class T> using vec4 = T __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Buffer-addressed 16-byte loads (buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen): the 128-bit resource and the scalar offset
// live in SGPRs, the lane contributes a 32-bit byte offset - no 64-bit VALU address arithmetic per load - and a lane whose offset
// is >= the buffer's byte count reads ZEROS: out-of-image taps and channel pads are masked by the address (one select on the
// offset) instead of a select per loaded element. Tensors are < 4 GiB (checked by vps_conv2d).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, const unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
template <class V>
__device__ __forceinline__ V buffer_load16(const __amdgpu_buffer_rsrc_t r, const unsigned voff, const unsigned soff) {
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
constexpr int LDS_LDH = 32;   // 16-bit elements per LDS row = 64 bytes, no padding: the four 16-byte chunks of a row are
                              // XOR-swizzled with (row>>2)&3, which makes the 8-byte/16-byte staging writes of two consecutive
                              // rows cover all 32 banks once and the 16-lane groups of the fragment ds_read_b128 hit 16
                              // distinct 16-byte slots (the padded 80-byte layout measured 33 % conflict cycles)
__device__ __forceinline__ int lds_swz(int row) { return (row >> 2) & 3; }

// The split arithmetics (vps_conv_desc.prec). NSA activation planes (staged in LDS), NSB weight planes (packed on the host),
// NT products per k-slab: term q multiplies activation plane PA[q] with weight plane PB[q], smallest magnitude first.
template <int MODE> struct Split;
// NLB: weight planes that are LOADED (the first NLB of the NSB packed ones); the rest is derived in registers (derive_weight_plane)
template <> struct Split<VPS_PREC_BF16> {
    typedef __bf16 elem;
    static constexpr int NSA = 1, NSB = 1, NT = 1, NLB = 1;
    static constexpr int PA[6] = {0, 0, 0, 0, 0, 0};
    static constexpr int PB[6] = {0, 0, 0, 0, 0, 0};
};
template <> struct Split<VPS_PREC_BF16X3> {
    typedef __bf16 elem;
    static constexpr int NSA = 2, NSB = 2, NT = 3, NLB = 2;
    static constexpr int PA[6] = {1, 0, 0, 0, 0, 0};
    static constexpr int PB[6] = {0, 1, 0, 0, 0, 0};
};
template <> struct Split<VPS_PREC_BF16X6> {
    typedef __bf16 elem;
    static constexpr int NSA = 3, NSB = 3, NT = 6, NLB = 3;
    static constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    static constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
};
// fp16 with a scaled residual: x = h0 + 2^-11 h1 (planes 0, 1); weights g0, g1, g2 = 2^-11 g0 (planes 0, 1, 2):
// x*w ~ h0*g1 + h1*g2 + h0*g0. See VPS_PREC_F16X3 in vps_hip.h for the error / range statement.
template <> struct Split<VPS_PREC_F16X3> {
    typedef _Float16 elem;
    static constexpr int NSA = 2, NSB = 3, NT = 3, NLB = 2;
    static constexpr int PA[6] = {0, 1, 0, 0, 0, 0};
    static constexpr int PB[6] = {1, 2, 0, 0, 0, 0};
};

// f16x3: weight plane 2 is 2^-11 * plane 0 (an exact scaling, rounded to the fp16 subnormals like the host's packing does: fp16
// denormals are on in the kernels' float mode), so it is not loaded - a third less weight traffic on the L1 side of every kernel,
// which is what the HBM-bound and the tap-by-tap weight-streaming layers are short of - but made from plane 0 by four packed fp16
// multiplies per fragment. The host still packs all three planes (kernels that are not converted read them).
template <int MODE>
__device__ __forceinline__ vec8<typename Split<MODE>::elem> derive_weight_plane(const vec8<typename Split<MODE>::elem> g0) {
    static_assert(MODE == VPS_PREC_F16X3, "only the fp16 mode has a derived weight plane");
    typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
    const _Float16 s = (_Float16)0.00048828125f;          // 2^-11
    const h16x8 k = {s, s, s, s, s, s, s, s};
    return g0 * k;
}

template <int MODE>
__device__ __forceinline__ f32x16 split_mfma(const vec8<typename Split<MODE>::elem> a, const vec8<typename Split<MODE>::elem> b, const f32x16 c) {
    if constexpr (MODE == VPS_PREC_F16X3) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// fp32 float4 -> NSA planes of 4 elements. `amax` tracks max |x| of the staged values in the fp16 mode (range report).
template <int MODE>
__device__ __forceinline__ void split_act(const f32x4 v, vec4<typename Split<MODE>::elem> (&out)[Split<MODE>::NSA], float& amax) {
#ifdef VPS_KO_NOSPLIT
    // KNOCK-OUT build (tools/build_knockout.sh, never the product): the staged float4 is reinterpreted as the two fp16 planes of four
    // values (0 VALU, no range report) - what a loader would cost if the PRODUCER had written the pair. Results are garbage; the
    // build only bounds what pre-split activations could buy.
    if constexpr (MODE == VPS_PREC_F16X3) {
        typedef vec4<_Float16> h4;
        struct two { h4 a, b; };
        const two p = __builtin_bit_cast(two, v);
        out[0] = p.a; out[1] = p.b;
        return;
    }
#endif
    if constexpr (MODE == VPS_PREC_F16X3) {
        // per PAIR of elements: one packed RNE conversion for h0 (v_cvt_pk_f16_f32), the exact residual x - h0 as one mixed-precision
        // FMA reading the fp16 half directly (v_fma_mix_f32: no convert-back, no separate subtract; the compiler does not form it), the exact scaling by 2^11, one
        // packed conversion for h1, one three-operand maximum for the range report: 3.5 instead of ~6 VALU instructions per staged
        // element. Same values bit for bit as h0 = fp16(x), h1 = fp16((x - h0) * 2^11).
        typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            const f32x2 x = {v[e], v[e + 1]};
            const h16x2 h0 = __builtin_convertvector(x, h16x2);
            // (x - h0) * 2^11 as ONE mixed-precision FMA per element on top of the scaling multiply: fma(h0, -2^11, x * 2^11) is exact
            // (the difference is representable) and, with a multiplier other than -1, the compiler selects v_fma_mix_f32 reading the
            // fp16 half in place (with -1 it folds the FMA into convert + subtract; inline asm would make the staging loops convergent)
            const f32x2 r = {__builtin_fmaf((float)h0[0], -2048.f, x[0] * 2048.f), __builtin_fmaf((float)h0[1], -2048.f, x[1] * 2048.f)};
            const h16x2 h1 = __builtin_convertvector(r, h16x2);
            out[0][e] = h0[0]; out[0][e + 1] = h0[1];
            out[1][e] = h1[0]; out[1][e + 1] = h1[1];
            amax = fmaxf(fmaxf(amax, fabsf(x[0])), fabsf(x[1]));
        }
    } else {
        f32x4 r = v;
#pragma unroll
        for (int p = 0; p < Split<MODE>::NSA; ++p) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const __bf16 h = (__bf16)r[e];
                out[p][e] = h;
                r[e] -= (float)h;
            }
        }
    }
}

// fp16 mode: an activation beyond the fp16 range was staged by this thread -> OR bit 0 into the caller's status word
template <int MODE>
__device__ __forceinline__ void report_range(const vps_conv_desc& d, const float amax) {
    if constexpr (MODE == VPS_PREC_F16X3) {
        if (d.status && !(amax <= 65504.f)) atomicOr(d.status, 1);
    }
}

}  // namespace
