// Persistent pointwise (1x1) convolution on the fp16 matrix cores (f16x3 arithmetic), round 6: the ResNet bottleneck 1x1 layers
// (mmdet/models/backbones/resnet.py:220-266 conv1 / conv3 / downsample), the FPN laterals (necks/fpn.py) and the 1x1 layers of the
// TCEA fusion (extra_necks/tcea_modules.py) on the large maps.
//
// On the uniform-lead kernel (conv_q.hip) such a layer is a grid of short-lived blocks: 2..16 k-steps of matrix work (0.7..5 us)
// inside a chain of latencies that nothing overlaps - kernel arguments, the first loads (2..3 us under load), the residual / scale /
// shift loads of the epilogue (another memory latency), 64 KB of stores that have to drain before the block retires (~6 us under
// load, tools/storebench) and the dispatch of the next block: `64->256 @256x512` spends 21 us per block for 96 KB of traffic.
// Here a block is PERSISTENT and the (tile, k-step) sequence is ONE pipeline:
//   * the loads of both operands run three k-steps ahead ACROSS tile boundaries (two register slots per operand, staged through
//     LDS like in conv_q.hip): the first steps of tile t+1 are requested before the epilogue of tile t;
//   * the stores of the epilogue are unconditional buffer stores (a lane outside the map / beyond cout stores past the end of the
//     buffer, which the hardware drops): the compiler can count them, so the waits of the next tile's first two k-steps are exact
//     `vmcnt`s that leave the stores in flight (memory operations retire in order: a wait for anything requested AFTER the stores
//     drains them). The tile loop is peeled accordingly - k-steps 0 and 1 of a tile are straight-line code behind the epilogue;
//   * the kernel's prologue issues the same number of (dropped) stores, so that the loop header merges two equal wait states;
//   * scale / shift of the block's column tile live in LDS (a block keeps its column tile: its weights stay in the XCD's L2);
//   * accumulation order per output element = the uniform-lead kernel's: results are bitwise equal (tests/test_hip_ops.py).
#include "conv_common.h"
#include <cstdio>

namespace {

constexpr int PMODE = VPS_PREC_F16X3;

template <int TN, bool HAS_RES, int NKU>
__global__ __launch_bounds__(256, 2)
void conv_pw_kernel(const vps_conv_desc d, const int M, const int tiles_m, const int tiles_n, const int nk, const int nit) {
    constexpr int TM = 2, WAVES_N = 2;
    constexpr int BN = WAVES_N * TN * 32;
    typedef Split<PMODE> SM;
    typedef _Float16 elem_t;
    typedef vec8<elem_t> x8;
    typedef vec4<elem_t> x4;
    constexpr int NSA = SM::NSA, NSB = SM::NSB, NLB = SM::NLB, NT = SM::NT;
    static_assert(NSA == 2 && NLB == 2 && NSB == 3 && NT == 3, "f16x3");
    constexpr int ABUF = NSA * BM * LDS_LDH;     // halfs of one activation buffer (both planes)
    constexpr int NFRAG = NLB * 2 * (BN / 32);   // 1 KB weight fragments of one k-step of the block tile: (plane, slab, column block)
    constexpr int BBUF = NFRAG * 512;
    static_assert((NFRAG * 64) % 256 == 0, "whole 16-byte chunks per thread");
    constexpr int NBL = NFRAG * 64 / 256;        // 16-byte weight chunks per thread and k-step
    constexpr int NST = TM * TN * 4;             // float4 stores per thread and tile

    __shared__ __attribute__((aligned(16))) elem_t As[2 * ABUF];
    __shared__ __attribute__((aligned(16))) elem_t Bs[2 * BBUF];
    __shared__ __attribute__((aligned(16))) float scsh[2 * BN];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;

    // ---- the block's tiles: a fixed column tile; pixel tiles (it * mper + mlocal) * 8 + xcd, it = 0 .. nit-1 (blocks of one XCD
    // that share a pixel tile - different column tiles - are neighbours in the dispatch order and run at the same time)
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, bj = blockIdx.x >> 3;
    const int tile_n = bj % tiles_n, mlocal = bj / tiles_n, mper = (G >> 3) / tiles_n;

    const int k4 = t & 7;      // 4-channel group of the 32-wide k-step staged by this thread (8 lanes = one 128-byte line)
    const int r0 = t >> 3;     // rows r0 + 32 i of the tile
    const int nbt = d.cout_pad >> 5, kst = d.kpad >> 4;
    const size_t wplane = (size_t)nbt * kst * 512;
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(d.w_split, (unsigned)(wplane * NSB * sizeof(elem_t)));
    const unsigned wbase = (unsigned)((((size_t)(tile_n * (BN / 32)) * kst) * 512) * sizeof(elem_t));
    const unsigned wlane = (unsigned)lane * 16u;
    const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(d.in, (unsigned)((size_t)d.N * d.H * d.W * d.in_ld * sizeof(float)));
    const __amdgpu_buffer_rsrc_t orsrc = make_rsrc(d.out, (unsigned)((size_t)d.N * d.Ho * d.Wo * d.out_ld * sizeof(float)));
    const int rs = d.res_shift;
    const __amdgpu_buffer_rsrc_t rrsrc = make_rsrc(HAS_RES ? d.res : d.in, HAS_RES ? (unsigned)((size_t)d.N * (d.Ho >> rs) * (d.Wo >> rs) * d.res_ld * sizeof(float)) : 0u);
    const unsigned ld4 = (unsigned)d.in_ld * 4u;
    const unsigned acoff = (unsigned)(d.in_coff + k4 * 4) * 4u;
    const bool plain = d.stride == 1;            // input pixel index == output pixel index

    // ---- load pipeline state: the next activation tile to request is k-step `lk` of iteration `lit`; the weights' k-step is `bk`
    int lk = 0, lit = 0, bk = 0;
    unsigned rowoff[4];        // byte offset of this thread's 4 rows of the tile being requested (0xFFFFFFF0: row outside)
    auto set_rows = [&](const int it) {
        const int tm = (it * mper + mlocal) * 8 + xcd;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = tm * BM + r0 + 32 * i;
            const bool ok = tm < tiles_m && m < M;
            unsigned pix = (unsigned)m;
            if (!plain) {
                const int mm = ok ? m : 0;
                const int qx = mm % d.Qw, tq = mm / d.Qw;
                const int qy = tq % d.Qh, n = tq / d.Qh;
                pix = (unsigned)((n * d.H + qy * d.stride) * d.W + qx * d.stride);
            }
            rowoff[i] = ok ? pix * ld4 + acoff : 0xFFFFFFF0u;
        }
    };
    f32x4 areg[2][4];
    x8 breg[2][NBL];
    float amax = 0.f;

    auto load_A = [&](const int slot) {
        const unsigned koff = (unsigned)lk * (BK * 4u);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            areg[slot][i] = buffer_load16<f32x4>(arsrc, rowoff[i] != 0xFFFFFFF0u ? rowoff[i] + koff : 0xFFFFFFF0u, 0u);
        if (++lk == nk) {
            lk = 0;
            ++lit;
            set_rows(lit);        // iterations past the end: every row outside -> zeros, never used
        }
    };
    // this thread's 16-byte chunks c = t + 256 j of fragment f = c / 64 = wave + 4 j = (plane * 2 + slab) * (BN/32) + column block
    auto load_B = [&](const int slot) {
#pragma unroll
        for (int j = 0; j < NBL; ++j) {
            const int f = wave + 4 * j, bcol = f % (BN / 32), pm = f / (BN / 32);
            breg[slot][j] = buffer_load16<x8>(wrsrc, wlane, wbase + (unsigned)(((size_t)(pm >> 1) * wplane + ((size_t)bcol * kst + 2 * bk + (pm & 1)) * 512) * sizeof(elem_t)));
        }
        if (++bk == nk) bk = 0;
    };
    auto store_A = [&](const int i, const int buf, const int slot) {
        x4 sp[NSA];
        split_act<PMODE>(areg[slot][i], sp, amax);
        const int row = r0 + 32 * i;
#pragma unroll
        for (int p = 0; p < NSA; ++p)
            *reinterpret_cast<x4*>(&As[buf * ABUF + p * (BM * LDS_LDH) + row * LDS_LDH + (((k4 >> 1) ^ lds_swz(row)) << 3) + ((k4 & 1) << 2)]) = sp[p];
    };
    auto store_B = [&](const int j, const int buf, const int slot) {
        *reinterpret_cast<x8*>(&Bs[buf * BBUF + (t + 256 * j) * 8]) = breg[slot][j];
    };

    const int frag_row = (wm * TM * 32 + (lane & 31)) * LDS_LDH;
    const int frag_sw = lds_swz(lane & 31);
    const int frag_chunk[2] = {(((lane >> 5)) ^ frag_sw) << 3, ((2 + (lane >> 5)) ^ frag_sw) << 3};
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

    // ---- prologue: scale / shift of the column tile -> LDS; k-step 0 of the first tile staged in buffers 0; then the requests a tile
    // finds in flight when it starts - A(1) W(1) in slots 1, A(2) W(2) in slots 0 - followed by as many (dropped) stores as an epilogue
    // issues: the tile loop's header then merges two equal wait states and its waits stay counted
    {
        const int c = t & (BN - 1);
        const int co = tile_n * BN + c;
        float v = t < BN ? 1.f : 0.f;
        if (co < d.cout) {
            if (t < BN && d.scale) v = d.scale[co];
            if (t >= BN && d.shift) v = d.shift[co];
        }
        if (t < 2 * BN) scsh[t] = v;
    }
    set_rows(0);
    load_B(0);
    load_A(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_A(i, 0, 0);
#pragma unroll
    for (int j = 0; j < NBL; ++j) store_B(j, 0, 0);
    load_B(1);
    load_A(1);
    if constexpr (NKU == 2) {
        // two k-steps: the weights of BOTH stay in the two weight buffers for the life of the block (a block keeps its column tile) -
        // no weight request, no weight staging in the tile loop: half of the loop's vector-memory instructions and LDS stores
#pragma unroll
        for (int j = 0; j < NBL; ++j) store_B(j, 1, 1);
    } else {
        load_B(0);
    }
    load_A(0);
    {
        // (distinct offsets past the end of any buffer the launcher admits: identical stores would be merged into one)
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NST; ++i) __builtin_amdgcn_raw_buffer_store_b128(z, orsrc, (int)(0xFFFFFE00u + 16u * (unsigned)i), 0, 0);
    }
    __syncthreads();

    constexpr int NM1 = NT * TM * TN;                // MFMAs per wave and slab
    constexpr int NW = 8;                            // work items of slab 1: 4 row stagings, the weight staging in two halves, the two requests

    // one k-step: MFMAs on buffers `cur`; k-step + 1 (register slots cur ^ 1) is staged into buffers cur ^ 1, k-step + 3 requested
    // into the slots just freed. FIRST: the first products of a tile start from zero accumulators (no separate clear)
    auto kstep = [&](auto cur_tag, auto first_tag) {
        constexpr int cur = decltype(cur_tag)::value;
        constexpr bool first = decltype(first_tag)::value;
        constexpr int slot = cur ^ 1;
        read_A(0, cur);
#pragma unroll
        for (int p = 0; p < NLB; ++p) read_B(0, p, cur);
        read_A(1, cur);
        __builtin_amdgcn_sched_barrier(0);
        auto prep = [&](const int q) {
            if (SM::PB[q] >= NLB) {
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[SM::PB[q]][b] = derive_weight_plane<PMODE>(bf[0][b]);
            }
        };
        // ---- slab 0: MFMAs, and after the last product that uses weight plane p, that plane's fragments of slab 1
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            prep(q);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    if (first && q == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[a][b] = split_mfma<PMODE>(bf[SM::PB[q]][b], af[0][SM::PA[q]][a], z);
                    } else {
                        acc[a][b] = split_mfma<PMODE>(bf[SM::PB[q]][b], af[0][SM::PA[q]][a], acc[a][b]);
                    }
                }
            bool last_use = SM::PB[q] < NLB;               // derived planes are not refilled
#pragma unroll
            for (int q2 = q + 1; q2 < NT; ++q2) last_use = last_use && SM::PB[q2] != SM::PB[q];
            // plane 0 is also the source of the derived plane: it has to outlive the product that multiplies with that one
#pragma unroll
            for (int q2 = q + 1; q2 < NT; ++q2) last_use = last_use && !(SM::PB[q] == 0 && SM::PB[q2] >= NLB);
            if (last_use) {
                __builtin_amdgcn_sched_barrier(0);
                read_B(1, SM::PB[q], cur);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- slab 1: MFMAs with the step's other work between them (program order pinned)
        auto work = [&](const int w) {
            if (w < 4) store_A(w, cur ^ 1, slot);
            else if (w == 4) {
                if constexpr (NKU != 2) {
#pragma unroll
                    for (int j = 0; j < (NBL + 1) / 2; ++j) store_B(j, cur ^ 1, slot);
                }
            } else if (w == 5) {
                if constexpr (NKU != 2) {
#pragma unroll
                    for (int j = (NBL + 1) / 2; j < NBL; ++j) store_B(j, cur ^ 1, slot);
                }
            } else if (w == 6) {
                if constexpr (NKU != 2) load_B(slot);
            } else load_A(slot);
        };
        int mf = 0;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            prep(q);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = split_mfma<PMODE>(bf[SM::PB[q]][b], af[1][SM::PA[q]][a], acc[a][b]);
                    ++mf;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
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

    // ---- epilogue on the transposed accumulators (conv_common.h): a lane owns pixel (lane & 31) of its TM sub-tiles and, per
    // register group g, channels cbase + 32 b + 8 g .. + 3. Same arithmetic as conv_epilogue's float4 path.
    const int prow = lane & 31;
    const int cl = wn * TN * 32 + 4 * (lane >> 5);          // first of this lane's channels inside the column tile (+ 32 b + 8 g)
    const unsigned ocol4 = (unsigned)(d.out_coff + tile_n * BN + cl) * 4u;
    const unsigned rcol4 = (unsigned)(d.res_coff + tile_n * BN + cl) * 4u;
    // the activation without a branch per element: y = max(t, 0) + ns * min(t, 0), ns = 0 (ReLU) | slope (leaky) | 1 (none): vps_act's values
    const float ns = d.act == VPS_ACT_RELU ? 0.f : (d.act == VPS_ACT_LEAKY ? d.slope : 1.f);
    auto epilogue = [&](const int it) {
        const int tm = (it * mper + mlocal) * 8 + xcd;
        unsigned ooff[TM], roff[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int m = tm * BM + wm * TM * 32 + a * 32 + prow;
            const bool inside = tm < tiles_m && m < M;
            ooff[a] = inside ? (unsigned)m * ((unsigned)d.out_ld * 4u) + ocol4 : 0xFFFFFFF0u;
            roff[a] = 0xFFFFFFF0u;
            if constexpr (HAS_RES) {
                unsigned rp = (unsigned)m;
                if (rs) {
                    const int mm = inside ? m : 0;
                    const int qx = mm % d.Qw, tq = mm / d.Qw;
                    const int qy = tq % d.Qh, n = tq / d.Qh;
                    rp = (unsigned)((n * (d.Ho >> rs) + (qy >> rs)) * (d.Wo >> rs) + (qx >> rs));
                }
                roff[a] = inside ? rp * ((unsigned)d.res_ld * 4u) + rcol4 : 0xFFFFFFF0u;
            }
        }
        f32x4 rv[HAS_RES ? TM : 1][HAS_RES ? TN : 1][4];
        if constexpr (HAS_RES) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const bool cok = tile_n * BN + cl + b * 32 + 8 * g < d.cout;
                        rv[a][b][g] = buffer_load16<f32x4>(rrsrc, (cok && roff[a] != 0xFFFFFFF0u) ? roff[a] + (unsigned)(b * 32 + 8 * g) * 4u : 0xFFFFFFF0u, 0u);
                    }
        }
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bool cok = tile_n * BN + cl + b * 32 + 8 * g < d.cout;      // cout % 4 == 0: a group is valid or invalid as a whole
                const f32x4 sc = *reinterpret_cast<const f32x4*>(&scsh[cl + b * 32 + 8 * g]);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(&scsh[BN + cl + b * 32 + 8 * g]);
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float tv = acc[a][b][4 * g + e] * sc[e] + sh[e];
                        if constexpr (HAS_RES) tv += rv[a][b][g][e];
                        v[e] = fmaxf(tv, 0.f) + ns * fminf(tv, 0.f);
                    }
                    const unsigned off = (cok && ooff[a] != 0xFFFFFFF0u) ? ooff[a] + (unsigned)(b * 32 + 8 * g) * 4u : 0xFFFFFFF0u;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, (int)off, 0, 0);
                }
            }
    };

    typedef std::integral_constant<int, 0> C0;
    typedef std::integral_constant<int, 1> C1;
    typedef std::integral_constant<bool, true> FT;
    typedef std::integral_constant<bool, false> FF;
    if constexpr (NKU > 0) {
        // nk == NKU, the tile body is straight-line code: every wait of a tile is an exact count, the stores of the previous epilogue
        // stay in flight until a k-step needs something that was requested after them (k-step 2, or k-step 0 of the tile after the
        // next one when nk == 2). An inner loop over the k-steps made the register allocator split the live ranges of the load slots
        // at the loop boundaries - copies of registers with loads in flight, i.e. a vmcnt(0) every two k-steps.
        for (int it = 0; it < nit; ++it) {
            kstep(C0{}, FT{});
            kstep(C1{}, FF{});
#pragma unroll
            for (int s = 2; s < NKU; s += 2) {
                kstep(C0{}, FF{});
                kstep(C1{}, FF{});
            }
            epilogue(it);
        }
    } else {
        // any even nk: ONE loop over the pairs of k-steps of all tiles, the epilogue behind a tile's last pair. The waits behind the
        // epilogue are merged with the path around it, i.e. the first of them drains the stores - one k-step pair of >= 8 per tile
        int ks = 0, it = 0;
        const int npairs = nit * (nk >> 1);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        for (int pr = 0; pr < npairs; ++pr) {
            kstep(C0{}, FF{});
            kstep(C1{}, FF{});
            ks += 2;
            if (ks == nk) {
                epilogue(it);
                ++it;
                ks = 0;
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
            }
        }
    }
    report_range<PMODE>(d, amax);
}

int pw_resident(int tn) {
    static int res[2] = {0, 0};
    int& r = res[tn == 2 ? 1 : 0];
    if (!r) {
        int per_cu = 0, dev = 0, cus = 256;
        if (tn == 2) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_pw_kernel<2, false, 0>, 256, 0);
        else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_pw_kernel<1, false, 0>, 256, 0);
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        r = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 256);
        if (getenv("VPS_DEBUG_OCC")) fprintf(stderr, "[vps] pointwise kernel <TN %d>: %d blocks/CU\n", tn, per_cu);
    }
    return r;
}

}  // namespace

// -> 1 if the persistent pointwise kernel takes this launch and was enqueued, 0 if the caller has to use another kernel.
// VPS_PW=0 in the environment switches the family off (A/B runs); VPS_PW_MIN_ROUNDS=r: layers with fewer than r rounds of resident
// blocks stay on the uniform-lead kernel (persistence pays from the second tile of a block on).
__attribute__((visibility("hidden")))
int vpsi_launch_conv_pw(const vps_conv_desc& d, int M, int tiles_m, int tiles_n, hipStream_t s) {
    const char* const on_env = getenv("VPS_PW");                 // read per call: tests switch it inside one process
    const int on = on_env ? atoi(on_env) : 1;
    static const int min_rounds = getenv("VPS_PW_MIN_ROUNDS") ? atoi(getenv("VPS_PW_MIN_ROUNDS")) : 2;
    if (!on || d.prec != VPS_PREC_F16X3 || d.offset || d.KH != 1 || d.KW != 1 || d.nclass != 1 || d.ksplit != 1 || d.gn_stats) return 0;
    if (d.pad_y[0] || d.pad_x[0] || (d.cin_pad & 31) || d.kpad != d.cin_pad || (d.tile_n != 128 && d.tile_n != 64)) return 0;
    const int nk = d.kpad / BK;
    if (nk < 2 || (nk & 1)) return 0;
    if (d.Ho != d.Qh || d.Wo != d.Qw || (d.stride == 1 && (d.H != d.Qh || d.W != d.Qw))) return 0;
    // float4 buffer stores / residual loads: 16-byte aligned channel windows of tensors below 4 GiB
    if (((d.cout | d.out_ld | d.out_coff) & 3) || ((uintptr_t)d.out & 15) || (size_t)d.N * d.Ho * d.Wo * d.out_ld * sizeof(float) >= 0xFFFFFE00ull) return 0;
    if (d.res && (((d.res_ld | d.res_coff) & 3) || ((uintptr_t)d.res & 15) || d.res_shift < 0 || d.res_shift > 4 ||
                  (size_t)d.N * (d.Ho >> d.res_shift) * (d.Wo >> d.res_shift) * d.res_ld * sizeof(float) >= 0xFFFFFE00ull)) return 0;
    if ((size_t)d.N * d.H * d.W * d.in_ld * sizeof(float) >= 0xFFFFFE00ull) return 0;
    const int tn = d.tile_n == 128 ? 2 : 1;
    const int resident = pw_resident(tn);
    const long total = (long)tiles_m * tiles_n;
    const int unit = 8 * tiles_n;                              // a grid is whole groups of (8 XCDs x the column tiles)
    if (resident < unit || total < (long)min_rounds * resident) return 0;
    // whole rounds: every block walks the same number of tiles
    const long rounds = (total + resident - 1) / resident;
    long grid = (total + rounds - 1) / rounds;
    grid = (grid + unit - 1) / unit * unit;
    if (grid > resident) grid = resident / unit * unit;
    const int mper = (int)(grid / unit);
    const int nit = (tiles_m + mper * 8 - 1) / (mper * 8);
#define VPS_PW_LAUNCH(TNV, RESV, NKV) hipLaunchKernelGGL((conv_pw_kernel<TNV, RESV, NKV>), dim3((unsigned)grid), dim3(256), 0, s, d, M, tiles_m, tiles_n, nk, nit)
#define VPS_PW_NK(TNV, RESV)                                         \
    do {                                                             \
        if (nk == 2) VPS_PW_LAUNCH(TNV, RESV, 2);                    \
        else if (nk == 4) VPS_PW_LAUNCH(TNV, RESV, 4);               \
        else if (nk == 8) VPS_PW_LAUNCH(TNV, RESV, 8);               \
        else VPS_PW_LAUNCH(TNV, RESV, 0);                            \
    } while (0)
    if (d.res) { if (tn == 2) VPS_PW_NK(2, true); else VPS_PW_NK(1, true); }
    else { if (tn == 2) VPS_PW_NK(2, false); else VPS_PW_NK(1, false); }
#undef VPS_PW_NK
#undef VPS_PW_LAUNCH
    return 1;
}
