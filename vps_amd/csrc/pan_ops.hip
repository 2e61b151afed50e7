// Panoptic head kernels: MaskRemoval (per-box overlap test on the device) and the fused
// "upsample fcn_score x4 + SegTerm crop + mask-logit paste + argmax" combine.
// Reference (relative to /root/reference/mmdet/models): utils/mask_removal.py:29-92,
// utils/unary_logits.py:81-108, panoptic/upsnetFPN.py:81, detectors/panoptic_fusetrack.py:585-597.
#include "common.h"
#include <cstdlib>

namespace {

// cv2.resize(src 28x28 float32, (w,h), INTER_LINEAR): half-pixel centres, source coordinate computed in
// double then cast to float, edge taps clamped with zero weight, horizontal pass then vertical pass.
__device__ __forceinline__ void cv_lin_coord(int d, int dst, int src, int& s0, int& s1, float& f) {
    const double scale = (double)src / (double)dst;
    float fx = (float)(((double)d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= src - 1) { fx = 0.f; sx = src - 1; }
    s0 = sx;
    s1 = min(sx + 1, src - 1);
    f = fx;
}

__device__ __forceinline__ float lerp_taps(const float* __restrict__ m28, int S, int x0, int x1, float fx, int y0, int y1, float fy) {
    const float r0 = m28[y0 * S + x0] * (1.f - fx) + m28[y0 * S + x1] * fx;
    const float r1 = m28[y1 * S + x0] * (1.f - fx) + m28[y1 * S + x1] * fx;
    return r0 * (1.f - fy) + r1 * fy;
}

__device__ __forceinline__ float resized_logit(const float* __restrict__ m28, int S, int dx, int dy, int w, int h) {
    int x0, x1, y0, y1; float fx, fy;
    cv_lin_coord(dx, w, S, x0, x1, fx);
    cv_lin_coord(dy, h, S, y0, y1, fy);
    return lerp_taps(m28, S, x0, x1, fx, y0, y1, fy);
}

struct BoxGeom {
    int bx1, by1, w, h;      // int32-truncated box origin and resize target size
    int x0, y0, x1, y1;      // clipped paste region [x0,x1) x [y0,y1)
};

__device__ __forceinline__ BoxGeom box_geom(int bx1, int by1, int bx2, int by2, int H, int W) {
    BoxGeom g;
    g.bx1 = bx1; g.by1 = by1;
    g.w = max(bx2 - bx1 + 1, 1);
    g.h = max(by2 - by1 + 1, 1);
    g.x0 = max(bx1, 0); g.x1 = min(bx2 + 1, W);
    g.y0 = max(by1, 0); g.y1 = min(by2 + 1, H);
    return g;
}

// counts[0] += #(logit>0) inside the clipped box, counts[1] += #(logit>0 and occupancy>=1)
__global__ __launch_bounds__(256)
void mask_count_kernel(const float* __restrict__ logit, int S, int bx1, int by1, int bx2, int by2, int H, int W,
                       const uint8_t* __restrict__ occ, int* __restrict__ counts) {
    const BoxGeom g = box_geom(bx1, by1, bx2, by2, H, W);
    const int rw = g.x1 - g.x0, rh = g.y1 - g.y0;
    int ms = 0, ov = 0;
    if (rw > 0 && rh > 0) {
        const long total = (long)rw * rh;
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
            const int xx = g.x0 + (int)(idx % rw), yy = g.y0 + (int)(idx / rw);
            const float v = resized_logit(logit, S, xx - g.bx1, yy - g.by1, g.w, g.h);
            if (v > 0.f) {
                ++ms;
                if (occ[(size_t)yy * W + xx] >= 1) ++ov;
            }
        }
    }
    for (int off = 32; off >= 1; off >>= 1) { ms += __shfl_xor(ms, off, 64); ov += __shfl_xor(ov, off, 64); }
    if ((threadIdx.x & 63) == 0 && (ms | ov)) { atomicAdd(&counts[0], ms); atomicAdd(&counts[1], ov); }
}

// keep iff mask_sum != 0 and overlap/mask_sum <= thr; kept boxes add their binary mask to the occupancy plane
__global__ __launch_bounds__(256)
void mask_commit_kernel(const float* __restrict__ logit, int S, int bx1, int by1, int bx2, int by2, int H, int W,
                        uint8_t* __restrict__ occ, const int* __restrict__ counts, double thr, int* __restrict__ flag) {
    const int ms = counts[0], ov = counts[1];
    const bool keep = ms != 0 && !((double)ov / (double)ms > thr);
    if (blockIdx.x == 0 && threadIdx.x == 0) *flag = keep ? 1 : 0;
    if (!keep) return;
    const BoxGeom g = box_geom(bx1, by1, bx2, by2, H, W);
    const int rw = g.x1 - g.x0, rh = g.y1 - g.y0;
    if (rw <= 0 || rh <= 0) return;
    const long total = (long)rw * rh;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int xx = g.x0 + (int)(idx % rw), yy = g.y0 + (int)(idx / rw);
        const float v = resized_logit(logit, S, xx - g.bx1, yy - g.by1, g.w, g.h);
        if (v > 0.f) occ[(size_t)yy * W + xx] += 1;
    }
}

// ------------------------------------------------------------------------------------------------
// The whole MaskRemoval box loop (mask_removal.py:56-88) in ONE launch: masks only interact within a class, so one
// 1024-thread workgroup per class walks the score-sorted box list, skips other classes, and for each of its boxes
// counts (block reduction), decides and commits. Replaces 3 launches per box (memset, count, commit).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void mask_removal_kernel(const float* __restrict__ logits, int S, const int* __restrict__ boxes, const int* __restrict__ cls0,
                         const int* __restrict__ mask_idx, int n, int H, int W, uint8_t* __restrict__ occ_all, double thr,
                         int* __restrict__ flags) {
    __shared__ int red[2][16];
    __shared__ int decision;
    const int cls = blockIdx.x;
    uint8_t* __restrict__ occ = occ_all + (size_t)cls * H * W;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = 0; i < n; ++i) {
        if (cls0[i] != cls) continue;                                   // uniform across the block
        const BoxGeom g = box_geom(boxes[4 * i], boxes[4 * i + 1], boxes[4 * i + 2], boxes[4 * i + 3], H, W);
        const float* __restrict__ lg = logits + (size_t)mask_idx[i] * S * S;
        const int rw = g.x1 - g.x0, rh = g.y1 - g.y0;
        const long total = (rw > 0 && rh > 0) ? (long)rw * rh : 0;
        int ms = 0, ov = 0;
        for (long idx = t; idx < total; idx += 1024) {
            const int xx = g.x0 + (int)(idx % rw), yy = g.y0 + (int)(idx / rw);
            if (resized_logit(lg, S, xx - g.bx1, yy - g.by1, g.w, g.h) > 0.f) {
                ++ms;
                if (occ[(size_t)yy * W + xx] >= 1) ++ov;
            }
        }
        for (int off = 32; off >= 1; off >>= 1) { ms += __shfl_xor(ms, off, 64); ov += __shfl_xor(ov, off, 64); }
        if (lane == 0) { red[0][wave] = ms; red[1][wave] = ov; }
        __syncthreads();
        if (t == 0) {
            int a = 0, b = 0;
            for (int w = 0; w < 16; ++w) { a += red[0][w]; b += red[1][w]; }
            const int keep = (a != 0 && !((double)b / (double)a > thr)) ? 1 : 0;
            decision = keep;
            flags[i] = keep;
        }
        __syncthreads();
        if (decision) {
            for (long idx = t; idx < total; idx += 1024) {
                const int xx = g.x0 + (int)(idx % rw), yy = g.y0 + (int)(idx / rw);
                if (resized_logit(lg, S, xx - g.bx1, yy - g.by1, g.w, g.h) > 0.f) occ[(size_t)yy * W + xx] += 1;
            }
        }
        __threadfence_block();
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Level-batched MaskRemoval: the host orders the boxes into dependency levels (a box depends on the EARLIER boxes of the
// same class whose rectangles intersect it); all boxes of one level are independent, so one count launch and one commit
// launch per level process them in parallel (grid.y = box within the level, grid.x = pixel blocks).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void mask_level_count_kernel(const float* __restrict__ logits, int S, const int* __restrict__ boxes, const int* __restrict__ cls0,
                             const int* __restrict__ mask_idx, const int* __restrict__ level, int H, int W,
                             const uint8_t* __restrict__ occ_all, int* __restrict__ counts) {
    const int i = level[blockIdx.y];
    const BoxGeom g = box_geom(boxes[4 * i], boxes[4 * i + 1], boxes[4 * i + 2], boxes[4 * i + 3], H, W);
    const float* __restrict__ lg = logits + (size_t)mask_idx[i] * S * S;
    const uint8_t* __restrict__ occ = occ_all + (size_t)cls0[i] * H * W;
    const int rw = g.x1 - g.x0, rh = g.y1 - g.y0;
    int ms = 0, ov = 0;
    if (rw > 0 && rh > 0) {
        const long total = (long)rw * rh;
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
            const int xx = g.x0 + (int)(idx % rw), yy = g.y0 + (int)(idx / rw);
            if (resized_logit(lg, S, xx - g.bx1, yy - g.by1, g.w, g.h) > 0.f) {
                ++ms;
                if (occ[(size_t)yy * W + xx] >= 1) ++ov;
            }
        }
    }
    for (int off = 32; off >= 1; off >>= 1) { ms += __shfl_xor(ms, off, 64); ov += __shfl_xor(ov, off, 64); }
    if ((threadIdx.x & 63) == 0 && (ms | ov)) { atomicAdd(&counts[2 * i], ms); atomicAdd(&counts[2 * i + 1], ov); }
}

__global__ __launch_bounds__(256)
void mask_level_commit_kernel(const float* __restrict__ logits, int S, const int* __restrict__ boxes, const int* __restrict__ cls0,
                              const int* __restrict__ mask_idx, const int* __restrict__ level, int H, int W,
                              uint8_t* __restrict__ occ_all, const int* __restrict__ counts, double thr, int* __restrict__ flags) {
    const int i = level[blockIdx.y];
    const int ms = counts[2 * i], ov = counts[2 * i + 1];
    const bool keep = ms != 0 && !((double)ov / (double)ms > thr);
    if (blockIdx.x == 0 && threadIdx.x == 0) flags[i] = keep ? 1 : 0;
    if (!keep) return;
    const BoxGeom g = box_geom(boxes[4 * i], boxes[4 * i + 1], boxes[4 * i + 2], boxes[4 * i + 3], H, W);
    const float* __restrict__ lg = logits + (size_t)mask_idx[i] * S * S;
    uint8_t* __restrict__ occ = occ_all + (size_t)cls0[i] * H * W;
    const int rw = g.x1 - g.x0, rh = g.y1 - g.y0;
    if (rw <= 0 || rh <= 0) return;
    const long total = (long)rw * rh;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int xx = g.x0 + (int)(idx % rw), yy = g.y0 + (int)(idx / rw);
        if (resized_logit(lg, S, xx - g.bx1, yy - g.by1, g.w, g.h) > 0.f) occ[(size_t)yy * W + xx] += 1;
    }
}

// ------------------------------------------------------------------------------------------------
// MaskRemoval in ONE launch, dependency-driven (round 5; replaces ~20 level launch pairs = 0.6 ms of a frame): one workgroup per
// box of the score-sorted walk. Box i depends on the EARLIER boxes j < i of its class whose rectangles intersect it (the only ones
// whose occupancy it can see, mask_removal.py:75-80); it finds them itself, waits until each has published its decision, then counts,
// decides and commits like the level kernels. Independent boxes run side by side, chains run back to back without a launch in between.
//   * cross-workgroup traffic goes through AGENT-scope relaxed atomics (occupancy words: atomic OR / atomic load; `done` flags: atomic
//     store / load), which are coherent across the XCDs' L2s access by access - no L2 write-back / invalidate fences (a device-scope
//     fence per workgroup costs an L2 flush beside the conv kernels of the other streams: DESIGN.md 3.1, split-K last-block experiment);
//     the flag is published after `s_waitcnt vmcnt(0)` of every lane + a barrier: all occupancy updates of the box are performed by then.
//   * no deadlock: a box waits for LOWER workgroup indices only and workgroups are dispatched in index order. The wait is bounded all
//     the same (SPIN_LIMIT polls ~ 1 s): on expiry status bit 2 is raised and the box decides on what it sees - a wedged GPU is not an option.
//   * cv2.resize coordinates (an fp64 division per axis) are tabulated once per box in LDS instead of per pixel, the 28x28 logits sit
//     in LDS; a thread handles one 4-pixel occupancy word per step (W % 4 == 0).
// ------------------------------------------------------------------------------------------------
constexpr int MR_THREADS = 1024;
constexpr int MR_TAB = 2048;             // largest box edge with tabulated coordinates (larger: per-pixel coordinates)
constexpr int MR_SPIN_LIMIT = 1 << 22;

__global__ __launch_bounds__(MR_THREADS)
void mask_removal_dep_kernel(const float* __restrict__ logits, int S, const int* __restrict__ boxes, const int* __restrict__ cls0,
                             const int* __restrict__ mask_idx, int n, int H, int W, unsigned* __restrict__ occ_words, double thr,
                             int* __restrict__ flags, int* __restrict__ done, int* __restrict__ status, const int spin_limit) {
    __shared__ float m28[32 * 32];
    __shared__ short tx0[MR_TAB], tx1[MR_TAB], ty0[MR_TAB], ty1[MR_TAB];
    __shared__ float tfx[MR_TAB], tfy[MR_TAB];
    __shared__ int deps[256];
    __shared__ int ndeps, red[2][MR_THREADS / 64], decision;
    const int i = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const BoxGeom g = box_geom(boxes[4 * i], boxes[4 * i + 1], boxes[4 * i + 2], boxes[4 * i + 3], H, W);
    const int cls = cls0[i];
    const int rw = g.x1 - g.x0, rh = g.y1 - g.y0;
    const bool tab = g.w <= MR_TAB && g.h <= MR_TAB && S <= 32;
    if (t == 0) ndeps = 0;
    for (int k = t; k < S * S; k += MR_THREADS) m28[k] = logits[(size_t)mask_idx[i] * S * S + k];
    if (tab) {
        // coordinates of the CLIPPED columns / rows only (index = position inside the clipped region)
        for (int k = t; k < rw; k += MR_THREADS) { int a, b; float f; cv_lin_coord(g.x0 + k - g.bx1, g.w, S, a, b, f); tx0[k] = (short)a; tx1[k] = (short)b; tfx[k] = f; }
        for (int k = t; k < rh; k += MR_THREADS) { int a, b; float f; cv_lin_coord(g.y0 + k - g.by1, g.h, S, a, b, f); ty0[k] = (short)a; ty1[k] = (short)b; tfy[k] = f; }
    }
    __syncthreads();
    // ---- dependencies: earlier same-class boxes whose clipped rectangles intersect this one (the host's level rule, panoptic_ops.py)
    for (int j = t; j < i; j += MR_THREADS) {
        if (cls0[j] != cls) continue;
        const BoxGeom q = box_geom(boxes[4 * j], boxes[4 * j + 1], boxes[4 * j + 2], boxes[4 * j + 3], H, W);
        if (q.x0 < g.x1 && g.x0 < q.x1 && q.y0 < g.y1 && g.y0 < q.y1) deps[atomicAdd(&ndeps, 1)] = j;
    }
    __syncthreads();
    if (t < ndeps) {
        int spins = 0;
        while (__hip_atomic_load(&done[deps[t]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > spin_limit) { atomicOr(status, 4); break; }
        }
    }
    __syncthreads();
    auto logit_at = [&](const int kx, const int ky) -> float {
        if (tab) return lerp_taps(m28, S, tx0[kx], tx1[kx], tfx[kx], ty0[ky], ty1[ky], tfy[ky]);
        return resized_logit(m28, S, g.x0 + kx - g.bx1, g.y0 + ky - g.by1, g.w, g.h);
    };
    // ---- the clipped region in 4-pixel words: columns xw0 .. xw1 (word-aligned), rows y0 .. y1
    const int xw0 = g.x0 & ~3, nwx = rw > 0 ? ((g.x1 + 3) >> 2) - (xw0 >> 2) : 0;
    const long nwords = rh > 0 ? (long)nwx * rh : 0;
    unsigned* __restrict__ plane = occ_words + ((size_t)cls * H * W >> 2);
    int ms = 0, ov = 0;
    // four words per thread and step: the four occupancy loads (agent scope: they miss the L2, ~2 us each) are requested together,
    // unconditionally (a word without a positive pixel is masked out afterwards) - one load per step made a 200 x 200 box a chain
    // of ten memory round trips per pass
    for (long wd0 = t; wd0 < nwords; wd0 += 4 * MR_THREADS) {
        unsigned posm[4], o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long wd = wd0 + (long)u * MR_THREADS;
            posm[u] = 0; o[u] = 0;
            if (wd < nwords) {
                const int ky = (int)(wd / nwx), xb = xw0 + 4 * (int)(wd - (long)ky * nwx);
                o[u] = __hip_atomic_load(&plane[((size_t)(g.y0 + ky) * W + xb) >> 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long wd = wd0 + (long)u * MR_THREADS;
            if (wd < nwords) {
                const int ky = (int)(wd / nwx), xb = xw0 + 4 * (int)(wd - (long)ky * nwx);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int xx = xb + e;
                    if (xx >= g.x0 && xx < g.x1 && logit_at(xx - g.x0, ky) > 0.f) posm[u] |= 0xFFu << (8 * e);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ms += __popc(posm[u]) >> 3;
            ov += __popc(o[u] & posm[u] & 0x01010101u);          // bytes of the occupancy word are 0 or 1
        }
    }
    for (int off = 32; off >= 1; off >>= 1) { ms += __shfl_xor(ms, off, 64); ov += __shfl_xor(ov, off, 64); }
    if (lane == 0) { red[0][wave] = ms; red[1][wave] = ov; }
    __syncthreads();
    if (t == 0) {
        int a = 0, b = 0;
        for (int w = 0; w < MR_THREADS / 64; ++w) { a += red[0][w]; b += red[1][w]; }
        const int keep = (a != 0 && !((double)b / (double)a > thr)) ? 1 : 0;
        decision = keep;
        flags[i] = keep;
    }
    __syncthreads();
    if (decision) {
        for (long wd = t; wd < nwords; wd += MR_THREADS) {
            const int ky = (int)(wd / nwx), xb = xw0 + 4 * (int)(wd - (long)ky * nwx);
            unsigned setm = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int xx = xb + e;
                if (xx >= g.x0 && xx < g.x1 && logit_at(xx - g.x0, ky) > 0.f) setm |= 1u << (8 * e);
            }
            if (setm) __hip_atomic_fetch_or(&plane[((size_t)(g.y0 + ky) * W + xb) >> 2], setm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // every lane's occupancy updates have been performed (agent scope) before the flag goes out
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (t == 0) __hip_atomic_store(&done[i], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------
// MaskRemoval WITHOUT a dependency chain (round 6): the decision of box i depends on the earlier boxes only through
//   mask_sum_i = #{p : p in mask_i},   overlap_i = #{p in mask_i : p in mask_j for a KEPT earlier box j of the class}
// (mask_removal.py:75-88: the occupancy plane of a class is the union of the kept masks). Give every box the bit `rank` = its position
// among the boxes of its class in the score-sorted walk (<= 64 per class) and every pixel, per class, the PATTERN of the boxes whose
// binary mask covers it. Then overlap_i = sum over the distinct patterns P with bit i of count(P) * [P & kept & (2^i - 1) != 0]:
//   * mask_pattern_kernel - one pass over the frame in 32 x 8 tiles: a workgroup bins the boxes that touch its tile, a thread builds
//     the patterns of its pixel, wavefront ballots add the per-box pixel counts (mask_sum), the patterns with two or more bits
//     (overlap regions) are counted in a per-class hash table in global memory (one atomic pair per distinct pattern and wavefront);
//   * mask_decide_kernel - one wavefront per class walks its boxes in order over the few hundred distinct patterns: no memory round
//     trip between two dependent boxes (the one-launch dependency kernel spent ~18 us per link of the chain: 365 us per frame), no
//     workgroup waits for another one, no occupancy plane (a 16 MB memset per frame).
// Integer counts, the same resized_logit > 0 test per pixel, the same double-precision threshold: decisions identical to the walk.
// A full hash table raises status bit 2 (value 4) like an expired wait of the dependency kernel: the caller repeats through vps_mask_level.
// ------------------------------------------------------------------------------------------------
constexpr int MH_HT = 2048;              // hash-table entries (64-bit key, 64-bit count) per class and group

// A class with more than 64 boxes (the synthetic bench frames: 97 of 100 detections in one class) is walked in two GROUPS of ranks:
// group 0 = ranks 0..63 exactly as above; group 1 = ranks 64..126 in a second pass once group 0 is decided - its pixel key is the
// pattern of the group-1 boxes plus bit 63 = "covered by a KEPT group-0 box" (the group-0 masks are evaluated again for the pixels
// that a group-1 mask covers). GRP selects the pass.
template <int GRP>
__global__ __launch_bounds__(256)
void mask_pattern_kernel(const float* __restrict__ logits, int S, const int* __restrict__ boxes, const int* __restrict__ cls0,
                         const int* __restrict__ mask_idx, const int* __restrict__ rank, int n, int ncls, int H, int W,
                         int* __restrict__ area, unsigned long long* __restrict__ table, const unsigned long long* __restrict__ kept0,
                         int* __restrict__ status, const int ht) {
    __shared__ int list[256];
    __shared__ int nlist;
    __shared__ unsigned clsmask;
    const int t = threadIdx.x, lane = t & 63;
    const int tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 8;
    if (t == 0) { nlist = 0; clsmask = 0; }
    __syncthreads();
    for (int b = t; b < n; b += 256) {
        const BoxGeom g = box_geom(boxes[4 * b], boxes[4 * b + 1], boxes[4 * b + 2], boxes[4 * b + 3], H, W);
        if ((rank[b] >> 6) <= GRP && g.x0 < tx0 + 32 && tx0 < g.x1 && g.y0 < ty0 + 8 && ty0 < g.y1) {
            list[atomicAdd(&nlist, 1)] = b;
            if ((rank[b] >> 6) == GRP) atomicOr(&clsmask, 1u << cls0[b]);   // classes with a box of THIS group on the tile
        }
    }
    __syncthreads();
    const int nl = nlist;
    const unsigned cm = clsmask;
    if (cm == 0) return;
    // a table is full (another workgroup said so): the call's result is void, later workgroups do not add to the damage
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4) return;
    const int x = tx0 + (t & 31), y = ty0 + (t >> 5);
    auto covers = [&](const int b) -> bool {
        const BoxGeom g = box_geom(boxes[4 * b], boxes[4 * b + 1], boxes[4 * b + 2], boxes[4 * b + 3], H, W);
        if (!(x >= g.x0 && x < g.x1 && y >= g.y0 && y < g.y1)) return false;
        return resized_logit(logits + (size_t)mask_idx[b] * S * S, S, x - g.bx1, y - g.by1, g.w, g.h) > 0.f;
    };
    for (int c = 0; c < ncls; ++c) {
        if (!(cm >> c & 1)) continue;
        unsigned long long pat = 0;
        for (int k = 0; k < nl; ++k) {
            const int b = list[k];
            if (cls0[b] != c || (rank[b] >> 6) != GRP) continue;         // uniform
            const bool pos = covers(b);
            const unsigned long long m = __ballot(pos);
            if (pos) pat |= 1ull << (rank[b] & 63);
            if (lane == 0 && m) atomicAdd(&area[b], __popcll(m));
        }
        bool multi = (pat & (pat - 1)) != 0;
        if (GRP == 1) {
            if (__ballot(pat != 0)) {                                    // a kept group-0 box over a pixel that a group-1 mask covers?
                const unsigned long long k0 = kept0[c];
                bool occ = false;
                for (int k = 0; k < nl; ++k) {
                    const int b = list[k];
                    if (cls0[b] != c || (rank[b] >> 6) != 0 || !(k0 >> rank[b] & 1ull)) continue;   // uniform
                    if (pat != 0 && !occ && covers(b)) occ = true;
                }
                if (occ) { pat |= 1ull << 63; multi = true; }
            }
        }
        // overlap regions: one (key, count) update per distinct pattern of the wavefront
        unsigned long long todo = __ballot(multi);
        unsigned long long* __restrict__ tab = table + (size_t)c * MH_HT * 2;  // the first `ht` entries are used (ht = MH_HT but in tests)
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const unsigned long long p0 = ((unsigned long long)(unsigned)__shfl((int)(pat >> 32), leader, 64) << 32) | (unsigned)__shfl((int)pat, leader, 64);
            const unsigned long long same = __ballot(multi && pat == p0);
            if (lane == leader) {
                unsigned h = (unsigned)((p0 * 0x9E3779B97F4A7C15ull) >> 53) & (unsigned)(ht - 1);
                const int nprobe = ht < 64 ? ht : 64;                    // a run of 64 taken entries = the table is (as good as) full
                int probe = 0;
                for (; probe < nprobe; ++probe, h = (h + 1) & (unsigned)(ht - 1)) {
                    const unsigned long long key = atomicCAS(&tab[2 * h], 0ull, p0);
                    if (key == 0ull || key == p0) { atomicAdd(&tab[2 * h + 1], (unsigned long long)__popcll(same)); break; }
                }
                if (probe == nprobe) atomicOr(status, 4);
            }
            todo &= ~same;
        }
    }
}

template <int GRP>
__global__ __launch_bounds__(64)
void mask_decide_kernel(const int* __restrict__ cls0, const int* __restrict__ rank, int n, const int* __restrict__ area,
                        const unsigned long long* __restrict__ table, unsigned long long* __restrict__ kept0, double thr,
                        int* __restrict__ flags) {
    __shared__ unsigned long long keys[MH_HT];
    __shared__ unsigned cnts[MH_HT];
    __shared__ int box_of_rank[64];
    __shared__ int ne, kc;
    const int c = blockIdx.x, lane = threadIdx.x;
    box_of_rank[lane] = -1;
    if (lane == 0) { ne = 0; kc = 0; }
    __syncthreads();
    for (int i = lane; i < n; i += 64)
        if (cls0[i] == c && (rank[i] >> 6) == GRP) { box_of_rank[rank[i] & 63] = i; atomicMax(&kc, (rank[i] & 63) + 1); }
    const unsigned long long* __restrict__ tab = table + (size_t)c * MH_HT * 2;
    for (int e = lane; e < MH_HT; e += 64) {
        const unsigned long long key = tab[2 * e];
        if (key) { const int slot = atomicAdd(&ne, 1); keys[slot] = key; cnts[slot] = (unsigned)tab[2 * e + 1]; }
    }
    __syncthreads();
    const int nent = ne, nbox = kc;
    unsigned long long kept = 0;
    for (int r = 0; r < nbox; ++r) {
        const int i = box_of_rank[r];
        if (i < 0) continue;                                             // (ranks are dense: not reached)
        // patterns that count: bit r and an earlier kept bit of this group - or, in group 1, bit 63 (a kept group-0 box)
        const unsigned long long earlier = (kept & ((1ull << r) - 1ull)) | (GRP == 1 ? 1ull << 63 : 0ull);
        int ov = 0;
        for (int e = lane; e < nent; e += 64) {
            const unsigned long long key = keys[e];
            if ((key >> r & 1ull) && (key & earlier)) ov += (int)cnts[e];
        }
        for (int off = 32; off >= 1; off >>= 1) ov += __shfl_xor(ov, off, 64);
        const int ms = area[i];
        const int keep = (ms != 0 && !((double)ov / (double)ms > thr)) ? 1 : 0;
        kept |= (unsigned long long)keep << r;
        if (lane == 0) flags[i] = keep;
    }
    if (GRP == 0 && lane == 0) kept0[c] = kept;
}

// ------------------------------------------------------------------------------------------------
// Fused panoptic combine. For every full-resolution pixel:
//   fcn_output[c] = bilinear x4 (align_corners=False) of fcn_score[c]           (upsnetFPN.py:81)
//   sem = argmax_c fcn_output[c]                                                 (panoptic_fusetrack.py:593)
//   stuff logits = fcn_output[0:nstuff]; instance j logit = SegTerm crop + pasted mask logit
//   pan = argmax over [stuff..., inst_0, inst_1, ...]  (softmax is monotone -> skipped; first max wins)
// Nothing of size [k,H,W] is materialised (reference: three such fp32 tensors + a 159 MB fcn_output).
// ------------------------------------------------------------------------------------------------
// Work layout: a block owns a 32 x 8 pixel tile. Its first wave bins the k instances against the tile (SegTerm crop or pasted-mask
// box touching it), keeping ascending order; the pixels then loop over that short list instead of over all k (k ~ 45, ~2 per
// tile: the per-pixel loop over k was 90 % of the kernel). An instance that misses the tile adds exactly 0 to each of its pixels,
// which still beats negative stuff logits: the FIRST such instance stays in the list (later ones cannot win: `>` is strict).
__global__ __launch_bounds__(256)
void panoptic_combine_kernel(const float* __restrict__ score, int score_ld, int Hs, int Ws, int nclass, int nstuff,
                             const vps_pan_inst* __restrict__ inst, int k, const int* __restrict__ k_dev, const float* __restrict__ mask_logits, int S,
                             uint8_t* __restrict__ pan, uint8_t* __restrict__ sem, int H, int W, int up) {
    if (k_dev) {
        k = k_dev[0];
        if (k > 255 - nstuff) {         // the uint8 map cannot name that many instances: refuse (status for the host), write nothing
            if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicOr(const_cast<int*>(k_dev) + 2, 1);
            return;
        }
    }
    __shared__ int list[256];
    __shared__ int nlist;
    const int tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 8;
    const int tx1 = min(tx0 + 32, W), ty1 = min(ty0 + 8, H);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int n = 0;
        bool have_skipped = false;
        for (int base = 0; base < k; base += 64) {
            const int j = base + lane;
            bool hit = false;
            if (j < k) {
                const vps_pan_inst in = inst[j];
                const BoxGeom g = box_geom(in.bx1, in.by1, in.bx2, in.by2, H, W);
                hit = (in.sx0 < tx1 && in.sx1 > tx0 && in.sy0 < ty1 && in.sy1 > ty0) || (g.x0 < tx1 && g.x1 > tx0 && g.y0 < ty1 && g.y1 > ty0);
            }
            unsigned long long hm = __ballot(hit);
            const unsigned long long skipped = __ballot(j < k) & ~hm;
            if (!have_skipped && skipped) {
                hm |= 1ULL << (__ffsll((long long)skipped) - 1);
                have_skipped = true;
            }
            if ((hm >> lane) & 1ULL) list[n + __popcll(hm & ((1ULL << lane) - 1ULL))] = j;
            n += __popcll(hm);
        }
        if (lane == 0) nlist = n;
    }
    // The source texels of the tile's bilinear x`up` upsampling (for up = 4: 10 x 4 texels of `nclass` logits) are staged in LDS once:
    // a pixel otherwise issues 4 x nclass strided 4-byte global reads (76 at 19 classes: the kernel was bound by the L1 tag rate,
    // 112 us for 15 MB of traffic). Same values, same arithmetic order: bit-identical maps. up = 1 or very wide maps read global memory.
    constexpr int TEX_MAX = 18 * 6, TEX_LD = 32;
    __shared__ float tex[TEX_MAX * TEX_LD];
    const float rs = 1.0f / (float)up;
    auto src = [&](const int o, const int lim) {
        float s_ = rs * ((float)o + 0.5f) - 0.5f; if (s_ < 0.f) s_ = 0.f;
        return min((int)s_, lim - 1);
    };
    const int xs0 = src(tx0, Ws), ys0 = src(ty0, Hs);
    const int TW = min(src(tx1 - 1, Ws) + 1, Ws - 1) - xs0 + 1, TH = min(src(ty1 - 1, Hs) + 1, Hs - 1) - ys0 + 1;
    const bool staged = up >= 2 && nclass <= TEX_LD && TW * TH <= TEX_MAX;
    if (staged) {
        for (int i = threadIdx.x; i < TW * TH * nclass; i += 256) {
            const int tt = i / nclass, c = i - tt * nclass;
            const int tyy = tt / TW, txx = tt - tyy * TW;
            tex[tt * TEX_LD + c] = score[((size_t)(ys0 + tyy) * Ws + xs0 + txx) * score_ld + c];
        }
    }
    __syncthreads();
    const int x = tx0 + (threadIdx.x & 31), y = ty0 + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    float sy = rs * ((float)y + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
    float sx = rs * ((float)x + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
    const int y0 = min((int)sy, Hs - 1), x0 = min((int)sx, Ws - 1);
    const int yp = y0 < Hs - 1 ? 1 : 0, xp = x0 < Ws - 1 ? 1 : 0;
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* p00 = staged ? tex + ((y0 - ys0) * TW + (x0 - xs0)) * TEX_LD : score + ((size_t)y0 * Ws + x0) * score_ld;
    const size_t tstep = staged ? (size_t)TEX_LD : (size_t)score_ld, rstep = staged ? (size_t)TW * TEX_LD : (size_t)Ws * score_ld;
    const float* p01 = p00 + (size_t)xp * tstep;
    const float* p10 = p00 + (size_t)yp * rstep;
    const float* p11 = p10 + (size_t)xp * tstep;
    float best_sem = -INFINITY, best_pan = -INFINITY;
    int i_sem = 0, i_pan = 0;
    for (int c = 0; c < nclass; ++c) {
        const float v = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
        if (v > best_sem) { best_sem = v; i_sem = c; }
        if (c < nstuff && v > best_pan) { best_pan = v; i_pan = c; }
    }
    const int n = nlist;
    for (int e = 0; e < n; ++e) {
        const int j = list[e];
        const vps_pan_inst in = inst[j];
        float v = 0.f;
        if (x >= in.sx0 && x < in.sx1 && y >= in.sy0 && y < in.sy1) {
            const int c = in.seg_ch;
            v = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
        }
        const BoxGeom g = box_geom(in.bx1, in.by1, in.bx2, in.by2, H, W);
        if (x >= g.x0 && x < g.x1 && y >= g.y0 && y < g.y1)
            v += resized_logit(mask_logits + (size_t)in.mask_idx * S * S, S, x - g.bx1, y - g.by1, g.w, g.h);
        if (v > best_pan) { best_pan = v; i_pan = nstuff + j; }
    }
    const size_t idx = (size_t)y * W + x;
    pan[idx] = (uint8_t)i_pan;
    sem[idx] = (uint8_t)i_sem;
}

}  // namespace

extern "C" int vps_mask_count(const float* logit, int S, int bx1, int by1, int bx2, int by2, int H, int W,
                              const uint8_t* occ, int32_t* counts, void* stream) {
    if (!logit || !occ || !counts || S < 2 || H <= 0 || W <= 0) return VPS_EARG(1);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), s);
    if (e != hipSuccess) return -(int)e;
    const long area = (long)max(min(bx2 + 1, W) - max(bx1, 0), 0) * max(min(by2 + 1, H) - max(by1, 0), 0);
    hipLaunchKernelGGL(mask_count_kernel, dim3(stream_grid(area > 0 ? area : 1, 256)), dim3(256), 0, s, logit, S, bx1, by1,
                       bx2, by2, H, W, occ, counts);
    return vps_launch_status();
}

extern "C" int vps_mask_commit(const float* logit, int S, int bx1, int by1, int bx2, int by2, int H, int W, uint8_t* occ,
                               const int32_t* counts, double thr, int32_t* flag, void* stream) {
    if (!logit || !occ || !counts || !flag || S < 2 || H <= 0 || W <= 0) return VPS_EARG(1);
    const long area = (long)max(min(bx2 + 1, W) - max(bx1, 0), 0) * max(min(by2 + 1, H) - max(by1, 0), 0);
    hipLaunchKernelGGL(mask_commit_kernel, dim3(stream_grid(area > 0 ? area : 1, 256)), dim3(256), 0, (hipStream_t)stream,
                       logit, S, bx1, by1, bx2, by2, H, W, occ, counts, thr, flag);
    return vps_launch_status();
}

extern "C" int vps_mask_removal(const float* logits, int S, const int32_t* boxes, const int32_t* cls0, const int32_t* mask_idx,
                                int n, int ncls, int H, int W, uint8_t* occ, double thr, int32_t* flags, void* stream) {
    if (!logits || !boxes || !cls0 || !mask_idx || !occ || !flags || S < 2 || n < 0 || ncls <= 0 || H <= 0 || W <= 0) return VPS_EARG(1);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(occ, 0, (size_t)ncls * H * W, s);
    if (e != hipSuccess) return -(int)e;
    if (n == 0) return 0;
    e = hipMemsetAsync(flags, 0, sizeof(int32_t) * n, s);
    if (e != hipSuccess) return -(int)e;
    hipLaunchKernelGGL(mask_removal_kernel, dim3(ncls), dim3(1024), 0, s, logits, S, boxes, cls0, mask_idx, n, H, W, occ, thr, flags);
    return vps_launch_status();
}

extern "C" int vps_mask_removal_dep(const float* logits, int S, const int32_t* boxes, const int32_t* cls0, const int32_t* mask_idx,
                                    int n, int ncls, int H, int W, uint8_t* occ, double thr, int32_t* flags, int32_t* done,
                                    int32_t* status, void* stream) {
    if (!logits || !boxes || !cls0 || !mask_idx || !occ || !flags || !done || !status || S < 2 || n < 0 || n > 256 || ncls <= 0 || H <= 0 || W <= 0)
        return VPS_EARG(1);
    if ((W & 3) || ((uintptr_t)occ & 3)) return VPS_EARG(2);                     // 4-pixel occupancy words
    if (S > 32) return VPS_EARG(3);                                              // the kernel keeps the S x S logits of a box in a 32 x 32 LDS array
    // VPS_MR_SPIN_LIMIT=n in the environment: polls a box spends on one dependency before it gives up with status bit 2 (tests force
    // the expiry path with 0; read per call)
    const char* const sl = getenv("VPS_MR_SPIN_LIMIT");
    const int spin_limit = sl ? atoi(sl) : MR_SPIN_LIMIT;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(occ, 0, (size_t)ncls * H * W, s);
    if (e != hipSuccess) return -(int)e;
    if (n == 0) return 0;
    e = hipMemsetAsync(done, 0, sizeof(int32_t) * n, s);
    if (e != hipSuccess) return -(int)e;
    hipLaunchKernelGGL(mask_removal_dep_kernel, dim3(n), dim3(MR_THREADS), 0, s, logits, S, boxes, cls0, mask_idx, n, H, W,
                       reinterpret_cast<unsigned*>(occ), thr, flags, done, status, spin_limit);
    return vps_launch_status();
}

extern "C" int vps_mask_removal_hist(const float* logits, int S, const int32_t* boxes, const int32_t* cls0, const int32_t* mask_idx,
                                     const int32_t* rank, int max_rank, int n, int ncls, int H, int W, int32_t* scratch, size_t scratch_bytes,
                                     double thr, int32_t* flags, int32_t* status, void* stream) {
    if (!logits || !boxes || !cls0 || !mask_idx || !rank || !scratch || !flags || !status || S < 2 || n < 0 || n > 256 || ncls <= 0 || ncls > 32 ||
        H <= 0 || W <= 0)
        return VPS_EARG(1);
    if (max_rank < 0 || max_rank > 126) return VPS_EARG(3);                              // two groups: ranks 0..63 | 64..126 (+ bit 63)
    const int ngrp = max_rank >= 64 ? 2 : 1;
    // area[n] | kept0[ncls] | table[ngrp][ncls][MH_HT][2], 64-bit words behind the int32 area
    const size_t area_bytes = ((size_t)n * sizeof(int32_t) + 7) & ~(size_t)7;
    const size_t tab_words = (size_t)ncls * MH_HT * 2;
    const size_t need = area_bytes + ((size_t)ncls + ngrp * tab_words) * sizeof(unsigned long long);
    if (scratch_bytes < need || ((uintptr_t)scratch & 7)) return VPS_EARG(2);
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return 0;
    const hipError_t e = hipMemsetAsync(scratch, 0, need, s);
    if (e != hipSuccess) return -(int)e;
    unsigned long long* kept0 = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(scratch) + area_bytes);
    unsigned long long* table = kept0 + ncls;
    // VPS_MR_HIST_CAP=c in the environment (a power of two <= 2048, read per call): entries of a class's pattern table that are used - tests
    // force the table-full status with a tiny table
    int ht = MH_HT;
    if (const char* const hc = getenv("VPS_MR_HIST_CAP")) { const int v = atoi(hc); if (v >= 1 && v <= MH_HT && !(v & (v - 1))) ht = v; }
    const dim3 tiles((W + 31) / 32, (H + 7) / 8);
    hipLaunchKernelGGL(mask_pattern_kernel<0>, tiles, dim3(256), 0, s, logits, S, boxes, cls0, mask_idx, rank, n, ncls, H, W, scratch, table,
                       (const unsigned long long*)kept0, status, ht);
    hipLaunchKernelGGL(mask_decide_kernel<0>, dim3(ncls), dim3(64), 0, s, cls0, rank, n, scratch, (const unsigned long long*)table, kept0, thr, flags);
    if (ngrp == 2) {
        hipLaunchKernelGGL(mask_pattern_kernel<1>, tiles, dim3(256), 0, s, logits, S, boxes, cls0, mask_idx, rank, n, ncls, H, W, scratch,
                           table + tab_words, (const unsigned long long*)kept0, status, ht);
        hipLaunchKernelGGL(mask_decide_kernel<1>, dim3(ncls), dim3(64), 0, s, cls0, rank, n, scratch, (const unsigned long long*)(table + tab_words), kept0,
                           thr, flags);
    }
    return vps_launch_status();
}

extern "C" int vps_mask_level(const float* logits, int S, const int32_t* boxes, const int32_t* cls0, const int32_t* mask_idx,
                              const int32_t* level, int nlevel, int max_area, int H, int W, uint8_t* occ, int32_t* counts,
                              double thr, int32_t* flags, void* stream) {
    if (!logits || !boxes || !cls0 || !mask_idx || !level || !occ || !counts || !flags || S < 2 || nlevel <= 0) return VPS_EARG(1);
    hipStream_t s = (hipStream_t)stream;
    int gx = (max_area + 255) / 256; if (gx > 256) gx = 256; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(mask_level_count_kernel, dim3(gx, nlevel), dim3(256), 0, s, logits, S, boxes, cls0, mask_idx, level, H, W, occ, counts);
    hipLaunchKernelGGL(mask_level_commit_kernel, dim3(gx, nlevel), dim3(256), 0, s, logits, S, boxes, cls0, mask_idx, level, H, W, occ,
                       counts, thr, flags);
    return vps_launch_status();
}

extern "C" int vps_panoptic_combine(const float* fcn_score, int score_ld, int Hs, int Ws, int nclass, int nstuff,
                                    const vps_pan_inst* inst, int k, const float* mask_logits, int S,
                                    uint8_t* pan, uint8_t* sem, int H, int W, void* stream) {
    if (!fcn_score || !pan || !sem || Hs <= 0 || Ws <= 0 || H <= 0 || W <= 0) return VPS_EARG(1);
    if (k < 0 || k > 255 - nstuff || (k > 0 && (!inst || !mask_logits)) || nclass < nstuff || nstuff < 0) return VPS_EARG(2);
    if (H % Hs || W % Ws || H / Hs != W / Ws) return VPS_EARG(3);
    hipLaunchKernelGGL(panoptic_combine_kernel, dim3((W + 31) / 32, (H + 7) / 8), dim3(256), 0, (hipStream_t)stream,
                       fcn_score, score_ld, Hs, Ws, nclass, nstuff, inst, k, (const int*)nullptr, mask_logits, S, pan, sem, H, W, H / Hs);
    return vps_launch_status();
}

extern "C" int vps_panoptic_combine_dev(const float* fcn_score, int score_ld, int Hs, int Ws, int nclass, int nstuff,
                                        const vps_pan_inst* inst, const int32_t* k_dev, const float* mask_logits, int S,
                                        uint8_t* pan, uint8_t* sem, int H, int W, void* stream) {
    if (!fcn_score || !pan || !sem || !inst || !k_dev || !mask_logits || Hs <= 0 || Ws <= 0 || H <= 0 || W <= 0) return VPS_EARG(1);
    if (nclass < nstuff || nstuff < 0) return VPS_EARG(2);
    if (H % Hs || W % Ws || H / Hs != W / Ws) return VPS_EARG(3);
    hipLaunchKernelGGL(panoptic_combine_kernel, dim3((W + 31) / 32, (H + 7) / 8), dim3(256), 0, (hipStream_t)stream,
                       fcn_score, score_ld, Hs, Ws, nclass, nstuff, inst, 0, k_dev, mask_logits, S, pan, sem, H, W, H / Hs);
    return vps_launch_status();
}
