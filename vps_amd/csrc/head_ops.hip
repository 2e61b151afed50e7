// Device-resident detection post-processing: the order-defining HOST logic of the reference's heads as single-workgroup kernels,
// so that a frame needs one mid-frame D2H (the detection list) instead of six stream drains.
//   rpn_collect       anchor_head.py:198-223 / rpn_head.py:55-104: per-level kept boxes -> [:nms_post] -> top max_num by score
//   maskroi_select    utils/mask_roi.py:43-95: bbox_transform + clip, class-agnostic flattening, score threshold, descending sort
//   maskroi_finish    utils/mask_roi.py:96-147: post-NMS list, max_det cap (ties kept), dummy row
//   track_assign      detectors/panoptic_fusetrack.py:424-469: arg-max of the comprehensive scores, greedy assignment with undo,
//                     new ids, in-place memory update
//   pan_instances     utils/mask_removal.py:81-91 (kept list) + utils/unary_logits.py:96-106 (SegTerm crop) -> vps_pan_inst table
// Everything here is index / threshold work on <= 8192 items: one workgroup, LDS-resident, latency-bound by design.
// Reference files relative to /root/reference/mmdet.
#include "common.h"

namespace {

typedef unsigned long long u64;

// keys[0 .. n2) sorted DESCENDING, n2 a power of two; the whole block takes part
template <int NT>
__device__ __forceinline__ void bitonic_sort_desc(u64* keys, const int n2) {
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += NT) {
                const int p = i ^ j;
                if (p > i) {
                    const u64 a = keys[i], b = keys[p];
                    const bool desc = (i & k) == 0;
                    if (desc ? a < b : a > b) { keys[i] = b; keys[p] = a; }
                }
            }
            __syncthreads();
        }
}

__device__ __forceinline__ int next_pow2(int n) {
    int p = 2;
    while (p < n) p <<= 1;
    return p;
}

constexpr int SORT_CAP = 8192;    // 64 KB of LDS keys

// ------------------------------------------------------------------------------------------------
// RPN: concatenate the kept boxes of every level (keep order = descending score, at most nms_post per level), take the
// max_num best by score (ties: the earlier position of the concatenation first, what a stable descending sort does).
// rpn_head.py:94-104. boxes [nlv][nmax][5], keep [nlv][nmax] (indices into the level's sorted list), nkeep [nlv].
// out [max_num][5] (rows >= *n_out zeroed), n_out = min(max_num, total).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void rpn_collect_kernel(const float* __restrict__ boxes, const int* __restrict__ keep, const int* __restrict__ nkeep, int nlv, int nmax,
                        int nms_post, int max_num, float* __restrict__ out, int* __restrict__ n_out) {
    __shared__ u64 keys[SORT_CAP];
    __shared__ int off[9];
    if (threadIdx.x == 0) {
        int s = 0;
        for (int l = 0; l < nlv; ++l) { off[l] = s; s += min(nkeep[l], nms_post); }
        off[nlv] = s;
    }
    __syncthreads();
    const int total = off[nlv];
    const int n2 = next_pow2(total);
    for (int i = threadIdx.x; i < n2; i += 1024) {
        u64 key = 0;
        if (i < total) {
            int l = 0;
            while (i >= off[l + 1]) ++l;
            const int src = l * nmax + keep[l * nmax + (i - off[l])];
            key = ((u64)__float_as_uint(boxes[(size_t)src * 5 + 4]) << 32) | (u64)(0xFFFFFFFFu - (unsigned)i);
        }
        keys[i] = key;
    }
    __syncthreads();
    bitonic_sort_desc<1024>(keys, n2);
    const int n = min(max_num, total);
    for (int i = threadIdx.x; i < max_num; i += 1024) {
        float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (i < n) {
            const int pos = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
            int l = 0;
            while (pos >= off[l + 1]) ++l;
            const int src = l * nmax + keep[l * nmax + (pos - off[l])];
            for (int e = 0; e < 5; ++e) v[e] = boxes[(size_t)src * 5 + e];
        }
        for (int e = 0; e < 5; ++e) out[(size_t)i * 5 + e] = v[e];
    }
    if (threadIdx.x == 0) *n_out = n;
}

// ------------------------------------------------------------------------------------------------
// RPN, per level (rpn_head.py:62-91 + core/bbox/transforms.py:34-68): objectness = sigmoid(rpn_cls) over the (h, w, anchor)
// positions, the nms_pre best by score (torch.topk; a level with <= nms_pre positions keeps all of them and nms() sorts them:
// the same descending list), their anchors + deltas -> delta2bbox -> [count][5] boxes in descending score order. Two launches
// for ALL levels instead of a host loop of torch kernels per level (sigmoid, topk, three gathers, decode):
//   rpn_score_kernel (the whole chip: the 0.5 M exponentials are the expensive part): key = bits of the fp32 score (all positive:
//      unsigned order = float order) of every position -> scratch, and a 4096-bin histogram of the keys' top 12 bits per level
//      (LDS-private per block, non-empty bins added to the level's global histogram);
//   rpn_select_kernel (one workgroup per level): radix select of the count-th largest key - the first level of the search reads
//      the global histogram, further 10-bit levels re-scan the level's keys, stopping as soon as everything at or above the
//      threshold bin fits the sort buffer; those candidates -> 64-bit keys (score bits | ~index) -> bitonic sort, descending: equal
//      scores in ascending index order (a stable descending sort; the reference's own order among exactly equal scores is
//      torch.topk's, unspecified); the first `count` are decoded (anchor = rounded base anchor + stride * (x, y), made on the fly).
// ------------------------------------------------------------------------------------------------
struct RpnLevels {
    const float* cls[8];
    const float* reg[8];
    int cls_ld[8], reg_ld[8], H[8], W[8];
    int koff[8];                // first key of the level in the scratch array
    float stride[8];
};
constexpr int RPN_BINS = 4096;
constexpr int RPN_UNR = 8;       // keys in flight per thread in the select kernel's scans

__global__ __launch_bounds__(256)
void rpn_score_kernel(const RpnLevels L, const int A, unsigned* __restrict__ keys, int* __restrict__ hist) {
    __shared__ int h[RPN_BINS];
    const int l = blockIdx.y;
    const int npos = L.H[l] * L.W[l];
    if ((int)blockIdx.x * 256 >= npos) return;
    for (int b = threadIdx.x; b < RPN_BINS; b += 256) h[b] = 0;
    __syncthreads();
    const float* __restrict__ cls = L.cls[l];
    const int ld = L.cls_ld[l];
    unsigned* __restrict__ kl = keys + L.koff[l];
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npos; p += gridDim.x * 256) {
        const float* row = cls + (size_t)p * ld;
        for (int a = 0; a < A; ++a) {
            const float s = 1.f / (1.f + expf(-row[a]));          // at::sigmoid on the device: 1 / (1 + exp(-x)) in fp32
            const unsigned k = __float_as_uint(s);
            kl[(size_t)p * A + a] = k;
            atomicAdd(&h[k >> 20], 1);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < RPN_BINS; b += 256)
        if (h[b]) atomicAdd(&hist[l * RPN_BINS + b], h[b]);
}

__global__ __launch_bounds__(1024)
void rpn_select_kernel(const RpnLevels L, const int A, const float* __restrict__ base_anchors, const int nms_pre, const float sx,
                       const float sy, const float sw, const float sh, const float img_h, const float img_w, const float max_ratio,
                       const unsigned* __restrict__ keys_all, int* __restrict__ hist_all, float* __restrict__ boxes) {
    __shared__ u64 keys[SORT_CAP];
    __shared__ int hist[RPN_BINS];
    __shared__ int scan[1024];
    __shared__ int sh_bin, sh_above, sh_cnt;
    const int l = blockIdx.x, t = threadIdx.x;
    const int n = L.H[l] * L.W[l] * A;
    const unsigned* __restrict__ kin = keys_all + L.koff[l];
    const int count = min(n, nms_pre);
    float* __restrict__ out = boxes + (size_t)l * nms_pre * 5;

    // ---- 1. threshold: keys whose bits above `shift` equal `prefix` are still undecided; `above` keys are known to be larger
    unsigned prefix = 0;
    int shift = 20, nbits = 12, above = 0;
    bool exact = false;          // the threshold key is known exactly (shift == 0 reached)
    bool first = true;
    // the level's global histogram is consumed here and left zero for the next launch
    for (int b = t; b < RPN_BINS; b += 1024) { hist[b] = hist_all[l * RPN_BINS + b]; hist_all[l * RPN_BINS + b] = 0; }
    __syncthreads();
    if (n > count) {
        for (;;) {
            const int nbin = 1 << nbits;
            if (!first) {
                for (int b = t; b < nbin; b += 1024) hist[b] = 0;
                __syncthreads();
                const int hs = shift + nbits;                  // bits >= hs are decided
                // RPN_UNR independent loads per thread and trip (clamped index, masked afterwards): with one load per trip the 384
                // trips of the largest level were 384 L2 round trips - 60..100 us per pass, most of the kernel's 213 us (r04 kernel stats)
                for (int c0 = 0; c0 < n; c0 += 1024 * RPN_UNR) {
                    unsigned kk[RPN_UNR];
#pragma unroll
                    for (int j = 0; j < RPN_UNR; ++j) kk[j] = kin[min(c0 + j * 1024 + t, n - 1)];
#pragma unroll
                    for (int j = 0; j < RPN_UNR; ++j)
                        if (c0 + j * 1024 + t < n && (kk[j] >> hs) == (prefix >> hs)) atomicAdd(&hist[(kk[j] >> shift) & (nbin - 1)], 1);
                }
                __syncthreads();
            }
            first = false;
            // bins from the top: thread t owns `per` consecutive bins (4 or 1)
            const int per = nbin >> 10;
            int own = 0;
            for (int e = 0; e < per; ++e) own += hist[nbin - 1 - (t * per + e)];
            scan[t] = own;
            __syncthreads();
            for (int off = 1; off < 1024; off <<= 1) {
                const int v = t >= off ? scan[t - off] : 0;
                __syncthreads();
                scan[t] += v;
                __syncthreads();
            }
            const int need = count - above;                // rank of the threshold among the undecided keys, from the top (>= 1)
            const int incl = scan[t], excl = incl - own;
            if (excl < need && need <= incl) {             // the threshold bin is one of this thread's
                int c = excl;
                for (int e = 0; e < per; ++e) {
                    const int b = nbin - 1 - (t * per + e);
                    if (c + hist[b] >= need) { sh_bin = b; sh_above = c; sh_cnt = hist[b]; break; }
                    c += hist[b];
                }
            }
            __syncthreads();
            const int bin = sh_bin, pop = sh_cnt;
            above += sh_above;
            prefix |= (unsigned)bin << shift;
            __syncthreads();
            if (shift == 0) { exact = true; break; }
            // stop when everything from this bin upwards is a short list to sort (a re-scan of the level's keys costs ~15 us for the
            // largest level, a bitonic pass over 8192 instead of 2048 keys more than that)
            if (above + pop <= min(SORT_CAP, max(2 * nms_pre, 2048))) break;
            shift -= 10; nbits = 10;
        }
    } else {
        shift = 32;                                        // every key is a candidate
    }

    // ---- 2. candidates: keys >= the lower bound of the threshold bin. With an exact threshold shared by more keys than there is
    // room for (degenerate score maps), the ties are taken in ascending index order (an ordered block scan, 1024 positions a round)
    __syncthreads();
    if (t == 0) sh_cnt = 0;
    __syncthreads();
    const unsigned lo = shift >= 32 ? 0u : (prefix >> shift) << shift;
    const int room_eq = exact ? count - above : SORT_CAP;  // ties that are taken
    for (int c0 = 0; c0 < n; c0 += 1024 * RPN_UNR) {
        unsigned kk[RPN_UNR];
#pragma unroll
        for (int j = 0; j < RPN_UNR; ++j) kk[j] = kin[min(c0 + j * 1024 + t, n - 1)];
#pragma unroll
        for (int j = 0; j < RPN_UNR; ++j) {
            const int i = c0 + j * 1024 + t;
            const unsigned k = kk[j];
            if (i < n && (exact ? k > lo : k >= lo)) {
                const int slot = atomicAdd(&sh_cnt, 1);
                keys[slot] = ((u64)k << 32) | (u64)(0xFFFFFFFFu - (unsigned)i);
            }
        }
    }
    __syncthreads();
    if (exact) {
        int taken = 0;                                     // uniform across the block
        const int base0 = sh_cnt;
        for (int c0 = 0; c0 < n && taken < room_eq; c0 += 1024) {
            const int i = c0 + t;
            const unsigned k = i < n ? kin[i] : 0u;
            const int f = (i < n && k == lo) ? 1 : 0;
            scan[t] = f;
            __syncthreads();
            for (int off = 1; off < 1024; off <<= 1) {
                const int v = t >= off ? scan[t - off] : 0;
                __syncthreads();
                scan[t] += v;
                __syncthreads();
            }
            const int rank = taken + scan[t] - f;          // this tie's position among all ties, in index order
            if (f && rank < room_eq) keys[base0 + rank] = ((u64)k << 32) | (u64)(0xFFFFFFFFu - (unsigned)i);
            taken += scan[1023];
            __syncthreads();
        }
        if (t == 0) sh_cnt = base0 + min(taken, room_eq);
        __syncthreads();
    }
    const int m = sh_cnt;                                  // count <= m <= SORT_CAP
    const int n2 = next_pow2(m);
    for (int i = m + t; i < n2; i += 1024) keys[i] = 0;
    __syncthreads();
    bitonic_sort_desc<1024>(keys, n2);

    // ---- 3. decode (delta2bbox, means 0; the arithmetic of delta2bbox_kernel in det_ops.hip)
    const float* __restrict__ reg = L.reg[l];
    const int rld = L.reg_ld[l], W = L.W[l];
    const float stride = L.stride[l];
    for (int i = t; i < nms_pre; i += 1024) {
        float o[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (i < count) {
            const u64 key = keys[i];
            const int idx = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
            const int pos = idx / A, a = idx - pos * A;
            const int y = pos / W, x = pos - y * W;
            const float* ba = base_anchors + ((size_t)l * A + a) * 4;
            const float fx = (float)x * stride, fy = (float)y * stride;
            const float a0 = ba[0] + fx, a1 = ba[1] + fy, a2 = ba[2] + fx, a3 = ba[3] + fy;
            const float* d = reg + (size_t)pos * rld + 4 * a;
            const float dx = d[0] * sx, dy = d[1] * sy;
            float dw = d[2] * sw, dh = d[3] * sh;
            dw = fminf(fmaxf(dw, -max_ratio), max_ratio);
            dh = fminf(fmaxf(dh, -max_ratio), max_ratio);
            const float px = (a0 + a2) * 0.5f, py = (a1 + a3) * 0.5f;
            const float pw = a2 - a0 + 1.0f, ph = a3 - a1 + 1.0f;
            const float gw = pw * expf(dw), gh = ph * expf(dh);
            const float gx = __fadd_rn(px, __fmul_rn(pw, dx)), gy = __fadd_rn(py, __fmul_rn(ph, dy));
            float x1 = gx - gw * 0.5f + 0.5f, y1 = gy - gh * 0.5f + 0.5f;
            float x2 = gx + gw * 0.5f - 0.5f, y2 = gy + gh * 0.5f - 0.5f;
            o[0] = fminf(fmaxf(x1, 0.f), img_w - 1.f); o[1] = fminf(fmaxf(y1, 0.f), img_h - 1.f);
            o[2] = fminf(fmaxf(x2, 0.f), img_w - 1.f); o[3] = fminf(fmaxf(y2, 0.f), img_h - 1.f);
            o[4] = __uint_as_float((unsigned)(key >> 32));
        }
        for (int e = 0; e < 5; ++e) out[(size_t)i * 5 + e] = o[e];
    }
}

// ------------------------------------------------------------------------------------------------
// MaskROI, first half (mask_roi.py:43-95 with class_agnostic=True, clip_boxes=True, bbox_class_agnostic=False):
// candidate q = roi*(nc-1) + (cls-1); those with prob > thr are sorted by descending score (gpu_nms.pyx:23-38:
// `order = scores.argsort()[::-1]`; equal scores: the larger q first = a stable ascending sort reversed) and their refined,
// clipped boxes are written as the [m][5] list the NMS takes. fp32 arithmetic in the reference's order (numpy float32,
// no fused multiply-add), exp correctly rounded.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void maskroi_select_kernel(const float* __restrict__ rois, const float* __restrict__ delta, const float* __restrict__ prob, int n,
                           const int* __restrict__ n_valid, int nc, float thr, float wx, float wy, float ww, float wh, float im_h,
                           float im_w, float* __restrict__ dets, int* __restrict__ cand, int* __restrict__ m_out) {
    __shared__ u64 keys[SORT_CAP];
    __shared__ int count;
    if (threadIdx.x == 0) count = 0;
    __syncthreads();
    const int nv = n_valid ? min(*n_valid, n) : n;
    const int total = nv * (nc - 1);
    for (int q = threadIdx.x; q < total; q += 1024) {
        const int r = q / (nc - 1), c = q - r * (nc - 1) + 1;
        const float p = prob[(size_t)r * nc + c];
        if (p > thr) {
            const int slot = atomicAdd(&count, 1);
            if (slot < SORT_CAP) keys[slot] = ((u64)__float_as_uint(p) << 32) | (u64)(unsigned)q;
        }
    }
    __syncthreads();
    const int m = min(count, SORT_CAP);
    const int n2 = next_pow2(m);
    for (int i = m + threadIdx.x; i < n2; i += 1024) keys[i] = 0;
    __syncthreads();
    bitonic_sort_desc<1024>(keys, n2);
    const float clipw = 4.1351666f;      // float32(np.log(1000. / 16.)) (bbox_transform.py:312-313)
    for (int i = threadIdx.x; i < m; i += 1024) {
        const int q = (int)(keys[i] & 0xFFFFFFFFull);
        const float score = __uint_as_float((unsigned)(keys[i] >> 32));
        const int r = q / (nc - 1), c = q - r * (nc - 1) + 1;
        const float x1 = rois[(size_t)r * 5 + 1], y1 = rois[(size_t)r * 5 + 2], x2 = rois[(size_t)r * 5 + 3], y2 = rois[(size_t)r * 5 + 4];
        const float* d = delta + (size_t)r * 4 * nc + 4 * c;
        // bbox_transform.py:290-330
        const float w = __fadd_rn(__fsub_rn(x2, x1), 1.f), h = __fadd_rn(__fsub_rn(y2, y1), 1.f);
        const float cx = __fadd_rn(x1, __fmul_rn(0.5f, w)), cy = __fadd_rn(y1, __fmul_rn(0.5f, h));
        const float dx = __fdiv_rn(d[0], wx), dy = __fdiv_rn(d[1], wy);
        const float dw = fminf(__fdiv_rn(d[2], ww), clipw), dh = fminf(__fdiv_rn(d[3], wh), clipw);
        const float pcx = __fadd_rn(__fmul_rn(dx, w), cx), pcy = __fadd_rn(__fmul_rn(dy, h), cy);
        const float pw = __fmul_rn((float)exp((double)dw), w), ph = __fmul_rn((float)exp((double)dh), h);
        float bx1 = __fsub_rn(pcx, __fmul_rn(0.5f, pw)), by1 = __fsub_rn(pcy, __fmul_rn(0.5f, ph));
        float bx2 = __fsub_rn(__fadd_rn(pcx, __fmul_rn(0.5f, pw)), 1.f), by2 = __fsub_rn(__fadd_rn(pcy, __fmul_rn(0.5f, ph)), 1.f);
        // clip_boxes (bbox_transform.py:45-60)
        bx1 = fmaxf(fminf(bx1, im_w - 1.f), 0.f); by1 = fmaxf(fminf(by1, im_h - 1.f), 0.f);
        bx2 = fmaxf(fminf(bx2, im_w - 1.f), 0.f); by2 = fmaxf(fminf(by2, im_h - 1.f), 0.f);
        float* o = dets + (size_t)i * 5;
        o[0] = bx1; o[1] = by1; o[2] = bx2; o[3] = by2; o[4] = score;
        cand[i] = q;
    }
    if (threadIdx.x == 0) { m_out[0] = m; m_out[1] = count > SORT_CAP; m_out[2] = nv; }
}

// ------------------------------------------------------------------------------------------------
// MaskROI, second half (mask_roi.py:96-147): the post-NMS list in keep order, cut to the max_det best BY VALUE (every
// detection whose score equals the max_det-th best stays: `>= image_thresh`), or the dummy row (score 1, box 0, class 0).
// res: [8] header (K, candidates, post-NMS count, status: bit 0 = more than SORT_CAP candidates, bit 1 = more than kcap rows,
// number of valid rois)
// then K rows of 8 floats (0, x1, y1, x2, y2, score, class, candidate index).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void maskroi_finish_kernel(const float* __restrict__ dets, const int* __restrict__ cand, const int* __restrict__ m_in,
                           const int* __restrict__ keep, const int* __restrict__ nkeep, int nc, int max_det, int kcap,
                           float* __restrict__ res) {
    const int nk = *nkeep;
    int K = nk;
    if (max_det > 0 && nk > max_det) {
        const float th = dets[(size_t)keep[max_det - 1] * 5 + 4];        // np.sort(scores)[-max_det]: the list is in descending order
        K = max_det;
        while (K < nk && dets[(size_t)keep[K] * 5 + 4] >= th) ++K;
    }
    int status = m_in[1] ? 1 : 0;
    if (K > kcap) { status |= 2; K = kcap; }
    if (nk == 0) {
        if (threadIdx.x < 8) res[8 + threadIdx.x] = threadIdx.x == 5 ? 1.f : 0.f;   // scores = 1, boxes = 0, cls_idx = 0 (mask_roi.py:136-142)
        K = 1;
    } else {
        for (int i = threadIdx.x; i < K; i += 256) {
            const int j = keep[i];
            const int q = cand[j];
            float* o = res + 8 + (size_t)i * 8;
            o[0] = 0.f;
            o[1] = dets[(size_t)j * 5 + 0]; o[2] = dets[(size_t)j * 5 + 1]; o[3] = dets[(size_t)j * 5 + 2]; o[4] = dets[(size_t)j * 5 + 3];
            o[5] = dets[(size_t)j * 5 + 4];
            o[6] = (float)(q % (nc - 1) + 1);
            o[7] = (float)q;
        }
    }
    if (threadIdx.x == 0) { res[0] = (float)K; res[1] = (float)m_in[0]; res[2] = (float)nk; res[3] = (float)status; res[4] = (float)m_in[2]; }
}

// ------------------------------------------------------------------------------------------------
// Tracking block (panoptic_fusetrack.py:424-469). comp [K][M+1] comprehensive scores (column 0 = "new object").
// Row arg-max (first maximum), then the reference's sequential loops on one thread: a detection whose best column is 0
// gets a new id; otherwise it takes memory entry obj = column-1 if its score beats the best one seen for obj so far (the
// previous holder is undone -> -1 and receives a new id in the second loop). Memory update as the reference's in-place
// writes leave it: entry obj holds the LAST detection assigned to it, new entries are appended in assignment order.
// emb [K][E], box [K][ldb] (first 4 used), label [K] int64; prev_* have room for M + K rows. scratch: int32 [M + 3K + 1].
// out: ids [K] int32, m_out[0] = new M.
// ------------------------------------------------------------------------------------------------
// row arg-max (FIRST maximum, like torch.max / numpy argmax), one workgroup per detection: the columns over the threads (coalesced,
// four loads in flight), (value, column) reduced with "greater value, or equal value and smaller column". One thread per row, as this
// was inside track_assign_kernel, walked a row of M + 1 floats with one dependent L2 round trip per element: 0.29 ms per frame once the
// memory holds a few thousand entries (r04 kernel stats).
__global__ __launch_bounds__(256)
void track_argmax_kernel(const float* __restrict__ comp, int M, int* __restrict__ mi, float* __restrict__ ml) {
    const int i = blockIdx.x;
    const float* __restrict__ row = comp + (size_t)i * (M + 1);
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c0 = 0; c0 <= M; c0 += 1024) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = row[min(c0 + u * 256 + (int)threadIdx.x, M)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = c0 + u * 256 + threadIdx.x;
            if (j <= M && (v[u] > best || (v[u] == best && j < bi))) { best = v[u]; bi = j; }
        }
    }
    // NaN scores never win a comparison: a row of NaNs keeps column 0x7fffffff here and column 0 in the reference's loop
    if (bi == 0x7fffffff) { bi = 0; best = row[0]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oj = __shfl_xor(bi, off, 64);
        if (ov > best || (ov == best && oj < bi)) { best = ov; bi = oj; }
    }
    __shared__ float sv[4];
    __shared__ int sj[4];
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; sj[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && sj[w] < bi)) { best = sv[w]; bi = sj[w]; }
        const float r0 = row[0];
        if (r0 != r0) { best = r0; bi = 0; }          // the reference's loop starts from row[0]: a NaN there is never beaten
        mi[i] = bi;
        ml[i] = best;
    }
}

__global__ __launch_bounds__(256)
void track_assign_kernel(const float* __restrict__ comp, int K, int M, int* __restrict__ scratch, int* __restrict__ ids, int* __restrict__ m_out) {
    int* setsrc = scratch;            // [M] detection written into entry obj, -1 = untouched
    int* addlist = scratch + M;       // [K] detections appended, in order
    int* mi = scratch + M + K;        // [K] arg-max column (track_argmax_kernel), then the id of the detection
    float* ml = reinterpret_cast<float*>(scratch + M + 2 * K);   // [K] likelihood of the arg-max
    __shared__ int nadd;
    for (int i = threadIdx.x; i < M; i += 256) setsrc[i] = -1;
    __syncthreads();
    if (threadIdx.x == 0) {
        // best_match_scores / best_match_ids of the reference live in prev-indexed scratch: reuse setsrc for the ids and keep the
        // scores in registers-by-lookup (ml of the current holder)
        int mem = M, na = 0;
        for (int i = 0; i < K; ++i) {
            const int col = mi[i];
            const float like = ml[i];
            int id = -1;
            if (col == 0) {
                id = mem++; addlist[na++] = i;
            } else {
                const int obj = col - 1;
                const int holder = setsrc[obj];                       // best_match_ids[obj]
                const float held = holder >= 0 ? ml[holder] : -100.f;  // best_match_scores[obj]
                if (like > held) {
                    id = obj;
                    if (holder >= 0) mi[holder] = -1;                 // undo: det_obj_ids[holder] = -1
                    setsrc[obj] = i;
                }
            }
            // ids are kept in mi from here on (col is consumed): >= 0 id, -1 unassigned
            mi[i] = id;
        }
        for (int i = 0; i < K; ++i)
            if (mi[i] < 0) { mi[i] = mem++; addlist[na++] = i; }
        nadd = na;
        m_out[0] = mem;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += 256) ids[i] = mi[i];
    if (threadIdx.x == 0) scratch[M + 3 * K] = nadd;
}

// memory update of the tracking block, one workgroup per touched entry: entry obj < M takes the LAST detection assigned to it,
// entries M .. M+nadd-1 are the appended detections in assignment order
__global__ __launch_bounds__(256)
void track_update_kernel(int K, int M, const float* __restrict__ emb, int E, const float* __restrict__ box, int ldb,
                         const long long* __restrict__ label, float* __restrict__ prev_emb, float* __restrict__ prev_box,
                         long long* __restrict__ prev_label, const int* __restrict__ scratch) {
    const int* setsrc = scratch;
    const int* addlist = scratch + M;
    const int na = scratch[M + 3 * K];
    for (int dst = blockIdx.x; dst < M + na; dst += gridDim.x) {
        const int src = dst < M ? setsrc[dst] : addlist[dst - M];
        if (src < 0) continue;
        for (int e = threadIdx.x; e < E; e += 256) prev_emb[(size_t)dst * E + e] = emb[(size_t)src * E + e];
        if (threadIdx.x < 4) prev_box[(size_t)dst * 4 + threadIdx.x] = box[(size_t)src * ldb + threadIdx.x];
        if (dst >= M && threadIdx.x == 0) prev_label[dst] = label[src];
    }
}

// ------------------------------------------------------------------------------------------------
// Kept list of MaskRemoval + instance table of the combine kernel. order [n]: detection visited at position p of the
// descending-score walk (mask_removal.py:49), flags [n]: kept at position p, rows [n][8] as written by maskroi_finish,
// tbox [n][4]: int32-truncated boxes in walk order. Nothing kept -> keep = [0] with an empty paste region
// (mask_removal.py:89-91). SegTerm crop (unary_logits.py:96-106 on boxes*4*0.25 == boxes): [int(x1), int(round(x2)+1)).
// ------------------------------------------------------------------------------------------------
struct ClassMap { int v[16]; };

__global__ __launch_bounds__(64)
void pan_instances_kernel(const int* __restrict__ order, const int* __restrict__ flags, const float* __restrict__ rows,
                          const int* __restrict__ tbox, int n, ClassMap cm, vps_pan_inst* __restrict__ inst, int* __restrict__ keep_out,
                          int* __restrict__ k_out) {
    if (threadIdx.x != 0) return;
    int k = 0;
    for (int p = 0; p < n; ++p) {
        if (!flags || !flags[p]) continue;
        const int i = order[p];
        const float* r = rows + (size_t)i * 8;
        vps_pan_inst it;
        const int c = (int)r[6];
        if (c == 0) { it.sx0 = it.sy0 = it.sx1 = it.sy1 = 0; it.seg_ch = 0; }
        else {
            it.sx0 = (int)r[1]; it.sy0 = (int)r[2];
            it.sx1 = (int)(rintf(r[3]) + 1.f); it.sy1 = (int)(rintf(r[4]) + 1.f);
            it.seg_ch = cm.v[c];
        }
        it.bx1 = tbox[p * 4 + 0]; it.by1 = tbox[p * 4 + 1]; it.bx2 = tbox[p * 4 + 2]; it.by2 = tbox[p * 4 + 3];
        it.mask_idx = i;
        inst[k] = it;
        keep_out[k] = i;
        ++k;
    }
    if (k == 0) {
        const float* r = rows;
        vps_pan_inst it;
        const int c = (int)r[6];
        if (c == 0) { it.sx0 = it.sy0 = it.sx1 = it.sy1 = 0; it.seg_ch = 0; }
        else {
            it.sx0 = (int)r[1]; it.sy0 = (int)r[2];
            it.sx1 = (int)(rintf(r[3]) + 1.f); it.sy1 = (int)(rintf(r[4]) + 1.f);
            it.seg_ch = cm.v[c];
        }
        it.bx1 = 0; it.by1 = 0; it.bx2 = -1; it.by2 = -1;     // empty paste region: the mask logits are all zero
        it.mask_idx = 0;
        inst[0] = it;
        keep_out[0] = 0;
        k = 1;
        k_out[1] = 0;      // masks_valid
    } else {
        k_out[1] = 1;
    }
    k_out[0] = k;
}

// ------------------------------------------------------------------------------------------------
// The frame's end-of-frame record in ONE launch (round 5; was six tiny copy / reduce launches behind the combine kernel, on an
// otherwise idle GPU): tail = [k, masks_valid, status, -, mem_count, f16 range report, -, -, keep[kcap], ids[kcap]] (int32).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void frame_tail_kernel(const int* __restrict__ kinfo, const int* __restrict__ keep, const int* __restrict__ ids, const int* __restrict__ mem_count,
                       const int* __restrict__ f16_status, int nslots, int K, int kcap, int* __restrict__ tail) {
    __shared__ int red[256];
    int m = 0;
    for (int i = threadIdx.x; i < nslots; i += 256) m = max(m, f16_status[i]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x < 4) tail[threadIdx.x] = kinfo[threadIdx.x];
    if (threadIdx.x == 4) tail[4] = mem_count ? mem_count[0] : 0;
    if (threadIdx.x == 5) tail[5] = f16_status ? red[0] : 0;
    for (int i = threadIdx.x; i < K; i += 256) {
        tail[8 + i] = keep[i];
        if (ids) tail[8 + kcap + i] = ids[i];
    }
}

}  // namespace

extern "C" int vps_frame_tail(const int32_t* kinfo, const int32_t* keep, const int32_t* ids, const int32_t* mem_count, const int32_t* f16_status,
                              int nslots, int K, int kcap, int32_t* tail, void* stream) {
    if (!kinfo || !keep || !tail || K < 0 || K > kcap || nslots < 0 || (nslots > 0 && !f16_status)) return VPS_EARG(1);
    hipLaunchKernelGGL(frame_tail_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, kinfo, keep, ids, mem_count, nslots > 0 ? f16_status : nullptr,
                       nslots, K, kcap, tail);
    return vps_launch_status();
}

extern "C" int vps_rpn_collect(const float* boxes, const int32_t* keep, const int32_t* nkeep, int nlv, int nmax, int nms_post,
                               int max_num, float* out, int32_t* n_out, void* stream) {
    if (!boxes || !keep || !nkeep || !out || !n_out) return VPS_EARG(1);
    if (nlv < 1 || nlv > 8 || nmax < 1 || nms_post < 1 || max_num < 1 || (long)nlv * min(nms_post, nmax) > SORT_CAP) return VPS_EARG(2);
    hipLaunchKernelGGL(rpn_collect_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, boxes, keep, nkeep, nlv, nmax, nms_post, max_num, out, n_out);
    return vps_launch_status();
}

extern "C" int vps_rpn_select(const float* const* cls, const int32_t* cls_ld, const float* const* reg, const int32_t* reg_ld,
                              const int32_t* Hs, const int32_t* Ws, const float* strides, int nlv, int A, const float* base_anchors,
                              int nms_pre, const float* stds, float img_h, float img_w, uint32_t* keys, int32_t* hist, float* boxes,
                              void* stream) {
    if (!cls || !cls_ld || !reg || !reg_ld || !Hs || !Ws || !strides || !base_anchors || !stds || !keys || !hist || !boxes) return VPS_EARG(1);
    if (nlv < 1 || nlv > 8 || A < 1 || nms_pre < 1 || nms_pre > SORT_CAP) return VPS_EARG(2);
    RpnLevels L;
    long off = 0, maxpos = 0;
    for (int l = 0; l < nlv; ++l) {
        if (!cls[l] || !reg[l] || Hs[l] < 1 || Ws[l] < 1 || cls_ld[l] < A || reg_ld[l] < 4 * A) return VPS_EARG(3);
        L.cls[l] = cls[l]; L.reg[l] = reg[l]; L.cls_ld[l] = cls_ld[l]; L.reg_ld[l] = reg_ld[l]; L.H[l] = Hs[l]; L.W[l] = Ws[l]; L.stride[l] = strides[l];
        L.koff[l] = (int)off;
        off += (long)Hs[l] * Ws[l] * A;
        if (off > 0x7fffffffL) return VPS_EARG(3);
        if ((long)Hs[l] * Ws[l] > maxpos) maxpos = (long)Hs[l] * Ws[l];
    }
    const float max_ratio = (float)fabs(log(16.0 / 1000.0));
    long gx = (maxpos + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(rpn_score_kernel, dim3((unsigned)gx, (unsigned)nlv), dim3(256), 0, (hipStream_t)stream, L, A, keys, hist);
    hipLaunchKernelGGL(rpn_select_kernel, dim3(nlv), dim3(1024), 0, (hipStream_t)stream, L, A, base_anchors, nms_pre, stds[0], stds[1], stds[2],
                       stds[3], img_h, img_w, max_ratio, keys, hist, boxes);
    return vps_launch_status();
}

extern "C" int vps_maskroi_select(const float* rois, const float* bbox_delta, const float* cls_prob, int n, const int32_t* n_valid,
                                  int num_classes, float score_thresh, const float* reg_weights, float im_h, float im_w, float* dets,
                                  int32_t* cand, int32_t* m_out, void* stream) {
    if (!rois || !bbox_delta || !cls_prob || !dets || !cand || !m_out || !reg_weights) return VPS_EARG(1);
    if (n < 1 || num_classes < 2 || (long)n * (num_classes - 1) > 0x7fffffffL) return VPS_EARG(2);
    hipLaunchKernelGGL(maskroi_select_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rois, bbox_delta, cls_prob, n, n_valid, num_classes,
                       score_thresh, reg_weights[0], reg_weights[1], reg_weights[2], reg_weights[3], im_h, im_w, dets, cand, m_out);
    return vps_launch_status();
}

extern "C" int vps_maskroi_finish(const float* dets, const int32_t* cand, const int32_t* m_in, const int32_t* keep, const int32_t* nkeep,
                                  int num_classes, int max_det, int kcap, float* res, void* stream) {
    if (!dets || !cand || !m_in || !keep || !nkeep || !res) return VPS_EARG(1);
    if (num_classes < 2 || kcap < 1) return VPS_EARG(2);
    hipLaunchKernelGGL(maskroi_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dets, cand, m_in, keep, nkeep, num_classes, max_det, kcap, res);
    return vps_launch_status();
}

extern "C" int vps_track_assign(const float* comp, int K, int M, const float* emb, int E, const float* box, int ldb, const int64_t* label,
                                float* prev_emb, float* prev_box, int64_t* prev_label, int32_t* scratch, int32_t* ids, int32_t* m_out,
                                void* stream) {
    if (!comp || !emb || !box || !label || !prev_emb || !prev_box || !prev_label || !scratch || !ids || !m_out) return VPS_EARG(1);
    if (K < 1 || M < 1 || E < 1 || ldb < 4) return VPS_EARG(2);
    // the sequential assignment on ONE workgroup, then the memory update spread over the touched entries
    hipLaunchKernelGGL(track_argmax_kernel, dim3((unsigned)K), dim3(256), 0, (hipStream_t)stream, comp, M, scratch + M + K,
                       reinterpret_cast<float*>(scratch + M + 2 * K));
    hipLaunchKernelGGL(track_assign_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, comp, K, M, scratch, ids, m_out);
    hipLaunchKernelGGL(track_update_kernel, dim3((unsigned)min(M + K, 2048)), dim3(256), 0, (hipStream_t)stream, K, M, emb, E, box, ldb,
                       reinterpret_cast<const long long*>(label), prev_emb, prev_box, reinterpret_cast<long long*>(prev_label), scratch);
    return vps_launch_status();
}

extern "C" int vps_pan_instances(const int32_t* order, const int32_t* flags, const float* rows, const int32_t* tbox, int n,
                                 const int32_t* class_mapping, int num_classes, vps_pan_inst* inst, int32_t* keep_out, int32_t* k_out,
                                 void* stream) {
    if (!rows || !inst || !keep_out || !k_out || !class_mapping) return VPS_EARG(1);
    if (n < 1 || num_classes < 1 || num_classes > 16 || (n > 0 && flags && (!order || !tbox))) return VPS_EARG(2);
    ClassMap cm;
    for (int i = 0; i < 16; ++i) cm.v[i] = i < num_classes ? class_mapping[i] : 0;
    hipLaunchKernelGGL(pan_instances_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, order, flags, rows, tbox, n, cm, inst, keep_out, k_out);
    return vps_launch_status();
}
