// Detection-side kernels: multi-level RoIAlign (NHWC), NMS (bitmask + on-device greedy reduce),
// box decode, IoU matrix, row softmax. Reference files relative to /root/reference/mmdet.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ------------------------------------------------------------------------------------------------
// ops/roi_align/src/roi_align_kernel.cu:16-124 + models/roi_extractors/single_level.py:54-107.
// One launch for all FPN levels: the level of each RoI is derived in-kernel, the output is NHWC
// [R][P][P][C] and threads run fastest over channels (float4), so each bilinear corner is a coalesced read.
// ------------------------------------------------------------------------------------------------
struct RoiLevels {
    const float* feat[4];
    int ld[4], H[4], W[4];
    float scale[4];
    int n;
};

__device__ __forceinline__ f32x4 roi_bilinear(const float* __restrict__ base, int ld, int height, int width, float y, float x) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) return z;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
    if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
    const float ly = y - (float)y_low, lx = x - (float)x_low;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const f32x4 lt = *reinterpret_cast<const f32x4*>(base + ((size_t)y_low * width + x_low) * ld);
    const f32x4 rt = *reinterpret_cast<const f32x4*>(base + ((size_t)y_low * width + x_high) * ld);
    const f32x4 lb = *reinterpret_cast<const f32x4*>(base + ((size_t)y_high * width + x_low) * ld);
    const f32x4 rb = *reinterpret_cast<const f32x4*>(base + ((size_t)y_high * width + x_high) * ld);
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    return w1 * lt + w2 * rt + w3 * lb + w4 * rb;
}

__global__ __launch_bounds__(256)
void roi_align_kernel(RoiLevels L, float finest_scale, const float* __restrict__ rois, int R, int C, int P, int sn,
                      float* __restrict__ out) {
    const int c4n = C >> 2;
    const long total = (long)R * P * P * c4n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        long t = idx / c4n;
        const int pw = (int)(t % P); t /= P;
        const int ph = (int)(t % P);
        const int r = (int)(t / P);
        const float* roi = rois + (size_t)r * 5;
        const int b = (int)roi[0];
        const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
        // map_roi_levels
        const float scale = sqrtf((x2 - x1 + 1.f) * (y2 - y1 + 1.f));
        int lvl = (int)floorf(log2f(scale / finest_scale + 1e-6f));
        lvl = max(0, min(lvl, L.n - 1));
        const float* feat; int ld, H, W; float ss;
        // (static selects keep the struct in SGPRs instead of scratch)
        if (lvl == 0) { feat = L.feat[0]; ld = L.ld[0]; H = L.H[0]; W = L.W[0]; ss = L.scale[0]; }
        else if (lvl == 1) { feat = L.feat[1]; ld = L.ld[1]; H = L.H[1]; W = L.W[1]; ss = L.scale[1]; }
        else if (lvl == 2) { feat = L.feat[2]; ld = L.ld[2]; H = L.H[2]; W = L.W[2]; ss = L.scale[2]; }
        else { feat = L.feat[3]; ld = L.ld[3]; H = L.H[3]; W = L.W[3]; ss = L.scale[3]; }
        const float roi_start_w = x1 * ss, roi_start_h = y1 * ss;
        const float roi_end_w = (x2 + 1.f) * ss, roi_end_h = (y2 + 1.f) * ss;
        const float roi_width = fmaxf(roi_end_w - roi_start_w, 0.f);
        const float roi_height = fmaxf(roi_end_h - roi_start_h, 0.f);
        const float bin_h = roi_height / (float)P, bin_w = roi_width / (float)P;
        const float* base = feat + (size_t)b * H * W * ld + 4 * c4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int iy = 0; iy < sn; ++iy) {
            const float y = roi_start_h + (float)ph * bin_h + ((float)iy + .5f) * bin_h / (float)sn;
            for (int ix = 0; ix < sn; ++ix) {
                const float x = roi_start_w + (float)pw * bin_w + ((float)ix + .5f) * bin_w / (float)sn;
                acc += roi_bilinear(base, ld, H, W, y, x);
            }
        }
        acc = acc / (float)(sn * sn);
        *reinterpret_cast<f32x4*>(out + (((size_t)r * P + ph) * P + pw) * C + 4 * c4) = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// Suppression bitmask of NMS. Same result as ops/nms/src/nms_kernel.cu:13-67 / utils/upsnet/nms/nms_kernel.cu:40-84 (word
// [row i][column block c] has bit j set iff IoU(box i, box 64c+j) > thr, +1 pixel convention, strict '>', j > i on the
// diagonal block), built the wave64 way: one wavefront per tile of the UPPER triangle only (the greedy pass never reads a
// word left of the diagonal), lane = column box held in registers; the 64 row boxes are broadcast lane by lane and the 64-bit
// word of a row IS the wave's ballot of "column lane is suppressed by this row". The IoU keeps the reference's operation order
// (interS / (Sa + Sb - interS), fp32) so decisions at the threshold are the same.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64)
void nms_mask_kernel(const float* __restrict__ boxes_all, int nmax, const int* __restrict__ counts, float thr,
                     unsigned long long* __restrict__ mask_all, int col_blocks_max) {
    const int batch = blockIdx.z;
    const int n = counts[batch];
    const int col_blocks = (n + 63) / 64;
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb || cb >= col_blocks) return;
    const float* boxes = boxes_all + (size_t)batch * nmax * 5;
    unsigned long long* mask = mask_all + (size_t)batch * nmax * col_blocks_max;
    const int lane = threadIdx.x;
    const int row_size = min(n - rb * 64, 64);
    const int ci = cb * 64 + lane, ri = rb * 64 + lane;
    const bool cvalid = ci < n;
    // column box of this lane and (for the broadcasts) row box of this lane; clamped loads, masked by cvalid / row_size
    const float* cp = boxes + (size_t)min(ci, n - 1) * 5;
    const float c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
    const float Sb = (c2 - c0 + 1.f) * (c3 - c1 + 1.f);
    const float* rp = boxes + (size_t)min(ri, n - 1) * 5;
    const float r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
    unsigned long long mine = 0ULL;
    for (int i = 0; i < row_size; ++i) {
        const float a0 = __shfl(r0, i, 64), a1 = __shfl(r1, i, 64), a2 = __shfl(r2, i, 64), a3 = __shfl(r3, i, 64);
        const float left = fmaxf(a0, c0), right = fminf(a2, c2);
        const float top = fmaxf(a1, c1), bottom = fminf(a3, c3);
        const float width = fmaxf(right - left + 1.f, 0.f), height = fmaxf(bottom - top + 1.f, 0.f);
        const float interS = width * height;
        const float Sa = (a2 - a0 + 1.f) * (a3 - a1 + 1.f);
        const bool sup = cvalid && (rb != cb || lane > i) && (interS / (Sa + Sb - interS) > thr);
        const unsigned long long word = __ballot(sup);
        if (lane == i) mine = word;
    }
    if (lane < row_size) mask[(size_t)ri * col_blocks_max + cb] = mine;
}

// greedy reduce of the bitmask (the host loop of nms_kernel.cu:99-123) on the device: one wavefront per
// batch entry; 64 boxes at a time — the intra-block chain is resolved with wave shuffles on the diagonal
// tile, then the kept rows are OR-ed into the running suppression words (lanes parallel over columns).
__global__ __launch_bounds__(256)
void nms_reduce_kernel(const unsigned long long* __restrict__ mask_all, int nmax, const int* __restrict__ counts,
                       int col_blocks_max, int* __restrict__ keep_all, int* __restrict__ nkeep) {
    // Greedy pass over the suppression matrix, 64 boxes at a time. The chain through the blocks is sequential; inside a block
    // wave 0 resolves the 64 x 64 diagonal tile with shuffles, then ALL 256 threads spread the kept rows over the later column
    // words: thread = (word j, 8 of the 64 rows), its <= 8 row words requested together (branch-free), OR-ed into the word in LDS.
    // (One wavefront walking <= 64 dependent loads per block was latency bound: 117 us for 2000 boxes.)
    extern __shared__ unsigned long long remv[];  // col_blocks_max words
    __shared__ unsigned long long kept_s;
    const int batch = blockIdx.x;
    const int n = counts[batch];
    const int col_blocks = (n + 63) / 64;
    const unsigned long long* mask = mask_all + (size_t)batch * nmax * col_blocks_max;
    int* keep = keep_all + (size_t)batch * nmax;
    const int t = threadIdx.x, lane = t & 63;
    for (int j = t; j < col_blocks; j += 256) remv[j] = 0ULL;
    __syncthreads();
    int num = 0;
    for (int blk = 0; blk < col_blocks; ++blk) {
        if (t < 64) {
            const int i = blk * 64 + lane;
            const unsigned long long diag = (i < n) ? mask[(size_t)i * col_blocks_max + blk] : 0ULL;
            unsigned long long cur = remv[blk];
            const int valid = min(n - blk * 64, 64);
            unsigned long long kept = 0ULL;
            for (int l = 0; l < valid; ++l) {
                const unsigned long long row = __shfl(diag, l, 64);
                if (!((cur >> l) & 1ULL)) { kept |= 1ULL << l; cur |= row; }
            }
            // write kept indices (ascending)
            if ((kept >> lane) & 1ULL) keep[num + __popcll(kept & ((1ULL << lane) - 1ULL))] = blk * 64 + lane;
            if (lane == 0) kept_s = kept;
        }
        __syncthreads();
        const unsigned long long kept = kept_s;
        num += __popcll(kept);
        // propagate suppression to the later column blocks
        const int part = t >> 5;
        const unsigned mine = (unsigned)(kept >> (8 * part)) & 0xffu;
        for (int j = blk + 1 + (t & 31); j < col_blocks; j += 32) {
            unsigned long long v[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int l = ((mine >> b) & 1u) ? 8 * part + b : 0;         // row 0 of the block is always mapped
                v[b] = mask[(size_t)(blk * 64 + l) * col_blocks_max + j];
            }
            unsigned long long acc = 0ULL;
#pragma unroll
            for (int b = 0; b < 8; ++b) acc |= ((mine >> b) & 1u) ? v[b] : 0ULL;
            if (acc) atomicOr(&remv[j], acc);
        }
        __syncthreads();
    }
    if (t == 0) nkeep[batch] = num;
}

// core/bbox/transforms.py:34-68 delta2bbox (means 0)
__global__ __launch_bounds__(256)
void delta2bbox_kernel(const float* __restrict__ anchors, const float* __restrict__ deltas, const float* __restrict__ scores,
                       float* __restrict__ out5, int n, float sx, float sy, float sw, float sh, float img_h, float img_w,
                       float max_ratio) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* a = anchors + (size_t)i * 4;
    const float* d = deltas + (size_t)i * 4;
    const float dx = d[0] * sx, dy = d[1] * sy;
    float dw = d[2] * sw, dh = d[3] * sh;
    dw = fminf(fmaxf(dw, -max_ratio), max_ratio);
    dh = fminf(fmaxf(dh, -max_ratio), max_ratio);
    const float px = (a[0] + a[2]) * 0.5f, py = (a[1] + a[3]) * 0.5f;
    const float pw = a[2] - a[0] + 1.0f, ph = a[3] - a[1] + 1.0f;
    const float gw = pw * expf(dw), gh = ph * expf(dh);
    const float gx = __fadd_rn(px, __fmul_rn(pw, dx)), gy = __fadd_rn(py, __fmul_rn(ph, dy));
    float x1 = gx - gw * 0.5f + 0.5f, y1 = gy - gh * 0.5f + 0.5f;
    float x2 = gx + gw * 0.5f - 0.5f, y2 = gy + gh * 0.5f - 0.5f;
    x1 = fminf(fmaxf(x1, 0.f), img_w - 1.f); y1 = fminf(fmaxf(y1, 0.f), img_h - 1.f);
    x2 = fminf(fmaxf(x2, 0.f), img_w - 1.f); y2 = fminf(fmaxf(y2, 0.f), img_h - 1.f);
    float* o = out5 + (size_t)i * 5;
    o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = scores[i];
}

// core/bbox/geometry.py:4-63 (mode iou, not aligned)
__global__ __launch_bounds__(256)
void bbox_overlaps_kernel(const float* __restrict__ a, int lda, int m, const float* __restrict__ b, int ldb, int n,
                          float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * n) return;
    const int i = idx / n, j = idx - i * n;
    const float* p = a + (size_t)i * lda;
    const float* q = b + (size_t)j * ldb;
    const float lt_x = fmaxf(p[0], q[0]), lt_y = fmaxf(p[1], q[1]);
    const float rb_x = fminf(p[2], q[2]), rb_y = fminf(p[3], q[3]);
    const float w = fmaxf(rb_x - lt_x + 1.f, 0.f), h = fmaxf(rb_y - lt_y + 1.f, 0.f);
    const float overlap = w * h;
    const float area1 = (p[2] - p[0] + 1.f) * (p[3] - p[1] + 1.f);
    const float area2 = (q[2] - q[0] + 1.f) * (q[3] - q[1] + 1.f);
    out[idx] = overlap / (area1 + area2 - overlap);
}

// One wavefront per row: the row is read with one coalesced load per 64 columns, the exponentials are evaluated one per
// lane, and the max / sum are then folded column by column in the original sequential order (values broadcast with
// readlane), so the result is bitwise the one-thread-per-row loop it replaces without its chain of dependent loads.
__global__ __launch_bounds__(256)
void row_softmax_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols, int mode) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* p = in + (size_t)r * cols;
    float mx = -INFINITY;
    for (int base = 0; base < cols; base += 64) {
        const float v = base + lane < cols ? p[base + lane] : -INFINITY;
        const int n = min(64, cols - base);
        for (int c = 0; c < n; ++c) mx = fmaxf(mx, __shfl(v, c, 64));
    }
    float s = 0.f;
    for (int base = 0; base < cols; base += 64) {
        const float e = base + lane < cols ? expf(p[base + lane] - mx) : 0.f;
        const int n = min(64, cols - base);
        for (int c = 0; c < n; ++c) s += __shfl(e, c, 64);
    }
    float* o = out + (size_t)r * cols;
    const float ls = logf(s);
    for (int base = 0; base < cols; base += 64) {
        if (base + lane < cols) {
            const float v = p[base + lane];
            o[base + lane] = mode == 0 ? expf(v - mx) / s : v - mx - ls;
        }
    }
}

}  // namespace

extern "C" int vps_roi_align(const float* const* feats, const int* ld, const int* Hs, const int* Ws, const float* scales,
                             int nlevels, float finest_scale, const float* rois, int R, int C, int P, int sample_num,
                             float* out, void* stream) {
    if (!feats || !ld || !Hs || !Ws || !scales || !rois || !out) return VPS_EARG(1);
    if (nlevels < 1 || nlevels > 4 || R < 0 || C <= 0 || (C & 3) || P <= 0 || sample_num <= 0) return VPS_EARG(2);
    if (R == 0) return 0;
    RoiLevels L;
    L.n = nlevels;
    for (int l = 0; l < 4; ++l) {
        const int s = l < nlevels ? l : nlevels - 1;
        if (!feats[s] || (ld[s] & 3)) return VPS_EARG(3);
        L.feat[l] = feats[s]; L.ld[l] = ld[s]; L.H[l] = Hs[s]; L.W[l] = Ws[s]; L.scale[l] = scales[s];
    }
    hipLaunchKernelGGL(roi_align_kernel, dim3(stream_grid((long)R * P * P * (C >> 2), 256)), dim3(256), 0,
                       (hipStream_t)stream, L, finest_scale, rois, R, C, P, sample_num, out);
    return vps_launch_status();
}

extern "C" int vps_nms_batched(const float* boxes, int nbatch, int nmax, const int32_t* counts_dev, float thr,
                               uint64_t* mask_ws, int32_t* keep, int32_t* nkeep, void* stream) {
    if (!boxes || !counts_dev || !mask_ws || !keep || !nkeep || nbatch <= 0 || nmax <= 0) return VPS_EARG(1);
    const int cb = (nmax + 63) / 64;
    if ((size_t)cb * sizeof(unsigned long long) > 60000) return VPS_EARG(2);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb, nbatch), dim3(64), 0, s, boxes, nmax, counts_dev, thr,
                       (unsigned long long*)mask_ws, cb);
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(nbatch), dim3(256), cb * sizeof(unsigned long long), s,
                       (const unsigned long long*)mask_ws, nmax, counts_dev, cb, keep, nkeep);
    return vps_launch_status();
}

extern "C" int vps_delta2bbox(const float* anchors, const float* deltas, const float* scores, float* boxes5, int n,
                              float std_x, float std_y, float std_w, float std_h, float img_h, float img_w, void* stream) {
    if (!anchors || !deltas || !scores || !boxes5 || n < 0) return VPS_EARG(1);
    if (n == 0) return 0;
    const float max_ratio = (float)fabs(log(16.0 / 1000.0));
    hipLaunchKernelGGL(delta2bbox_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, anchors, deltas, scores,
                       boxes5, n, std_x, std_y, std_w, std_h, img_h, img_w, max_ratio);
    return vps_launch_status();
}

extern "C" int vps_bbox_overlaps(const float* a, int lda, int m, const float* b, int ldb, int n, float* out, void* stream) {
    if (!a || !b || !out || m < 0 || n < 0) return VPS_EARG(1);
    if (m == 0 || n == 0) return 0;
    hipLaunchKernelGGL(bbox_overlaps_kernel, dim3(cdiv((long)m * n, 256)), dim3(256), 0, (hipStream_t)stream, a, lda, m, b,
                       ldb, n, out);
    return vps_launch_status();
}

extern "C" int vps_row_softmax(const float* in, float* out, int rows, int cols, int mode, void* stream) {
    if (!in || !out || rows < 0 || cols <= 0) return VPS_EARG(1);
    if (rows == 0) return 0;
    hipLaunchKernelGGL(row_softmax_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, in, out, rows, cols, mode);
    return vps_launch_status();
}
