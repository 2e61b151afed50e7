// Panoptic post-processing on the device: the per-frame body of
// /root/reference/tools/dataset/cityscapes_vps.py:183-224 (CityscapesVPS.get_unified_pan_result) as three passes over
// uint8 maps instead of ~250 boolean masks of 2 M pixels each on the host:
//   vps_unify_hist    hist[id][c] = #pixels with pan == id and seg == c for instance ids, pan_count[id] for every id
//   vps_unify_tables  the reference's per-instance decisions, sequential over the present ids, on one wavefront
//   vps_unify_write   out[p] = (seg_table, ins_table, obj_table)[pan[p]], uint8 [H][W][3]
// Integer work: bit-exact against the reference function (tests/test_postprocess.py, golden from the real function).
#include "common.h"

namespace {

// 16 pixels per thread (one 16-byte load of each map), runs inside the chunk folded into one update each. The workgroup
// keeps a private LDS histogram [256 ids][32 classes] + pan_count (semantic labels >= 32 go to the global table directly)
// and flushes its non-zero bins once: global atomics per frame drop from one per run to a few thousand per workgroup.
__global__ __launch_bounds__(256)
void unify_hist_kernel(const uint8_t* __restrict__ pan, const uint8_t* __restrict__ seg, long npix, int id_last_stuff,
                       int32_t* __restrict__ hist, int32_t* __restrict__ pan_count) {
    __shared__ int32_t lh[256 * 32];
    __shared__ int32_t cnt[256];
    for (int i = threadIdx.x; i < 256 * 32; i += 256) lh[i] = 0;
    cnt[threadIdx.x] = 0;
    __syncthreads();
    auto flush_run = [&](int key, int len) {
        if (key >= 0) {
            const int id = key >> 8, c = key & 255;
            if (c < 32) atomicAdd(&lh[id * 32 + c], len); else atomicAdd(&hist[key], len);
            atomicAdd(&cnt[id], len);
        } else {
            atomicAdd(&cnt[-1 - key], len);
        }
    };
    const long nchunk = (npix + 15) >> 4;
    for (long ch = (long)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunk; ch += (long)gridDim.x * blockDim.x) {
        const long p0 = ch << 4;
        uint8_t pv[16], sv[16];
        int n = 16;
        if (p0 + 16 <= npix) {
            *reinterpret_cast<uint4*>(pv) = *reinterpret_cast<const uint4*>(pan + p0);
            *reinterpret_cast<uint4*>(sv) = *reinterpret_cast<const uint4*>(seg + p0);
        } else {
            n = (int)(npix - p0);
            for (int i = 0; i < n; ++i) { pv[i] = pan[p0 + i]; sv[i] = seg[p0 + i]; }
        }
        int run_key = -1, run_len = 0;
        for (int i = 0; i < n; ++i) {
            const int id = pv[i];
            const int key = id > id_last_stuff ? id * 256 + sv[i] : -1 - id;   // stuff pixels: keyed by id only
            if (key != run_key) {
                if (run_len) flush_run(run_key, run_len);
                run_key = key; run_len = 0;
            }
            ++run_len;
        }
        if (run_len) flush_run(run_key, run_len);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256 * 32; i += 256)
        if (lh[i]) atomicAdd(&hist[(i >> 5) * 256 + (i & 31)], lh[i]);
    if (cnt[threadIdx.x]) atomicAdd(&pan_count[threadIdx.x], cnt[threadIdx.x]);
}

// One workgroup of 16 wavefronts. Phase A (parallel, one wavefront per id): arg-max class, its count and the total of every
// present instance id. Phase B (one thread, LDS only): the reference's sequential walk over the present ids in ascending
// order (the running index `idx` makes it sequential). Phase C (parallel): stuff areas of the updated map, void rule, tables.
// status: 0 ok, 1 = an instance id has no entry in cls_ind, 2 = no entry in obj_id (the reference raises IndexError there;
// the host wrapper does the same).
__global__ __launch_bounds__(1024)
void unify_tables_kernel(const int32_t* __restrict__ hist, const int32_t* __restrict__ pan_count, const int32_t* __restrict__ cls_ind,
                         int k, const int32_t* __restrict__ obj_id, int nobj, int id_last_stuff, long stuff_area_limit,
                         uint8_t* __restrict__ tables, int32_t* __restrict__ status) {
    __shared__ int seg_t[256], ins_t[256], obj_t[256], pcnt[256], top_c[256], top_n[256], cls_l[256], oid_l[256];
    __shared__ long tot[256], area[256];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t < 256) {
        cls_l[t] = t < k ? cls_ind[t] : 0;             // phase B reads LDS only (a global load per instance was most of its time)
        oid_l[t] = (obj_id && t < nobj) ? obj_id[t] : 0;
        seg_t[t] = t;                                  // pan_seg = pan.copy()
        ins_t[t] = t <= id_last_stuff ? 0 : t;         // pan_ins = pan.copy(); pan_ins[pan_ins <= id_last_stuff] = 0
        obj_t[t] = t;                                  // pan_obj = pan.copy()
        area[t] = 0;
        pcnt[t] = pan_count[t];
    }
    __syncthreads();
    // A: cls, cnt = np.unique(seg[region], return_counts=True); argmax = first maximum in ascending class order.
    // Wavefront w owns ids 16w .. 16w+15; its 16 row loads are all issued before the first reduction (issued one by one
    // behind the `present` test they cost 16 serial memory latencies per wavefront).
    int4 rows[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) rows[j] = *reinterpret_cast<const int4*>(hist + (wave * 16 + j) * 256 + lane * 4);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int pid = wave * 16 + j;
        int best = -1, best_c = 0;
        long sum = 0;
        const int v[4] = {rows[j].x, rows[j].y, rows[j].z, rows[j].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sum += v[e];
            if (v[e] > best) { best = v[e]; best_c = lane * 4 + e; }
        }
        for (int off = 32; off >= 1; off >>= 1) {
            const int ob = __shfl_xor(best, off, 64), oc = __shfl_xor(best_c, off, 64);
            sum += __shfl_xor(sum, off, 64);
            if (ob > best || (ob == best && oc < best_c)) { best = ob; best_c = oc; }
        }
        if (lane == 0) { top_c[pid] = best_c; top_n[pid] = best; tot[pid] = sum; }
    }
    __syncthreads();
    // B: the reference walks the present ids in ascending order with a running index `idx`; idx of id t is the number of
    // present instance ids below t, so every id can be decided independently once that rank is known (ballot + popcount).
    __shared__ int wave_present[4], st_sh;
    if (t == 0) st_sh = 0;
    const bool present = t < 256 && t > id_last_stuff && pcnt[t & 255] > 0;
    const unsigned long long bal = __ballot(present);
    if (t < 256 && lane == 0) wave_present[wave] = __popcll(bal);
    __syncthreads();
    if (present) {
        int my_idx = __popcll(bal & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) my_idx += wave_present[w];
        const int pid = t;
        if (pid == 255) {
            seg_t[255] = 255; ins_t[255] = 0;
        } else {
            const int ci = pid - id_last_stuff - 1;
            if (ci >= k) {
                atomicMax(&st_sh, 1);      // (an IndexError in the reference; which of several the host reports does not matter)
            } else {
                const int inst_cls = cls_l[ci] + id_last_stuff;
                const int best_c = top_c[pid];
                const bool to_stuff = best_c != inst_cls && 2 * (long)top_n[pid] >= tot[pid] && best_c <= id_last_stuff;   // max/sum >= 0.5, exactly
                if (!to_stuff) {
                    seg_t[pid] = inst_cls;
                    ins_t[pid] = my_idx + 1;
                    if (obj_id) {
                        if (my_idx >= nobj) atomicMax(&st_sh, 2);
                        else obj_t[pid] = oid_l[my_idx] + 1;
                    }
                } else {
                    seg_t[pid] = best_c;
                    ins_t[pid] = 0;
                    obj_t[pid] = 0;
                }
            }
        }
    }
    __syncthreads();
    if (t == 0) *status = st_sh;
    // C: stuff classes smaller than the limit become void; areas are those of the UPDATED semantic map
    if (t < 256 && pcnt[t] > 0) atomicAdd(reinterpret_cast<unsigned long long*>(&area[seg_t[t] & 255]), (unsigned long long)pcnt[t]);
    __syncthreads();
    if (t < 256) {
        const int c = seg_t[t] & 255;
        int sv = seg_t[t];
        if (c <= id_last_stuff && area[c] > 0 && area[c] < stuff_area_limit) sv = 255;
        tables[t] = (uint8_t)sv;
        tables[256 + t] = (uint8_t)ins_t[t];
        tables[512 + t] = (uint8_t)obj_t[t];           // uint8 maps: values wrap modulo 256 like the reference's in-place stores
    }
}

__global__ __launch_bounds__(256)
void unify_write_kernel(const uint8_t* __restrict__ pan, long npix, const uint8_t* __restrict__ tables, uint8_t* __restrict__ out) {
    __shared__ uint8_t t[768];
    for (int i = threadIdx.x; i < 768; i += 256) t[i] = tables[i];
    __syncthreads();
    const long nquad = (npix + 3) >> 2;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += (long)gridDim.x * blockDim.x) {
        const long p0 = q << 2;
        if (p0 + 4 <= npix) {
            const uint32_t v = *reinterpret_cast<const uint32_t*>(pan + p0);
            uint8_t o[12];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int id = (v >> (8 * i)) & 255;
                o[3 * i] = t[id]; o[3 * i + 1] = t[256 + id]; o[3 * i + 2] = t[512 + id];
            }
            uint32_t* dst = reinterpret_cast<uint32_t*>(out + 3 * p0);     // 12-byte groups: 3*p0 is a multiple of 4
            dst[0] = *reinterpret_cast<uint32_t*>(o); dst[1] = *reinterpret_cast<uint32_t*>(o + 4); dst[2] = *reinterpret_cast<uint32_t*>(o + 8);
        } else {
            for (long p = p0; p < npix; ++p) {
                const int id = pan[p];
                out[3 * p] = t[id]; out[3 * p + 1] = t[256 + id]; out[3 * p + 2] = t[512 + id];
            }
        }
    }
}

}  // namespace

extern "C" int vps_unify_hist(const uint8_t* pan, const uint8_t* seg, int64_t npix, int id_last_stuff, int32_t* hist,
                              int32_t* pan_count, void* stream) {
    if (!pan || !seg || !hist || !pan_count || npix <= 0 || id_last_stuff < 0 || id_last_stuff > 254) return VPS_EARG(1);
    if (((uintptr_t)pan | (uintptr_t)seg) & 15) return VPS_EARG(2);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(hist, 0, sizeof(int32_t) * 256 * 256, s);
    if (e != hipSuccess) return -(int)e;
    e = hipMemsetAsync(pan_count, 0, sizeof(int32_t) * 256, s);
    if (e != hipSuccess) return -(int)e;
    long g = ((npix + 15) >> 4) / (256 * 4); if (g > 512) g = 512; if (g < 1) g = 1;     // >= 4 chunks per thread, <= 2 workgroups per CU
    hipLaunchKernelGGL(unify_hist_kernel, dim3((unsigned)g), dim3(256), 0, s, pan, seg, (long)npix, id_last_stuff, hist, pan_count);
    return vps_launch_status();
}

extern "C" int vps_unify_tables(const int32_t* hist, const int32_t* pan_count, const int32_t* cls_ind, int k, const int32_t* obj_id,
                                int nobj, int id_last_stuff, int64_t stuff_area_limit, uint8_t* tables, int32_t* status, void* stream) {
    if (!hist || !pan_count || !tables || !status || k < 0 || (k > 0 && !cls_ind) || nobj < 0 || id_last_stuff < 0 || id_last_stuff > 254)
        return VPS_EARG(1);
    hipLaunchKernelGGL(unify_tables_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, hist, pan_count, cls_ind, k, obj_id, nobj,
                       id_last_stuff, (long)stuff_area_limit, tables, status);
    return vps_launch_status();
}

extern "C" int vps_unify_write(const uint8_t* pan, int64_t npix, const uint8_t* tables, uint8_t* out, void* stream) {
    if (!pan || !tables || !out || npix <= 0) return VPS_EARG(1);
    if (((uintptr_t)pan | (uintptr_t)out) & 3) return VPS_EARG(2);
    hipLaunchKernelGGL(unify_write_kernel, dim3(stream_grid((long)((npix + 3) >> 2), 256)), dim3(256), 0, (hipStream_t)stream, pan,
                       (long)npix, tables, out);
    return vps_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Input side: Normalize(mean, std, to_rgb) -> Pad(size_divisor) -> ImageToTensor of the test pipeline
// (configs/cityscapes/fusetrack.py:184-188; mmdet/datasets/pipelines/transforms.py:258-269 (Pad), :310-318 (Normalize),
// formating.py:52-67; mmcv 0.2.14 imnormalize = (float32(img)[, BGR->RGB] - mean) / std, impad_to_multiple = zero pad
// bottom/right) in one pass from the decoded uint8 HWC image: 6 MB uploaded instead of 25 MB, no host float work.
// fp32 subtraction and correctly rounded division: bit-exact with the NumPy expression.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256)
void image_prep_kernel(const uint8_t* __restrict__ img, int H, int W, int Hp, int Wp, float m0, float m1, float m2, float s0, float s1,
                       float s2, int to_rgb, float pad_val, float* __restrict__ out) {
    const long total = (long)Hp * Wp;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const int x = (int)(p % Wp), y = (int)(p / Wp);
        float v0 = pad_val, v1 = pad_val, v2 = pad_val;
        if (y < H && x < W) {
            const uint8_t* px = img + ((size_t)y * W + x) * 3;
            const float c0 = (float)px[to_rgb ? 2 : 0], c1 = (float)px[1], c2 = (float)px[to_rgb ? 0 : 2];
            v0 = (c0 - m0) / s0; v1 = (c1 - m1) / s1; v2 = (c2 - m2) / s2;
        }
        out[p] = v0; out[total + p] = v1; out[2 * total + p] = v2;
    }
}
}  // namespace

extern "C" int vps_image_prep(const uint8_t* img, int H, int W, int Hp, int Wp, const float* mean, const float* std, int to_rgb,
                              float pad_val, float* out, void* stream) {
    if (!img || !mean || !std || !out || H <= 0 || W <= 0 || Hp < H || Wp < W) return VPS_EARG(1);
    hipLaunchKernelGGL(image_prep_kernel, dim3(stream_grid((long)Hp * Wp, 256)), dim3(256), 0, (hipStream_t)stream, img, H, W, Hp, Wp,
                       mean[0], mean[1], mean[2], std[0], std[1], std[2], to_rgb, pad_val, out);
    return vps_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Resize of the test pipeline (transforms.py:107-122 -> mmcv.imrescale -> cv2.resize INTER_LINEAR on the decoded uint8 image):
// OpenCV's 8-bit bilinear, fixed point (resize.cpp: 11-bit coefficients; horizontal pass into int, vertical pass
// ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2), or the area average of an exact 2x shrink. Index / weight tables come
// from the host (they are per row / per column).
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256)
void resize_u8_kernel(const uint8_t* __restrict__ src, int H0, int W0, uint8_t* __restrict__ dst, int H, int W, int C,
                      const int32_t* __restrict__ xtab, const int32_t* __restrict__ ytab, int area2x) {
    const long total = (long)H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int y = (int)(idx / W), x = (int)(idx - (long)y * W);
        if (area2x) {
            const uint8_t* p = src + ((size_t)(2 * y) * W0 + 2 * x) * C;
            for (int c = 0; c < C; ++c)
                dst[idx * C + c] = (uint8_t)((p[c] + p[C + c] + p[(size_t)W0 * C + c] + p[(size_t)W0 * C + C + c] + 2) >> 2);
            continue;
        }
        const int x0 = xtab[3 * x], a0 = xtab[3 * x + 1], a1 = xtab[3 * x + 2];
        const int y0 = ytab[3 * y], b0 = ytab[3 * y + 1], b1 = ytab[3 * y + 2];
        const int x1 = min(x0 + 1, W0 - 1), y1 = min(y0 + 1, H0 - 1);
        const uint8_t* r0 = src + (size_t)y0 * W0 * C;
        const uint8_t* r1 = src + (size_t)y1 * W0 * C;
        for (int c = 0; c < C; ++c) {
            const int s0 = r0[x0 * C + c] * a0 + r0[x1 * C + c] * a1;
            const int s1 = r1[x0 * C + c] * a0 + r1[x1 * C + c] * a1;
            const int v = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2;
            dst[idx * C + c] = (uint8_t)min(max(v, 0), 255);
        }
    }
}
}  // namespace

extern "C" int vps_resize_u8(const uint8_t* src, int H0, int W0, uint8_t* dst, int H, int W, int C, const int32_t* xtab,
                             const int32_t* ytab, void* stream) {
    if (!src || !dst || H0 <= 0 || W0 <= 0 || H <= 0 || W <= 0 || C <= 0 || C > 4) return VPS_EARG(1);
    const int area2x = (W0 == 2 * W && H0 == 2 * H) ? 1 : 0;
    if (!area2x && (!xtab || !ytab)) return VPS_EARG(2);
    hipLaunchKernelGGL(resize_u8_kernel, dim3(stream_grid((long)H * W, 256)), dim3(256), 0, (hipStream_t)stream, src, H0, W0, dst, H, W, C,
                       xtab, ytab, area2x);
    return vps_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Second half of the output path: tools/dataset/cityscapes_vps.py:97-159 (converter_2ch_track_core). Per frame the reference
// builds one boolean mask per segment (np.unique over 1000*seg + obj) to paint it, take its bounding box and count it.
//   vps_segment_stats  per (seg, obj) pair: pixel count and bounding box, one pass (table [65536][5] int32)
//   vps_segment_paint  out[p] = lut[seg, obj] (colours chosen on the host by the reference's own colour generator)
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256)
void segment_stats_init_kernel(int32_t* __restrict__ stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;     // 65536 entries
    stats[5 * i] = 0; stats[5 * i + 1] = 0x7fffffff; stats[5 * i + 2] = 0x7fffffff; stats[5 * i + 3] = -1; stats[5 * i + 4] = -1;
}

// one thread per 8-pixel run of a row: runs of one segment are folded into one update (count, x range) each
__global__ __launch_bounds__(256)
void segment_stats_kernel(const uint8_t* __restrict__ pan2, int H, int W, int32_t* __restrict__ stats) {
    const int runs_per_row = (W + 7) >> 3;
    const long total = (long)H * runs_per_row;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int y = (int)(idx / runs_per_row), x0 = (int)(idx % runs_per_row) * 8;
        const int n = min(8, W - x0);
        const uint8_t* p = pan2 + ((size_t)y * W + x0) * 3;
        int key = -1, cnt = 0, xs = 0;
        for (int i = 0; i <= n; ++i) {
            const int k = i < n ? (p[3 * i] << 8 | p[3 * i + 2]) : -2;
            if (k != key) {
                if (cnt) {
                    int32_t* s = stats + 5 * key;
                    atomicAdd(&s[0], cnt); atomicMin(&s[1], xs); atomicMin(&s[2], y); atomicMax(&s[3], x0 + i - 1); atomicMax(&s[4], y);
                }
                key = k; cnt = 0; xs = x0 + i;
            }
            ++cnt;
        }
    }
}

__global__ __launch_bounds__(256)
void segment_paint_kernel(const uint8_t* __restrict__ pan2, long npix, const uint8_t* __restrict__ lut, uint8_t* __restrict__ out) {
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const int key = pan2[3 * p] << 8 | pan2[3 * p + 2];
        const uint8_t* c = lut + 3 * key;
        out[3 * p] = c[0]; out[3 * p + 1] = c[1]; out[3 * p + 2] = c[2];
    }
}
}  // namespace

extern "C" int vps_segment_stats(const uint8_t* pan_2ch, int H, int W, int32_t* stats, void* stream) {
    if (!pan_2ch || !stats || H <= 0 || W <= 0) return VPS_EARG(1);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(segment_stats_init_kernel, dim3(256), dim3(256), 0, s, stats);
    hipLaunchKernelGGL(segment_stats_kernel, dim3(stream_grid((long)H * ((W + 7) >> 3), 256)), dim3(256), 0, s, pan_2ch, H, W, stats);
    return vps_launch_status();
}

extern "C" int vps_segment_paint(const uint8_t* pan_2ch, int64_t npix, const uint8_t* lut, uint8_t* out, void* stream) {
    if (!pan_2ch || !lut || !out || npix <= 0) return VPS_EARG(1);
    hipLaunchKernelGGL(segment_paint_kernel, dim3(stream_grid((long)npix, 256)), dim3(256), 0, (hipStream_t)stream, pan_2ch, (long)npix, lut, out);
    return vps_launch_status();
}

// ------------------------------------------------------------------------------------------------
// VPQ evaluation: the confusion counts of tools/eval_vpq.py:150-157. The reference stacks the id maps of
// every nframes-long window and runs np.unique on nframes x H x W uint64 keys, for every window and every window length.
// Here every frame is counted ONCE into a dense (gt segment x predicted segment) table — the segment ids of a frame are
// known from the two JSONs — and the host sums the per-frame tables of a window (vps_amd/evaluate.py).
// ------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ int id_index(const uint32_t* __restrict__ ids, int n, uint32_t v) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {                      // sorted, unique
        const int mid = (lo + hi) >> 1;
        const uint32_t m = ids[mid];
        if (m == v) return mid;
        if (m < v) lo = mid + 1; else hi = mid - 1;
    }
    return n;                               // not listed
}

// 8 pixels per thread, runs of one (gt, pred) pair folded; LDS-private table when it fits
__global__ __launch_bounds__(256)
void pair_count_kernel(const uint8_t* __restrict__ gt, const uint8_t* __restrict__ pred, long npix, const uint32_t* __restrict__ gt_ids,
                       int ngt, const uint32_t* __restrict__ pred_ids, int npred, int32_t* __restrict__ counts, int use_lds) {
    extern __shared__ int32_t tab[];
    const int cols = npred + 1, size = (ngt + 1) * cols;
    if (use_lds) {
        for (int i = threadIdx.x; i < size; i += blockDim.x) tab[i] = 0;
        __syncthreads();
    }
    const long nchunk = (npix + 7) >> 3;
    for (long ch = (long)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunk; ch += (long)gridDim.x * blockDim.x) {
        const long p0 = ch << 3;
        const int n = (int)min(8L, npix - p0);
        uint32_t lg = 0xffffffffu, lp = 0xffffffffu;
        int gi = 0, pi = 0, run = 0, run_cell = -1;
        for (int i = 0; i < n; ++i) {
            const uint8_t* a = gt + 3 * (p0 + i);
            const uint8_t* b = pred + 3 * (p0 + i);
            const uint32_t g = a[0] | (a[1] << 8) | (a[2] << 16), q = b[0] | (b[1] << 8) | (b[2] << 16);
            if (g != lg) { gi = id_index(gt_ids, ngt, g); lg = g; }
            if (q != lp) { pi = id_index(pred_ids, npred, q); lp = q; }
            const int cell = gi * cols + pi;
            if (cell != run_cell) {
                if (run) atomicAdd(use_lds ? &tab[run_cell] : &counts[run_cell], run);
                run_cell = cell; run = 0;
            }
            ++run;
        }
        if (run) atomicAdd(use_lds ? &tab[run_cell] : &counts[run_cell], run);
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < size; i += blockDim.x)
            if (tab[i]) atomicAdd(&counts[i], tab[i]);
    }
}
}  // namespace

extern "C" int vps_pair_count(const uint8_t* gt_rgb, const uint8_t* pred_rgb, int64_t npix, const uint32_t* gt_ids, int ngt,
                              const uint32_t* pred_ids, int npred, int32_t* counts, void* stream) {
    if (!gt_rgb || !pred_rgb || !counts || npix <= 0 || ngt < 0 || npred < 0 || (ngt > 0 && !gt_ids) || (npred > 0 && !pred_ids)) return VPS_EARG(1);
    const long size = (long)(ngt + 1) * (npred + 1);
    if (size > (1L << 26)) return VPS_EARG(2);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * size, s);
    if (e != hipSuccess) return -(int)e;
    const int use_lds = size <= 12288;
    long g = ((npix + 7) >> 3) / (256 * 4); if (g > 512) g = 512; if (g < 1) g = 1;
    hipLaunchKernelGGL(pair_count_kernel, dim3((unsigned)g), dim3(256), use_lds ? sizeof(int32_t) * size : 0, s, gt_rgb, pred_rgb, (long)npix,
                       gt_ids, ngt, pred_ids, npred, counts, use_lds);
    return vps_launch_status();
}
