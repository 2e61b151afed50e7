/*
 * vps_hip.h — C-ABI of libvpship.so: hand-written HIP kernels (gfx950 / MI355X) for the
 * VPSNet FuseTrack inference hot path (PanopticFuseTrack.simple_test).
 *
 * Conventions
 *   - Every entry point returns 0 on success, a negative hipError_t otherwise, -1000-x for argument
 *     errors. Nothing throws, nothing allocates, nothing synchronises: the caller owns all device
 *     memory (including workspaces) and passes the hipStream_t (as void*) to launch on.
 *   - All tensors are fp32 device pointers unless stated. The internal activation layout is NHWC
 *     with an explicit per-pixel channel stride `ld` (floats) and channel offset `coff`, so that
 *     producers write straight into slices of concat buffers (no torch.cat on the path).
 *   - "ref:" lines cite the interface in mcahny/vps (relative to the reference root) that the
 *     symbol replaces.
 */
#ifndef VPS_HIP_H
#define VPS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ----------------------------------------------------------------------------------------------
 * Library info
 * -------------------------------------------------------------------------------------------- */
/* returns the ABI version of the library (bumped on any signature change) */
int vps_abi_version(void);
/* human readable build string: arch, kernels */
const char* vps_build_info(void);

/* ----------------------------------------------------------------------------------------------
 * Dense contraction: conv / transposed conv / linear / deformable conv as implicit GEMM on MFMA
 *   ref: every torch.nn.Conv2d / ConvTranspose2d / Linear + BatchNorm2d(eval) + ReLU/LeakyReLU on the
 *        path (SURVEY §8a a2,a7,a8,a11-a15,a17,a19,a20), mmdet/ops/dcn/src/deform_conv_cuda.cpp:151-250
 *        (deform_conv_forward_cuda) + deform_conv_cuda_kernel.cu:189-241 (deformable_im2col).
 * GEMM view: out[m][co] = act( scale[co] * sum_k A[m][k] * Wp[co][k] + shift[co] + res[m][co] )
 *   m  = (n, qy, qx) output-grid pixel,  k = (ky*KW + kx)*cin_pad + ci
 *   A[m][k] = in[n][qy*stride - pad_y + ky][qx*stride - pad_x + kx][ci]   (0 outside the image)
 *   deformable: A sampled bilinearly at (.. + dh, .. + dw) from `offset` (per-corner zeroing).
 * Transposed convs are `nclass` = s*s independent convs (one per output parity class); class c
 * (py = c / os_x, px = c % os_x) uses weight block c, pads pad_y[py], pad_x[px] and writes output pixel
 * (qy*os_y + py, qx*os_x + px).
 * -------------------------------------------------------------------------------------------- */
enum { VPS_ACT_NONE = 0, VPS_ACT_RELU = 1, VPS_ACT_LEAKY = 2 };
/* arithmetic of the contraction (accumulation is always fp32):
 *   VPS_PREC_F32    exact fp32 MFMA (v_mfma_f32_32x32x2_f32), needs `w`
 *   VPS_PREC_BF16   operands rounded to bf16 (RNE) when staged, 1 bf16 MFMA per product (~2^-8 relative): the plain
 *                   "bf16 in, fp32 accumulate" arithmetic of BASELINE config 5; activations stay fp32 in HBM; needs `w_split`
 *   VPS_PREC_BF16X3 fp32 operands split into 2 bf16 terms, 3 bf16 MFMAs per product (~2^-16 relative), needs `w_split`
 *   VPS_PREC_BF16X6 3 bf16 terms, 6 bf16 MFMAs per product (~2^-23 relative, fp32-grade), needs `w_split`
 *   VPS_PREC_F16X3  fp32 operands split into 2 fp16 terms with a scaled residual, 3 fp16 MFMAs per product, needs `w_split`:
 *                     x = h0 + 2^-11*h1,  h0 = fp16(x), h1 = fp16((x - h0) * 2^11)        (22 significand bits)
 *                     w = g0 + g1,        g0 = fp16(w), g1 = fp16(w - g0),  g2 = 2^-11*g0 (exact for |g0| >= 2^-3, i.e. weights less than 2^14 below their channel's maximum; fp16-rounded below), w pre-scaled per output channel
 *                     x*w ~ h0*g0 + h0*g1 + h1*g2   (dropped: 2^-11*h1*g1 and the two residual roundings, <= 3*2^-22 relative)
 *                   full precision for 2^-14 <= |x| <= 65504 (below: absolute error <= 2^-36; above: fp16 overflow, reported
 *                   through `status`); weights within 2^-15 of their channel's largest keep 22 bits. */
enum { VPS_PREC_F32 = 0, VPS_PREC_BF16 = 1, VPS_PREC_BF16X3 = 2, VPS_PREC_BF16X6 = 3, VPS_PREC_F16X3 = 4 };

typedef struct vps_conv_desc {
    /* input activation, NHWC */
    const float* in;
    int32_t N, H, W;
    int32_t in_ld;      /* floats per pixel (multiple of 4) */
    int32_t in_coff;    /* first channel used (multiple of 4) */
    int32_t cin_pad;    /* channels contracted, multiple of 4; weights are zero for the pad */
    /* packed weights [nclass][cout_pad][kpad], k contiguous; kpad multiple of 32, cout_pad of tile_n */
    const float* w;
    int32_t cout, cout_pad, kpad;
    int32_t KH, KW, stride;
    int32_t pad_y[2], pad_x[2];
    /* output */
    float* out;
    int32_t Ho, Wo;     /* full output spatial size */
    int32_t out_ld, out_coff;
    int32_t Qh, Qw;     /* GEMM pixel grid per class (== Ho,Wo for a conv) */
    int32_t os_y, os_x; /* 1 for conv, s for a stride-s transposed conv */
    int32_t nclass;     /* os_y*os_x */
    /* epilogue */
    const float* scale; /* [cout] or NULL (=1) */
    const float* shift; /* [cout] or NULL (=0) */
    const float* res;   /* residual NHWC or NULL; read at (oy>>res_shift, ox>>res_shift) */
    int32_t res_ld, res_coff, res_shift;
    int32_t act;        /* VPS_ACT_* */
    float slope;        /* LeakyReLU negative slope */
    /* deformable sampling (NULL = plain conv): NHWC [N][Ho][Wo][off_ld], ch 2*(ky*KW+kx) = dh, +1 = dw */
    const float* offset;
    int32_t off_ld;
    /* tiling */
    int32_t tile_n;     /* 32, 64 or 128; 256 for a deformable layer (offset set) in VPS_PREC_F16X3 with korder 1 and 256 | cout_pad */
    int32_t ksplit;     /* >=1; >1 needs ws of ksplit*M*cout_pad floats (M = nclass*N*Qh*Qw) */
    float* ws;
    /* split modes: 16-bit weight planes, no `w`. P planes: bf16 1, bf16x3 2, bf16x6 3 (plane p = bf16 RNE of the residual after p
     * terms), f16x3 3 (g0, g1, 2^-11*g0 of the per-channel pre-scaled weight; the scale's inverse is folded into `scale`).
     *   with `offset` (deformable) AND korder 0:  [P][nclass][cout_pad][kpad] (two-barrier kernel, weights through LDS)
     *   otherwise - deformable layers in the chunk-major order included - MFMA-fragment order (weights go straight to registers, one coalesced 1 KB load per fragment):
     *                               [P][nclass][cout_pad/32][kpad/16][lane 0..63][8], lane = 32*((k/8)%2) + cout%32 */
    int32_t prec;       /* VPS_PREC_* */
    const void* w_split;
    /* k ordering of the packed weights: 0 = tap-major k = (ky*KW+kx)*cin_pad + ci;
     * 1 = chunk-major k = ((ci/32)*KH*KW + ky*KW+kx)*32 + ci%32, kpad = KH*KW*ceil(cin_pad/32)*32 (L2-friendly) */
    int32_t korder;
    /* VPS_PREC_F16X3: device word that gets bit 0 OR-ed in when an activation beyond the fp16 range (|x| > 65504) was staged
     * (the result of that launch is then not fp32-grade); NULL = not reported. Never written in the other modes. */
    int32_t* status;
    /* GroupNorm sums of the OUTPUT taken by the epilogue (the deformable conv + GroupNorm towers, upsnetFPN.py:39-52): when non-NULL,
     * the sum and the sum of squares of the stored values of every group of gn_cpg consecutive output channels (4, or a multiple
     * of 8) are ADDED to gn_stats[r][2 g], gn_stats[r][2 g + 1] (doubles; the caller zeroes all gn_rep copies; a block adds to
     * copy r = block index % gn_rep, gn_rep a power of two, so that the atomics of ~1000 blocks spread over gn_rep * 4 cache lines)
     * - the statistics pass of vps_groupnorm_relu without re-reading the tensor; finish with vps_groupnorm_apply, which adds the
     * copies up. Deformable launches (offset != NULL) of the split-operand modes only, ksplit == 1, no residual,
     * cout/out_ld/out_coff multiples of 4; VPS_EARG otherwise. */
    double* gn_stats;
    int32_t gn_cpg, gn_rep;
    /* ksplit > 1: int32 [nclass * tiles] tickets, ZERO on entry (the launch leaves them zero). Non-NULL: the block that finishes a
     * tile's last split sums the partials (in split order) and applies the epilogue itself; NULL: a separate reduce launch does.
     * tiles = ceil(M/128) * cout_pad/tile_n for the pipelined kernels; the halo kernels tile in 8x16 / 8x32 patches:
     * N * ceil(Qh/8) * ceil(Qw/16) * cout_pad/tile_n is an upper bound for all of them. */
    int32_t* tile_counter;
    /* VPS_PREC_F16X3, optional: the weights of a thin-input layer (cin_pad 4 / 8 / 12, cout_pad 64, KH == KW, korder 0) in the packing
     * of the thin-input kernel (csrc/conv_thin.hip): fp16 [KH][plane 0..1][ceil(KW*cin_pad/16)][cout/32][lane = 32*(k/8 % 2) + cout % 32][8],
     * k = position within one kernel ROW's KW*cin_pad values (tap-major, zero-padded to a multiple of 16), same per-channel scaling
     * as w_split. NULL, or a shape the kernel has no instance for: the launch uses w_split. */
    const void* w_thin;
} vps_conv_desc;

int vps_conv2d(const vps_conv_desc* d, void* stream);

/* ----------------------------------------------------------------------------------------------
 * FlowNet2 gather / scan ops (HBM bound)
 * All take explicit element strides (sn, sc, sh, sw) so the same kernel serves the reference's NCHW
 * operator API and the NHWC pipeline.
 * -------------------------------------------------------------------------------------------- */
typedef struct vps_tensor4 {
    float* p;
    int64_t sn, sc, sh, sw; /* element strides */
} vps_tensor4;

/* ref: resample2d_package/resample2d.py:40-49, resample2d_kernel.cu:15-72 (kernel_size=1, bilinear,
 * border clamp of the four tap indices). out[b,c,y,x] = bilinear(in[b,c], x+flow[b,0,y,x], y+flow[b,1,y,x]) */
int vps_resample2d(vps_tensor4 in, vps_tensor4 flow, vps_tensor4 out,
                   int B, int C, int H, int W, void* stream);

/* ref: channelnorm_package/channelnorm.py:31-38, channelnorm_kernel.cu:18-60. out[b,0,y,x] = sqrt(sum_c in^2) */
int vps_channelnorm(vps_tensor4 in, vps_tensor4 out, int B, int C, int H, int W, void* stream);

/* ref: correlation_package/correlation.py:47-61, correlation_cuda.cc:10-87,
 * correlation_cuda_kernel.cu:46-147 (kernel_size=1, stride1=1). Inputs NHWC (ld/coff), output channel
 * tc = (tj+r)*(2r+1)+(ti+r), r = max_disp/stride2, written to out[.., out_coff+tc] with optional LeakyReLU.
 * out = (1/C) * sum_c in1[y,x,c] * in2[y+tj*s2, x+ti*s2, c], zero outside. */
int vps_correlation(const float* in1, int ld1, int coff1, const float* in2, int ld2, int coff2,
                    float* out, int out_ld, int out_coff,
                    int N, int H, int W, int C, int max_disp, int stride2,
                    int act, float slope, void* stream);
/* the same operator in SPLIT fp16 on the matrix cores (round 5; the two configurations of the path with C = 256: max_disp 20 / stride2 2
 * and max_disp 4 / stride2 1; every other shape runs the exact kernels): operands as fp16 pairs with a scaled residual (22 significand
 * bits), a*b ~ h0 k0 + 2^-11 (h0 k1 + h1 k0), fp32 accumulate - fp32-grade like VPS_PREC_F16X3 for 2^-14 <= |x| <= 65504. status: device
 * word that receives bit 0 when an operand lies beyond the fp16 range: the caller then repeats the call with vps_correlation. */
int vps_correlation_f16(const float* in1, int ld1, int coff1, const float* in2, int ld2, int coff2,
                        float* out, int out_ld, int out_coff, int N, int H, int W, int C,
                        int max_disp, int stride2, int act, float slope, int32_t* status, void* stream);

/* ref: flow_modules/flow_modules.py:126-148 (WarpingLayer): grid = linspace(-1,1) + flow/((W-1)/2),
 * F.grid_sample(bilinear, zeros padding, align_corners=False). NHWC in/out, flow NHWC [N,H,W,flow_ld] (ch0=x,1=y) */
int vps_flow_warp(const float* in, int in_ld, int in_coff, const float* flow, int flow_ld, int flow_coff,
                  float* out, int out_ld, int out_coff, int N, int H, int W, int C, void* stream);

/* layout transposes: NCHW (contiguous) <-> NHWC (ld/coff); pad channels [C, Cpad) are written as zero */
int vps_nchw_to_nhwc(const float* in, float* out, int out_ld, int out_coff, int N, int C, int H, int W,
                     int Cpad, void* stream);
int vps_nhwc_to_nchw(const float* in, int in_ld, int in_coff, float* out, int N, int C, int H, int W, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Resize / pool / elementwise (NHWC)
 * -------------------------------------------------------------------------------------------- */
/* F.interpolate(mode=bilinear, align_corners=False) (mode 0) or nearest (mode 1); out = alpha * resized.
 * ref: flownet2.py:143,155 (upsample1/2 bilinear x4), :167,180 (nearest x4), panoptic_fusetrack.py:140-142,
 * upsnetFPN.py:73-80, tcea_modules.py:72 */
int vps_resize(const float* in, int in_ld, int in_coff, int Hi, int Wi,
               float* out, int out_ld, int out_coff, int Ho, int Wo,
               int N, int C, int mode, float alpha, void* stream);
/* 3x3 stride-2 pad-1 max (mode 0, -inf pad; ref resnet.py:465, tcea_modules.py:27) or avg pool
 * (mode 1, count_include_pad=True; ref tcea_modules.py:28) */
int vps_pool3x3s2(const float* in, int in_ld, int in_coff, int Hi, int Wi,
                  float* out, int out_ld, int out_coff, int N, int C, int mode, void* stream);
/* BFP gather: out = (sum_l nearest_resize(level_l)) / L at level-0 resolution; ratio[l] = H0 / H_l.
 * ref: extra_necks/bfp_tcea.py:96-109 (refine_level = 0) */
int vps_bfp_gather(const float* const* levels, const int* ld, const int* ratio, int nlevels,
                   float* out, int out_ld, int out_coff, int N, int H0, int W0, int C, void* stream);
/* BFP scatter: out = adaptive_max_pool2d(bsf, (H0/ratio, W0/ratio)) + level. ref: bfp_tcea.py:139-147 */
int vps_bfp_scatter(const float* bsf, int bsf_ld, const float* level, int lvl_ld, float* out, int out_ld,
                    int N, int H0, int W0, int C, int ratio, void* stream);
/* the same for ALL levels in one pass over bsf (round 6): level l = adaptive_max_pool2d(bsf, (H0 >> l, W0 >> l)) + levels[l], l < nlevels
 * (2..5; ratio 2^l), H0 and W0 multiples of 2^(nlevels-1), C a multiple of 32. Bitwise vps_bfp_scatter per level; bsf is read once. */
int vps_bfp_scatter_all(const float* bsf, int bsf_ld, const float* const* levels, const int* lvl_ld, float* const* outs,
                        const int* out_ld, int nlevels, int N, int H0, int W0, int C, void* stream);
/* y = x*a + b elementwise over a slice (used for flow scaling) */
int vps_axpb(const float* in, int in_ld, int in_coff, float* out, int out_ld, int out_coff,
             int64_t npix, int C, float a, float b, void* stream);

/* FlowNet2 input prep. ref: utils/flow_utils.py:5-10 (denormalize), flownet2.py:135-139 (rgb_mean over both
 * frames, (x-mean)/rgb_max, cat). img/ref NCHW [1,3,H,W] normalised; out NHWC [H*W][out_ld] ch0-2 img, 3-5 ref.
 * `partial` is a caller workspace of 3*nblk doubles, `mean3` 3 floats. */
int vps_flow_prep(const float* img, const float* ref, const float* mean3, const float* std3,
                  float* out, int out_ld, int H, int W, double* partial, int nblk, float* rgb_mean_out,
                  void* stream);
/* the same with the pair zero-padded (in 0..255 RGB space, bottom / right) to [Hp][Wp] before the mean is taken and x6 is
 * written — ref: panoptic_fusetrack.py:125-128 (800x1600 -> 832x1664, 200x400 -> 256x448; FlowNet2 needs multiples of 64). */
int vps_flow_prep_pad(const float* img, const float* ref, const float* mean3, const float* std3,
                      float* out, int out_ld, int H, int W, int Hp, int Wp, double* partial, int nblk, float* rgb_mean_out,
                      void* stream);
/* FlowNet2 inter-stage tensor build (ref flownet2.py:142-151,154-163,166-174,179-187): given a 1/4-res
 * 2-channel flow (NHWC, ld/coff) compute flow_full = up4(div_mode ? flow/mul : flow*mul) (up_mode 0
 * bilinear / 1 nearest), warped = resample2d(x[3:6], flow_full), diff = x[0:3]-warped, and write any of:
 *   x[0:3] -> out[img_off..+3], flow_full (/flow_out_div if >0) -> out[flow_off..+2],
 *   warped -> out[warp_off..+3], ||diff||2 -> out[diffnorm_off], ||flow_full||2 -> out[flownorm_off]
 * (offset < 0 = skip). H, W multiples of 4. */
int vps_flow_stage(const float* x6, int x_ld, const float* flow_lo, int flo_ld, int flo_coff,
                   int H, int W, int up_mode, float mul, int div_mode,
                   float* out, int out_ld, int flow_off, float flow_out_div, int warp_off,
                   int diffnorm_off, int flownorm_off, int img_off, void* stream);

/* The same stage writing WHOLE 12-float pixels (three 16-byte stores; x6 with x_ld == 8, out with out_ld == 12, both 16-byte
 * aligned), same formulas as vps_flow_stage (+ vps_axpb for the image channels):
 *   mode 0, FlowNetS input (flownet2.py:142-163): out = [x6[0..5] | warp(img2, f) | f / mul | ||img1 - warp||], f = bilinear x4 of
 *           flow_a * mul (flow_b unused);
 *   mode 1, FlowNetFusion input (flownet2.py:166-187): out = [img1 | f_b | f_a | |f_b| | |f_a| | ||img1 - warp(img2, f_b)|| |
 *           ||img1 - warp(img2, f_a)|| | 0], f_a = nearest x4 of flow_a * mul (FlowNetS_2), f_b = nearest x4 of flow_b / mul (FlowNetSD). */
int vps_flow_stage_full(const float* x6, int x_ld, const float* flow_a, int a_ld, int a_coff,
                        const float* flow_b, int b_ld, int b_coff, int H, int W, int mode, float mul,
                        float* out, int out_ld, void* stream);

/* GroupNorm(G) + ReLU over NHWC (N=1): stats pass then apply pass. ref: upsnetFPN.py:39-52 (GroupNorm(32)).
 * in is a coff-0 buffer, the result goes to out[.., out_coff + c]. `stats` workspace: 2*G doubles (zeroed by the call). */
int vps_groupnorm_relu(const float* in, int in_ld, float* out, int out_ld, int out_coff, int64_t npix, int C, int G,
                       const float* gamma, const float* beta, float eps, int relu,
                       double* stats, void* stream);
/* The apply pass alone: `stats` already holds nrep copies [nrep][2*G] of partial sums (vps_conv_desc.gn_stats / gn_rep of the
 * conv that produced `in`); 2*G <= 512. */
int vps_groupnorm_apply(const float* in, int in_ld, float* out, int out_ld, int out_coff, int64_t npix, int C, int G,
                        const float* gamma, const float* beta, float eps, int relu,
                        const double* stats, int nrep, void* stream);

/* TCEA temporal attention, N=2 frames, center 0. ref: utils/tcea_modules.py:50-65.
 * emb [npix][>=2C] holds tAtt_1(frame0) in channels [0,C) and tAtt_1(frame1) in [C,2C); emb_ref = tAtt_2(frame0);
 * fea0/fea1 are the two aligned frames (pointers already offset to their first channel, 16B aligned).
 * out[npix][2C] = [fea0 * sigmoid(<emb0,emb_ref>) | fea1 * sigmoid(<emb1,emb_ref>)] (frame-major channels). */
int vps_tcea_temporal(const float* emb, int emb_ld, const float* emb_ref, int ref_ld,
                      const float* fea0, int f0_ld, const float* fea1, int f1_ld,
                      float* out, int out_ld, int64_t npix, int C, void* stream);
/* out = fea * sigmoid(att) * 2 + att_add. ref: tcea_modules.py:74-77 */
int vps_tcea_modulate(const float* fea, const float* att, const float* att_add, float* out,
                      int64_t n, void* stream);
/* the same on channel windows: pointers at the first channel of each window, leading dimensions in floats, C channels (4 | C, ld) */
int vps_tcea_modulate_ld(const float* fea, int fea_ld, const float* att, int att_ld, const float* att_add, int add_ld,
                         float* out, int out_ld, int64_t npix, int C, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Detection ops
 * -------------------------------------------------------------------------------------------- */
/* ref: mmdet/ops/roi_align/roi_align.py:59-87, src/roi_align_kernel.cu:16-124 + roi_extractors/
 * single_level.py:54-107 (level mapping lvl = floor(log2(sqrt(w*h)/56 + 1e-6)) clamped to [0,nlvl-1]).
 * feats: NHWC levels (ld each), spatial_scale 1/stride; rois [R][5] (b,x1,y1,x2,y2); out NHWC [R][P][P][C] */
int vps_roi_align(const float* const* feats, const int* ld, const int* Hs, const int* Ws, const float* scales,
                  int nlevels, float finest_scale, const float* rois, int R, int C, int P, int sample_num,
                  float* out, void* stream);

/* ref: mmdet/ops/nms/src/nms_kernel.cu:13-130 and utils/upsnet/nms/nms_kernel.cu:40-150.
 * Batched: boxes [nbatch][nmax][5] each sorted by descending score, counts_dev[nbatch] valid boxes (device),
 * IoU with the +1 convention, suppress if IoU > thr. mask_ws: nbatch*nmax*ceil(nmax/64) uint64.
 * keep [nbatch][nmax] int32 (indices into the sorted list, ascending), nkeep[nbatch] (device).
 * The greedy reduce (a host loop in the reference) runs on the device, one wavefront per batch entry. */
int vps_nms_batched(const float* boxes, int nbatch, int nmax, const int32_t* counts_dev, float thr,
                    uint64_t* mask_ws, int32_t* keep, int32_t* nkeep, void* stream);

/* ref: core/bbox/transforms.py:34-68 (delta2bbox, means 0, stds given, clamp to img, wh_ratio_clip 16/1000).
 * anchors [n][4], deltas [n][4] -> boxes5 [n][5] (x1,y1,x2,y2,score) */
int vps_delta2bbox(const float* anchors, const float* deltas, const float* scores, float* boxes5, int n,
                   float std_x, float std_y, float std_w, float std_h, float img_h, float img_w, void* stream);

/* ref: core/bbox/geometry.py:4-63 (bbox_overlaps, +1 convention). a [m][4/5] lda, b [n][ldb] -> out [m][n] */
int vps_bbox_overlaps(const float* a, int lda, int m, const float* b, int ldb, int n, float* out, void* stream);

/* row softmax / log-softmax for small matrices: in [rows][cols] -> out. mode 0 softmax, 1 log_softmax */
int vps_row_softmax(const float* in, float* out, int rows, int cols, int mode, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Panoptic head
 * -------------------------------------------------------------------------------------------- */
/* ref: utils/mask_removal.py:29-92, body of the per-box loop (boxes visited in descending cls_prob order by
 * the host): resize the SxS (28x28) mask logits to the int32-truncated box (cv2.resize INTER_LINEAR on float32),
 * binarise > 0, compare with the class occupancy plane (uint8 [H][W]).
 *   vps_mask_count : counts[0] = #mask pixels in the clipped box, counts[1] = #mask pixels already occupied
 *   vps_mask_commit: keep = counts[0] != 0 && !(counts[1]/counts[0] > thr) decided ON THE DEVICE;
 *                    *flag = keep; kept masks are added to the occupancy plane. No host sync between boxes. */
int vps_mask_count(const float* logit, int S, int bx1, int by1, int bx2, int by2, int H, int W,
                   const uint8_t* occ, int32_t* counts, void* stream);
int vps_mask_commit(const float* logit, int S, int bx1, int by1, int bx2, int by2, int H, int W,
                    uint8_t* occ, const int32_t* counts, double thr, int32_t* flag, void* stream);

/* the whole MaskRemoval box loop in ONE launch (one workgroup per class walks the score-sorted boxes; masks of different
 * classes never interact). boxes [n][4] int32-truncated (x1,y1,x2,y2), cls0[n] 0-based class, mask_idx[n] row of logits,
 * all in descending-score order and on the device; occ: ncls*H*W uint8 workspace (zeroed by the call); flags[n] = kept. */
int vps_mask_removal(const float* logits, int S, const int32_t* boxes, const int32_t* cls0, const int32_t* mask_idx,
                     int n, int ncls, int H, int W, uint8_t* occ, double thr, int32_t* flags, void* stream);

/* the same loop in ONE launch with one workgroup PER BOX of the score-sorted walk (round 5): a box finds the earlier same-class boxes
 * whose rectangles intersect its own, waits (bounded) until each has published its decision, then counts, decides (flags[i]) and
 * commits - independent boxes run side by side, dependent ones back to back without a launch in between. Arguments as above
 * (n <= 256, 2 <= S <= 32, W % 4 == 0, occ 4-byte aligned; occ and `done` [n] are zeroed by the call). status: bit 2 (value 4) is OR-ed in
 * when a wait expired (boxes wait for lower workgroup indices only, but HIP promises neither dispatch order nor progress): the flags of
 * that call are not valid - the caller repeats the walk with vps_mask_level (vps_amd/detector.py does, counted in
 * panoptic_ops.MR_RECOVERIES). VPS_MR_SPIN_LIMIT in the environment overrides the number of polls per dependency (tests). */
int vps_mask_removal_dep(const float* logits, int S, const int32_t* boxes, const int32_t* cls0, const int32_t* mask_idx,
                         int n, int ncls, int H, int W, uint8_t* occ, double thr, int32_t* flags, int32_t* done,
                         int32_t* status, void* stream);

/* the same decisions WITHOUT a dependency chain (round 6, the detector's default): rank[n] = position of box i among the boxes of its
 * class in the score-sorted walk, max_rank = its largest value (<= 126: at most 127 boxes per class; ncls <= 32, n <= 256). One pass over
 * the frame gives every pixel its per-class pattern of covering boxes (per-box pixel counts by wavefront ballots, patterns with >= 2
 * bits counted in a per-class hash table), then one wavefront per class walks its boxes over the distinct patterns: overlap_i = sum of
 * count(P) over the patterns P with bit i and a kept earlier bit (mask_removal.py:75-88). A class with more than 64 boxes takes a second
 * pass for its ranks 64..126 (key = their pattern + "covered by a kept box of ranks 0..63"). scratch: 8-byte aligned,
 * (n rounded up to 2) int32 + (ncls + groups * ncls * 4096) 64-bit words with groups = 1 + (max_rank >= 64), zeroed by the call.
 * status: bit 2 (value 4) is OR-ed in when a hash table (2048 distinct overlap patterns per class and group) is full: the flags of that
 * call are not valid - the caller repeats the walk with vps_mask_level, as for vps_mask_removal_dep. No occupancy plane, no
 * cross-workgroup wait. */
int vps_mask_removal_hist(const float* logits, int S, const int32_t* boxes, const int32_t* cls0, const int32_t* mask_idx,
                          const int32_t* rank, int max_rank, int n, int ncls, int H, int W, int32_t* scratch, size_t scratch_bytes,
                          double thr, int32_t* flags, int32_t* status, void* stream);

/* one dependency LEVEL of the MaskRemoval loop (count launch + commit launch): `level` = nlevel indices (device) into the
 * score-sorted box arrays whose boxes are mutually independent (no earlier same-class box of the same level intersects
 * them); counts [n][2] and occ must have been zeroed by the caller before the first level; max_area = largest clipped
 * box area of the level (grid sizing). Keep decision on the device, flags[i] written. */
int vps_mask_level(const float* logits, int S, const int32_t* boxes, const int32_t* cls0, const int32_t* mask_idx,
                   const int32_t* level, int nlevel, int max_area, int H, int W, uint8_t* occ, int32_t* counts,
                   double thr, int32_t* flags, void* stream);

/* one kept instance of the panoptic combine */
typedef struct vps_pan_inst {
    int32_t sx0, sy0, sx1, sy1;     /* SegTerm crop [x0,x1) x [y0,y1): int(b), int(round(b)+1) (unary_logits.py:102-105) */
    int32_t seg_ch;                 /* class_mapping[cls] channel of fcn_output */
    int32_t bx1, by1, bx2, by2;     /* MaskRemoval int32-truncated box (mask_removal.py:56-62) */
    int32_t mask_idx;               /* row of mask_logits [*][S][S] */
} vps_pan_inst;

/* ref: upsnetFPN.py:81 (bilinear x4 of fcn_score), utils/unary_logits.py:81-108 (SegTerm),
 * mask_removal.py:88 (paste), panoptic_fusetrack.py:588-597 (cat, softmax, argmax; argmax of fcn_output).
 * fcn_score NHWC [Hs][Ws][score_ld]; fcn_output (x H/Hs bilinear) is recomputed per pixel, never stored.
 * pan/sem: uint8 [H][W] (ids 0..nstuff-1 stuff, nstuff+j = j-th instance; k <= 255-nstuff). */
int vps_panoptic_combine(const float* fcn_score, int score_ld, int Hs, int Ws, int nclass, int nstuff,
                         const vps_pan_inst* inst, int k, const float* mask_logits, int S,
                         uint8_t* pan, uint8_t* sem, int H, int W, void* stream);

/* same, with the instance count read from DEVICE memory (k_dev[0] <= kmax; k_dev[0] > 255 - nstuff raises status bit 0 of
 * k_dev[2] and writes nothing): the table comes from vps_pan_instances, the host never sees the kept list before the launch */
int vps_panoptic_combine_dev(const float* fcn_score, int score_ld, int Hs, int Ws, int nclass, int nstuff,
                             const vps_pan_inst* inst, const int32_t* k_dev, const float* mask_logits, int S,
                             uint8_t* pan, uint8_t* sem, int H, int W, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Device-resident detection post-processing (csrc/head_ops.hip): the order-defining HOST code of the reference's heads as
 * single-workgroup kernels, so that a frame needs ONE mid-frame D2H (the detection list) and one at its end (kept list + ids).
 * -------------------------------------------------------------------------------------------- */
/* HOST functions (no device work, no stream): PNG decode for the input pipeline - `mmcv.imread` in datasets/pipelines/loading.py:43-68
 * (cv2.imread(IMREAD_COLOR): BGR uint8). Called through the C-ABI they run without the Python interpreter lock, so a pool of plain
 * threads scales. `file` = the file's bytes. vps_png_info: size / channel count of an 8-bit non-interlaced RGB, RGBA or grey PNG;
 * any other PNG flavour (16-bit, palette, interlaced) returns an argument error and the caller uses its general decoder.
 * vps_png_decode_bgr8: out [H][W][3] BGR (alpha dropped, grey replicated), out_capacity >= H*W*3 bytes. */
int vps_png_info(const uint8_t* file, int64_t nbytes, int32_t* H, int32_t* W, int32_t* channels);
int vps_png_decode_bgr8(const uint8_t* file, int64_t nbytes, uint8_t* out, int64_t out_capacity);

/* ref: models/anchor_heads/rpn_head.py:62-91 (sigmoid objectness, `scores.topk(nms_pre)`, gathers) + core/anchor/anchor_generator.py:55-72
 * (grid anchors) + core/bbox/transforms.py:34-68 (delta2bbox, means 0, clipped to the image) for ALL levels: one chip-wide scoring
 * launch + one select / sort / decode launch with one workgroup per level. cls[l] / reg[l]: NHWC maps [H_l][W_l][ld] of level l
 * (device pointers in host arrays; channels [0, A) / [0, 4A) used), base_anchors: device [nlv][A][4] (the rounded base anchors of
 * gen_base_anchors), strides / stds [4]: host arrays. Scratch: keys [sum_l H_l*W_l*A] uint32, hist [nlv * 4096] int32 ZERO on entry
 * (the launch leaves it zero). boxes [nlv][nms_pre][5]: the min(H_l*W_l*A, nms_pre) best positions of level l in DESCENDING score
 * order (equal scores: ascending position), rows beyond that count zeroed. nlv <= 8, nms_pre <= 8192. */
int vps_rpn_select(const float* const* cls, const int32_t* cls_ld, const float* const* reg, const int32_t* reg_ld,
                   const int32_t* Hs, const int32_t* Ws, const float* strides, int nlv, int A, const float* base_anchors,
                   int nms_pre, const float* stds, float img_h, float img_w, uint32_t* keys, int32_t* hist, float* boxes, void* stream);

/* ref: models/anchor_heads/rpn_head.py:94-104 (`mlvl_proposals` cat, `[:nms_post]` per level, top `max_num` by score).
 * boxes [nlv][nmax][5] per level in descending score order, keep [nlv][nmax] / nkeep [nlv] as written by vps_nms_batched.
 * out [max_num][5] (rows >= n_out[0] zeroed), n_out[0] = min(max_num, sum_l min(nkeep[l], nms_post)). nlv*nms_post <= 8192. */
int vps_rpn_collect(const float* boxes, const int32_t* keep, const int32_t* nkeep, int nlv, int nmax, int nms_post, int max_num,
                    float* out, int32_t* n_out, void* stream);

/* ref: models/utils/mask_roi.py:43-95 (class_agnostic=True, clip_boxes=True) + utils/upsnet/bbox/bbox_transform.py:45-60,290-330
 * + utils/upsnet/nms/gpu_nms.pyx:23-38 (`order = scores.argsort()[::-1]`).
 * rois [n][5], bbox_delta [n][4*nc], cls_prob [n][nc]; n_valid (device, may be NULL): rows >= n_valid[0] are ignored.
 * Candidates q = roi*(nc-1) + cls-1 with prob > score_thresh, sorted by descending score (equal scores: larger q first),
 * refined + clipped boxes in fp32 in the reference's operation order -> dets [<= 8192][5] (x1,y1,x2,y2,score), cand [<= 8192] = q,
 * m_out[0] = count, m_out[1] = 1 if more than 8192 candidates passed (the list is then truncated), m_out[2] = rois considered.
 * reg_weights: host float[4]. */
int vps_maskroi_select(const float* rois, const float* bbox_delta, const float* cls_prob, int n, const int32_t* n_valid,
                       int num_classes, float score_thresh, const float* reg_weights, float im_h, float im_w, float* dets,
                       int32_t* cand, int32_t* m_out, void* stream);

/* ref: models/utils/mask_roi.py:96-147 (post-NMS list, `max_det` cap by VALUE: `>= image_thresh` keeps ties; dummy row).
 * keep / nkeep: output of vps_nms_batched on `dets`. res: float [8 + 8*kcap]: res[0] = K, res[1] = candidates, res[2] = post-NMS
 * count, res[3] = status (bit 0: candidate overflow, bit 1: more than kcap detections), res[4] = rois considered, then K rows
 * (0, x1, y1, x2, y2, score, class, candidate index). */
int vps_maskroi_finish(const float* dets, const int32_t* cand, const int32_t* m_in, const int32_t* keep, const int32_t* nkeep,
                       int num_classes, int max_det, int kcap, float* res, void* stream);

/* ref: models/detectors/panoptic_fusetrack.py:424-469 (arg-max of comp_scores, greedy assignment with undo, new ids, memory
 * update). comp [K][M+1]; emb [K][E], box [K][ldb], label [K]; prev_emb [>= M+K][E], prev_box [>= M+K][4], prev_label [>= M+K]
 * are updated in place; scratch int32 [M + 3K + 1]; ids [K]; m_out[0] = new memory size. */
int vps_track_assign(const float* comp, int K, int M, const float* emb, int E, const float* box, int ldb, const int64_t* label,
                     float* prev_emb, float* prev_box, int64_t* prev_label, int32_t* scratch, int32_t* ids, int32_t* m_out,
                     void* stream);

/* the end-of-frame record a host reads once per frame, gathered in one launch: tail int32 [8 + 2*kcap] =
 * [kinfo[0..3] (k, masks valid, status, -), mem_count[0] or 0, max over f16_status[0..nslots) or 0, -, -, keep[0..K), ids[0..K) at 8 + kcap]
 * (ids, mem_count, f16_status may be NULL / nslots 0) */
int vps_frame_tail(const int32_t* kinfo, const int32_t* keep, const int32_t* ids, const int32_t* mem_count, const int32_t* f16_status,
                   int nslots, int K, int kcap, int32_t* tail, void* stream);

/* ref: models/utils/mask_removal.py:81-91 (kept list; nothing kept -> [0] with zero logits) + utils/unary_logits.py:96-106
 * (SegTerm crop of boxes*4*0.25). order [n] / flags [n] / tbox [n][4]: MaskRemoval's walk (vps_mask_level), rows [n][8]:
 * vps_maskroi_finish. class_mapping: host int32 [num_classes]. -> inst [n], keep_out [n], k_out[0] = k, k_out[1] = masks valid. */
int vps_pan_instances(const int32_t* order, const int32_t* flags, const float* rows, const int32_t* tbox, int n,
                      const int32_t* class_mapping, int num_classes, vps_pan_inst* inst, int32_t* keep_out, int32_t* k_out,
                      void* stream);

/* ----------------------------------------------------------------------------------------------
 * Panoptic post-processing. Replaces the per-frame body of
 * tools/dataset/cityscapes_vps.py:183-224 (CityscapesVPS.get_unified_pan_result): ~250 boolean masks + np.unique over
 * 2 M pixels per frame on the host become three passes over the uint8 maps on the device. All integer, bit-exact.
 *   pan, seg  uint8 [npix] (pan ids <= id_last_stuff are stuff classes, id_last_stuff+1+j = j-th instance, 255 = void)
 *   hist      int32 [256][256] (instance rows only), pan_count int32 [256]; both zeroed by vps_unify_hist
 *   cls_ind   int32 [k] (panoptic_cls_inds), obj_id int32 [nobj] or NULL (de-duplicated object ids, host side)
 *   tables    uint8 [3][256]: output values of the three channels per pan id
 *   status    int32 [1]: 0 ok, 1 = instance id without cls_ind entry, 2 = without obj_id entry (reference: IndexError)
 *   out       uint8 [npix][3] = (pan_seg, pan_ins, pan_obj) */
int vps_unify_hist(const uint8_t* pan, const uint8_t* seg, int64_t npix, int id_last_stuff, int32_t* hist, int32_t* pan_count,
                   void* stream);
int vps_unify_tables(const int32_t* hist, const int32_t* pan_count, const int32_t* cls_ind, int k, const int32_t* obj_id, int nobj,
                     int id_last_stuff, int64_t stuff_area_limit, uint8_t* tables, int32_t* status, void* stream);
int vps_unify_write(const uint8_t* pan, int64_t npix, const uint8_t* tables, uint8_t* out, void* stream);

/* Second half of the output path: tools/dataset/cityscapes_vps.py:97-159 (converter_2ch_track_core) without its per-segment
 * boolean masks. pan_2ch uint8 [H][W][3] = (pan_seg, pan_ins, pan_obj); a segment is a (pan_seg, pan_obj) pair, key =
 * seg*256 + obj (the reference's 1000*seg + obj).
 *   stats  int32 [65536][5] = (pixel count, xmin, ymin, xmax, ymax), initialised by the call (count 0 = absent)
 *   lut    uint8 [65536][3] colour per key (chosen on the host), out uint8 [npix][3] */
int vps_segment_stats(const uint8_t* pan_2ch, int H, int W, int32_t* stats, void* stream);
int vps_segment_paint(const uint8_t* pan_2ch, int64_t npix, const uint8_t* lut, uint8_t* out, void* stream);

/* VPQ evaluation: confusion counts of tools/eval_vpq.py:150-157 for ONE frame.
 *   gt_rgb, pred_rgb  uint8 [npix][3] panoptic PNGs (segment id = R + 256 G + 65536 B)
 *   gt_ids, pred_ids  sorted unique uint32 ids (device), typically the ids of the frame's segments_info plus 0 (VOID)
 *   counts            int32 [ngt+1][npred+1], zeroed by the call; row / column n collects the ids that are not listed */
int vps_pair_count(const uint8_t* gt_rgb, const uint8_t* pred_rgb, int64_t npix, const uint32_t* gt_ids, int ngt,
                   const uint32_t* pred_ids, int npred, int32_t* counts, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Input preparation: Normalize -> Pad(size_divisor) -> ImageToTensor of the test pipeline in one pass.
 * Replaces mmdet/datasets/pipelines/transforms.py:258-269, :310-318 and formating.py:52-67 (mmcv 0.2.14 imnormalize,
 * impad_to_multiple) for a decoded uint8 [H][W][3] image already on the device.
 *   mean, std  HOST pointers to 3 floats (channel order of the OUTPUT, i.e. RGB when to_rgb)
 *   out        fp32 [3][Hp][Wp], (c - mean) / std in fp32, bottom/right padding = pad_val. Bit-exact with NumPy. */
/* Resize of the same pipeline (transforms.py:107-122 -> mmcv.imrescale -> cv2.resize, INTER_LINEAR) on the decoded uint8 image:
 * OpenCV's 8-bit fixed-point bilinear (11-bit coefficients), or the area average when the size is halved exactly.
 *   xtab / ytab  DEVICE int32 [W][3] / [H][3]: source index, weight of it, weight of its successor (weights sum to 2048);
 *                unused (may be NULL) for the exact 2x shrink
 *   src / dst    uint8 [H0][W0][C] -> [H][W][C], C <= 4 */
int vps_resize_u8(const uint8_t* src, int H0, int W0, uint8_t* dst, int H, int W, int C, const int32_t* xtab,
                  const int32_t* ytab, void* stream);

int vps_image_prep(const uint8_t* img, int H, int W, int Hp, int Wp, const float* mean, const float* std, int to_rgb,
                   float pad_val, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VPS_HIP_H */
