// TEST INFRASTRUCTURE ONLY (see oracle/build_ref.py): raw-pointer launchers around the REFERENCE's own kernel bodies, which
// build_ref.py puts on the include path as ref_*.inc fragments taken from /root/reference at build time. This file replaces
// the ATen launchers of the reference extensions (at::Tensor plumbing, torch-1.4 dispatch macros) and nothing else: grid and
// block shapes, scratch-tensor zero fills and output geometry are the reference's, cited per function.
// All tensors are contiguous NCHW fp32 device buffers, like the reference's (it asserts / forces contiguity).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

// ---- CUDA-isms used by the kernel bodies -----------------------------------------------------------------------------
// correlation_cuda_kernel.cu:6-7,17-21,129: 32-thread blocks reduced with warp shuffles over offsets 16..1. One such block is
// the lower half of a wave64; a width-32 shuffle reproduces the CUDA warp's partial-sum tree exactly.
#ifdef warpSize
#undef warpSize
#endif
#define warpSize 32
#define __shfl_down_sync(mask, val, offset) __shfl_down((val), (offset), 32)
#define __syncwarp() ((void)0)
template <typename T>
__host__ __device__ inline T THCCeilDiv(T a, T b) { return (a + b - 1) / b; }   // THC/THCDeviceUtils.cuh

namespace ref_corr {
#include "ref_correlation.inc"
}
namespace ref_flow {
#include "ref_resample2d.inc"
#include "ref_channelnorm.inc"
}
namespace ref_roi {
#include "ref_roi_align.inc"
}
namespace ref_dcn {
#include "ref_deform.inc"
}
namespace ref_nms {
#include "ref_nms.inc"
}

static int status() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(int)e;
    e = hipDeviceSynchronize();
    return e == hipSuccess ? 0 : -(int)e;
}

// correlation_cuda.cc:10-87 + correlation_cuda_kernel.cu:383-405. r1, r2: caller scratch [B, H+2p, W+2p, C] (zero-filled here
// like rInput.fill_(0)); out [B, D*D, outH, outW]. Returns the output height/width through oh/ow.
extern "C" int ref_correlation_forward(const float* in1, const float* in2, float* r1, float* r2, float* out, int B, int C, int H, int W,
                                       int pad_size, int kernel_size, int max_displacement, int stride1, int stride2, int* oh, int* ow) {
    const int kernel_radius = (kernel_size - 1) / 2;
    const int border_radius = kernel_radius + max_displacement;
    const int pH = H + 2 * pad_size, pW = W + 2 * pad_size;
    const int D = (max_displacement / stride2) * 2 + 1;
    const int outH = (int)ceil(static_cast<float>(pH - 2 * border_radius) / static_cast<float>(stride1));
    const int outW = (int)ceil(static_cast<float>(pW - 2 * border_radius) / static_cast<float>(stride1));
    *oh = outH; *ow = outW;
    if (!out) return 0;                          // geometry query
    hipMemset(r1, 0, sizeof(float) * (size_t)B * pH * pW * C);
    hipMemset(r2, 0, sizeof(float) * (size_t)B * pH * pW * C);
    hipMemset(out, 0, sizeof(float) * (size_t)B * D * D * outH * outW);
    dim3 blocks_grid(B, H, W), threads_block(THREADS_PER_BLOCK);
    ref_corr::channels_first<float><<<blocks_grid, threads_block>>>(in1, r1, C, H, W, pad_size);
    ref_corr::channels_first<float><<<blocks_grid, threads_block>>>(in2, r2, C, H, W, pad_size);
    dim3 totalBlocksCorr(B, outH, outW);
    ref_corr::correlation_forward<float><<<totalBlocksCorr, threads_block>>>(out, D * D, outH, outW, r1, C, H, W, r2, pad_size, kernel_size,
                                                                            max_displacement, stride1, stride2);
    return status();
}

static long4 sizes4(int a, int b, int c, int d) { return make_long4(a, b, c, d); }
static long4 strides4(int a, int b, int c, int d) { return make_long4((long)b * c * d, (long)c * d, d, 1); }

// resample2d_kernel.cu:200-242: one thread per output element, 512 threads per block
extern "C" int ref_resample2d_forward(const float* in1, const float* flow, float* out, int B, int C, int H, int W, int kernel_size,
                                      int bilinear) {
    const int n = B * C * H * W;
    ref_flow::kernel_resample2d_update_output<float><<<(n + CUDA_NUM_THREADS - 1) / CUDA_NUM_THREADS, CUDA_NUM_THREADS>>>(
        n, in1, sizes4(B, C, H, W), strides4(B, C, H, W), flow, sizes4(B, 2, H, W), strides4(B, 2, H, W), out, sizes4(B, C, H, W),
        strides4(B, C, H, W), kernel_size, bilinear != 0);
    return status();
}

// channelnorm_kernel.cu:100-125
extern "C" int ref_channelnorm_forward(const float* in1, float* out, int B, int C, int H, int W, int norm_deg) {
    const int n = B * H * W;
    ref_flow::kernel_channelnorm_update_output<float><<<(n + CUDA_NUM_THREADS - 1) / CUDA_NUM_THREADS, CUDA_NUM_THREADS>>>(
        n, in1, sizes4(B, C, H, W), strides4(B, C, H, W), out, sizes4(B, 1, H, W), strides4(B, 1, H, W), norm_deg);
    return status();
}

// roi_align_kernel.cu:8-14,126-147: grid-stride, 1024 threads, <= 65000 blocks. rois [R,5] = (batch, x1, y1, x2, y2)
extern "C" int ref_roi_align_forward(const float* feat, const float* rois, float spatial_scale, int sample_num, int channels, int height,
                                     int width, int num_rois, int pooled_h, int pooled_w, float* out) {
    const int output_size = num_rois * pooled_h * pooled_w * channels;
    int blocks = (output_size + 1024 - 1) / 1024;
    if (blocks > 65000) blocks = 65000;
    ref_roi::ROIAlignForward<float><<<blocks, 1024>>>(output_size, feat, rois, spatial_scale, sample_num, channels, height, width, pooled_h,
                                                     pooled_w, out);
    return status();
}

// deform_conv_cuda_kernel.cu:244-276 (deformable_im2col): columns [C*kh*kw, B*Hc*Wc]
extern "C" int ref_deformable_im2col(const float* im, const float* offset, int channels, int height, int width, int ksize_h, int ksize_w,
                                     int pad_h, int pad_w, int stride_h, int stride_w, int dilation_h, int dilation_w, int parallel_imgs,
                                     int deformable_group, float* col) {
    const int height_col = (height + 2 * pad_h - (dilation_h * (ksize_h - 1) + 1)) / stride_h + 1;
    const int width_col = (width + 2 * pad_w - (dilation_w * (ksize_w - 1) + 1)) / stride_w + 1;
    const int num_kernels = channels * height_col * width_col * parallel_imgs;
    const int channel_per_deformable_group = channels / deformable_group;
    int blocks = (num_kernels + 1024 - 1) / 1024;
    if (blocks > 65535) blocks = 65535;
    ref_dcn::deformable_im2col_gpu_kernel<float><<<blocks, 1024>>>(num_kernels, im, offset, height, width, ksize_h, ksize_w, pad_h, pad_w,
                                                                  stride_h, stride_w, dilation_h, dilation_w, channel_per_deformable_group,
                                                                  parallel_imgs, channels, deformable_group, height_col, width_col, col);
    return status();
}

// mmdet/ops/nms/src/nms_kernel.cu:70-130 (nms_cuda) from the point where the boxes are score-sorted (the sort and the final
// index mapping are torch calls made by the test): boxes_sorted [n,5] on the device -> kept positions (ascending) in keep_out.
extern "C" int ref_mmdet_nms_sorted(const float* boxes_dev, int boxes_num, float nms_overlap_thresh, long long* keep_out_ll, int* num_out) {
    using namespace ref_nms;
    const int col_blocks = THCCeilDiv(boxes_num, threadsPerBlock);
    unsigned long long* mask_dev = nullptr;
    if (hipMalloc(&mask_dev, sizeof(unsigned long long) * (size_t)boxes_num * col_blocks) != hipSuccess) return -2;
    dim3 blocks(THCCeilDiv(boxes_num, threadsPerBlock), THCCeilDiv(boxes_num, threadsPerBlock));
    dim3 threads(threadsPerBlock);
    nms_kernel<<<blocks, threads>>>(boxes_num, nms_overlap_thresh, boxes_dev, mask_dev);
    std::vector<unsigned long long> mask_host((size_t)boxes_num * col_blocks);
    hipError_t e = hipMemcpy(&mask_host[0], mask_dev, sizeof(unsigned long long) * (size_t)boxes_num * col_blocks, hipMemcpyDeviceToHost);
    hipFree(mask_dev);
    if (e != hipSuccess) return -(int)e;
    std::vector<unsigned long long> remv(col_blocks);
    memset(&remv[0], 0, sizeof(unsigned long long) * col_blocks);
    int64_t* keep_out = reinterpret_cast<int64_t*>(keep_out_ll);
    // the reference's own greedy pass (nms_kernel.cu:111-123), statements included verbatim from the reference file
#include "ref_nms_reduce.inc"
    *num_out = num_to_keep;
    return 0;
}
