"""TEST INFRASTRUCTURE ONLY — ctypes loader of oracle/_ref/libvpsref.so, the reference's OWN native kernels compiled for
gfx950 by oracle/build_ref.py (correlation_cuda_kernel.cu, resample2d_kernel.cu, channelnorm_kernel.cu, roi_align_kernel.cu,
deform_conv_cuda_kernel.cu, ops/nms/src/nms_kernel.cu, utils/upsnet/nms/nms_kernel.cu). The wrappers take / return
contiguous NCHW fp32 torch CUDA tensors like the reference's Python `Function`s. Only tests import this module; the product
never does. The functions synchronise (the reference kernels run on the null stream).
"""
import ctypes
import os

import numpy as np
import torch

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'libvpsref.so')
_lib = None


def available():
    return os.path.exists(_PATH)


def load():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError('%s missing: run `python oracle/build_ref.py` where /root/reference is mounted (it travels to '
                               'the GPU box with the snapshot)' % _PATH)
        _lib = ctypes.CDLL(_PATH)
    return _lib


def _p(t):
    assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.int64, torch.int32)
    return ctypes.c_void_p(t.data_ptr())


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed: %d' % (what, rc))


def correlation(in1, in2, pad_size, kernel_size, max_displacement, stride1, stride2):
    """correlation_package/correlation.py:47-61 -> correlation_cuda.cc:10-87"""
    lib = load()
    in1, in2 = in1.contiguous(), in2.contiguous()
    B, C, H, W = in1.shape
    oh, ow = ctypes.c_int(), ctypes.c_int()
    lib.ref_correlation_forward(None, None, None, None, None, B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2,
                                ctypes.byref(oh), ctypes.byref(ow))
    D = (max_displacement // stride2) * 2 + 1
    r1 = torch.empty(B, H + 2 * pad_size, W + 2 * pad_size, C, device=in1.device)
    r2 = torch.empty_like(r1)
    out = torch.empty(B, D * D, oh.value, ow.value, device=in1.device)
    torch.cuda.synchronize()
    _ck(lib.ref_correlation_forward(_p(in1), _p(in2), _p(r1), _p(r2), _p(out), B, C, H, W, pad_size, kernel_size, max_displacement,
                                    stride1, stride2, ctypes.byref(oh), ctypes.byref(ow)), 'ref_correlation_forward')
    return out


def resample2d(in1, flow, kernel_size=1, bilinear=True):
    """resample2d_package/resample2d.py:40-49"""
    in1, flow = in1.contiguous(), flow.contiguous()
    B, C, H, W = in1.shape
    out = torch.empty_like(in1)
    torch.cuda.synchronize()
    _ck(load().ref_resample2d_forward(_p(in1), _p(flow), _p(out), B, C, H, W, kernel_size, int(bilinear)), 'ref_resample2d_forward')
    return out


def channelnorm(x, norm_deg=2):
    """channelnorm_package/channelnorm.py:31-38"""
    x = x.contiguous()
    B, C, H, W = x.shape
    out = torch.empty(B, 1, H, W, device=x.device)
    torch.cuda.synchronize()
    _ck(load().ref_channelnorm_forward(_p(x), _p(out), B, C, H, W, norm_deg), 'ref_channelnorm_forward')
    return out


def roi_align(feat, rois, out_size, spatial_scale, sample_num):
    """mmdet/ops/roi_align/roi_align.py:9-33 (RoIAlignFunction.forward) on one feature level"""
    feat, rois = feat.contiguous(), rois.contiguous()
    _, C, H, W = feat.shape
    R = rois.shape[0]
    out = torch.zeros(R, C, out_size, out_size, device=feat.device)
    if R:
        torch.cuda.synchronize()
        _ck(load().ref_roi_align_forward(_p(feat), _p(rois), ctypes.c_float(spatial_scale), sample_num, C, H, W, R, out_size, out_size,
                                         _p(out)), 'ref_roi_align_forward')
    return out


def deform_conv(x, offset, weight, stride=1, padding=1, dilation=1):
    """deform_conv_cuda.cpp:151-250 with im2col_step = 1, groups = deformable_groups = 1: the reference's own im2col kernel,
    then the GEMM weight[Cout, C*kh*kw] x columns (torch.matmul on the device stands in for ATen addmm_)."""
    x, offset = x.contiguous(), offset.contiguous()
    N, C, H, W = x.shape
    co, ci, kh, kw = weight.shape
    ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    outs = []
    for b in range(N):
        col = torch.empty(C * kh * kw, ho * wo, device=x.device)
        torch.cuda.synchronize()
        _ck(load().ref_deformable_im2col(_p(x[b:b + 1]), _p(offset[b:b + 1].contiguous()), C, H, W, kh, kw, padding, padding, stride, stride,
                                         dilation, dilation, 1, 1, _p(col)), 'ref_deformable_im2col')
        outs.append((weight.reshape(co, -1).double() @ col.double()).float().view(co, ho, wo))
    return torch.stack(outs, 0)


def deformable_im2col(x, offset, kh=3, kw=3, stride=1, padding=1, dilation=1):
    """just the column buffer [C*kh*kw, Ho*Wo] of one image (batch 1)"""
    x, offset = x.contiguous(), offset.contiguous()
    _, C, H, W = x.shape
    ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    col = torch.empty(C * kh * kw, ho * wo, device=x.device)
    torch.cuda.synchronize()
    _ck(load().ref_deformable_im2col(_p(x), _p(offset), C, H, W, kh, kw, padding, padding, stride, stride, dilation, dilation, 1, 1,
                                     _p(col)), 'ref_deformable_im2col')
    return col


def nms_mmdet(dets, thr):
    """mmdet/ops/nms/nms_wrapper.py:8-49 + src/nms_kernel.cu:70-130: dets [n,5] CUDA tensor -> (dets[inds], inds ascending)"""
    if dets.shape[0] == 0:
        return dets, dets.new_zeros(0, dtype=torch.long)
    n = dets.shape[0]
    # nms_wrapper.py: `scores.sort(0, descending=True)` — torch's default sort is NOT stable, so the reference's order among EXACTLY
    # equal scores is whatever its sort implementation produces (unspecified). This harness, the oracle (oracle/ops.py:nms_mmdet)
    # and the product (vps_amd/heads.py, operators.py) all pin the stable order; the 'ties' case of test_ref_native_gpu.py therefore
    # checks the compiled kernel's suppression logic on tied SCORES under that one order, not the reference's tie order.
    order = torch.sort(dets[:, 4], descending=True, stable=True)[1]
    boxes_sorted = dets.index_select(0, order).contiguous()
    keep = np.zeros(n, dtype=np.int64)
    num = ctypes.c_int()
    torch.cuda.synchronize()
    _ck(load().ref_mmdet_nms_sorted(_p(boxes_sorted), n, ctypes.c_float(thr), keep.ctypes.data_as(ctypes.c_void_p), ctypes.byref(num)),
        'ref_mmdet_nms_sorted')
    k = torch.from_numpy(keep[:num.value]).to(dets.device)
    inds = torch.sort(order[k])[0]
    return dets[inds], inds


def nms_upsnet(dets, thr, device_id=0):
    """utils/upsnet/nms/gpu_nms.pyx:23-38 around the reference's own `_nms` (nms_kernel.cu:97-150, compiled from its file):
    dets numpy float32 [n,5] on the HOST -> list(order[keep])"""
    lib = load()
    fn = getattr(lib, '_Z4_nmsPiS_PKfiifi')
    fn.restype = None
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n = dets.shape[0]
    keep = np.zeros(n, dtype=np.int32)
    num = ctypes.c_int()
    order = dets[:, 4].argsort()[::-1]
    sorted_dets = np.ascontiguousarray(dets[order, :])
    torch.cuda.synchronize()
    fn(keep.ctypes.data_as(ctypes.c_void_p), ctypes.byref(num), sorted_dets.ctypes.data_as(ctypes.c_void_p), n, 5, ctypes.c_float(thr),
       device_id)
    return list(order[keep[:num.value]])
