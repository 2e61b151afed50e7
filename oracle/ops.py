"""Oracle restatements of the reference's CUDA-only operators and of cv2.resize (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: none of these can be executed from the reference in this environment (CUDA-only sources, no
binaries, no cv2); they follow the cited .cu/.cpp sources line by line. All paths relative to /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------------
# mmdet/models/flow_modules/correlation_package/correlation_cuda.cc:18-38 (output geometry) and
# correlation_cuda_kernel.cu:73-147 (kernel): kernel_size=1, stride1=1, pad_size == max_displacement.
# out[n, tc, y, x] = sum_c in1[n,c,y,x] * in2[n,c,y+tj*s2,x+ti*s2] / (k*k*C), tc = (tj+r)*(2r+1) + (ti+r)
# ----------------------------------------------------------------------------------------------------
def correlation(in1, in2, pad_size, kernel_size, max_displacement, stride1, stride2):
    assert kernel_size == 1 and stride1 == 1 and pad_size == max_displacement
    n, c, h, w = in1.shape
    r = max_displacement // stride2
    d = 2 * r + 1
    p2 = F.pad(in2, (pad_size, pad_size, pad_size, pad_size))
    out = in1.new_zeros(n, d * d, h, w)
    nelems = float(kernel_size * kernel_size * c)
    for tj in range(-r, r + 1):
        for ti in range(-r, r + 1):
            y0 = pad_size + tj * stride2
            x0 = pad_size + ti * stride2
            sh = p2[:, :, y0:y0 + h, x0:x0 + w]
            out[:, (tj + r) * d + (ti + r)] = (in1 * sh).sum(1) / nelems
    return out


# ----------------------------------------------------------------------------------------------------
# resample2d_package/resample2d_kernel.cu:15-72 (kernel_size=1, bilinear=True): border-clamped taps, the
# weights are evaluated in double ((1. - alpha) ...) and each product is rounded to float before the add.
# ----------------------------------------------------------------------------------------------------
def resample2d(in1, flow):
    b, c, h, w = in1.shape
    assert flow.shape == (b, 2, h, w)
    xs = torch.arange(w, dtype=torch.float32).view(1, 1, w)
    ys = torch.arange(h, dtype=torch.float32).view(1, h, 1)
    xf = xs + flow[:, 0]
    yf = ys + flow[:, 1]
    fx, fy = torch.floor(xf), torch.floor(yf)
    alpha = (xf - fx).double().unsqueeze(1)
    beta = (yf - fy).double().unsqueeze(1)
    xL = fx.long().clamp(0, w - 1)
    xR = (fx + 1).long().clamp(0, w - 1)
    yT = fy.long().clamp(0, h - 1)
    yB = (fy + 1).long().clamp(0, h - 1)
    flat = in1.reshape(b, c, h * w)

    def tap(yy, xx):
        idx = (yy * w + xx).view(b, 1, h * w).expand(b, c, h * w)
        return flat.gather(2, idx).view(b, c, h, w).double()

    val = torch.zeros(b, c, h, w, dtype=torch.float32)
    val = val + ((1. - alpha) * (1. - beta) * tap(yT, xL)).float()
    val = val + ((alpha) * (1. - beta) * tap(yT, xR)).float()
    val = val + ((1. - alpha) * (beta) * tap(yB, xL)).float()
    val = val + ((alpha) * (beta) * tap(yB, xR)).float()
    return val


# channelnorm_package/channelnorm_kernel.cu:51-59
def channelnorm(x):
    acc = torch.zeros_like(x[:, 0])
    for c in range(x.shape[1]):
        acc = acc + x[:, c] * x[:, c]
    return acc.sqrt().unsqueeze(1)


# ----------------------------------------------------------------------------------------------------
# mmdet/ops/roi_align/src/roi_align_kernel.cu:16-124. feat [B,C,H,W], rois [R,5] (b,x1,y1,x2,y2)
# ----------------------------------------------------------------------------------------------------
def roi_align(feat, rois, out_size, spatial_scale, sample_num):
    r = rois.shape[0]
    _, c, h, w = feat.shape
    p = out_size
    if r == 0:
        return feat.new_zeros(0, c, p, p)
    f32 = torch.float32
    ss = torch.tensor(spatial_scale, dtype=f32)
    bidx = rois[:, 0].long()
    start_w = rois[:, 1] * ss
    start_h = rois[:, 2] * ss
    end_w = (rois[:, 3] + 1) * ss
    end_h = (rois[:, 4] + 1) * ss
    roi_w = (end_w - start_w).clamp(min=0.)
    roi_h = (end_h - start_h).clamp(min=0.)
    bin_h = roi_h / p
    bin_w = roi_w / p
    ph = torch.arange(p, dtype=f32)
    it = torch.arange(sample_num, dtype=f32)
    # y[r, ph, iy] = start_h + ph*bin_h + (iy+.5)*bin_h/sample_num
    y = start_h.view(r, 1, 1) + ph.view(1, p, 1) * bin_h.view(r, 1, 1) + \
        (it.view(1, 1, -1) + .5) * bin_h.view(r, 1, 1) / float(sample_num)
    x = start_w.view(r, 1, 1) + ph.view(1, p, 1) * bin_w.view(r, 1, 1) + \
        (it.view(1, 1, -1) + .5) * bin_w.view(r, 1, 1) / float(sample_num)
    y = y.reshape(r, p * sample_num)
    x = x.reshape(r, p * sample_num)
    ny, nx = y.shape[1], x.shape[1]
    Y = y.view(r, ny, 1).expand(r, ny, nx)
    X = x.view(r, 1, nx).expand(r, ny, nx)
    oob = (Y < -1.0) | (Y > h) | (X < -1.0) | (X > w)
    Yc = Y.clamp(min=0.)
    Xc = X.clamp(min=0.)
    y_low = Yc.long()
    x_low = Xc.long()
    ytop = y_low >= h - 1
    xtop = x_low >= w - 1
    y_low = torch.where(ytop, torch.full_like(y_low, h - 1), y_low)
    x_low = torch.where(xtop, torch.full_like(x_low, w - 1), x_low)
    y_high = torch.where(ytop, y_low, y_low + 1)
    x_high = torch.where(xtop, x_low, x_low + 1)
    Yc = torch.where(ytop, y_low.float(), Yc)
    Xc = torch.where(xtop, x_low.float(), Xc)
    ly = Yc - y_low.float()
    lx = Xc - x_low.float()
    hy = 1. - ly
    hx = 1. - lx
    out = feat.new_zeros(r, c, ny, nx)
    for i in range(r):
        fm = feat[bidx[i]].reshape(c, h * w)

        def g(yy, xx):
            return fm[:, (yy[i] * w + xx[i]).reshape(-1)].view(c, ny, nx)

        w1 = (hy[i] * hx[i]); w2 = (hy[i] * lx[i]); w3 = (ly[i] * hx[i]); w4 = (ly[i] * lx[i])
        v = w1 * g(y_low, x_low) + w2 * g(y_low, x_high) + w3 * g(y_high, x_low) + w4 * g(y_high, x_high)
        out[i] = torch.where(oob[i].unsqueeze(0), torch.zeros_like(v), v)
    out = out.view(r, c, p, sample_num, p, sample_num)
    # accumulate in the kernel's order (iy outer, ix inner), then divide
    acc = feat.new_zeros(r, c, p, p)
    for iy in range(sample_num):
        for ix in range(sample_num):
            acc = acc + out[:, :, :, iy, :, ix]
    return acc / float(sample_num * sample_num)


# ----------------------------------------------------------------------------------------------------
# mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:83-113 (bilinear, per-corner zeroing), :189-241 (im2col,
# offset channel 2*(i*kw+j) = dh, +1 = dw, valid iff -1 < h < H and -1 < w < W) and
# deform_conv_cuda.cpp:200-236 (GEMM weight[Cout, C*kh*kw] x columns). groups = deformable_groups = 1.
# ----------------------------------------------------------------------------------------------------
def deform_conv(x, offset, weight, stride=1, padding=1, dilation=1):
    n, c, h, w = x.shape
    co, ci, kh, kw = weight.shape
    assert ci == c
    ho = (h + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    wo = (w + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    assert offset.shape == (n, 2 * kh * kw, ho, wo)
    outs = []
    hs = (torch.arange(ho, dtype=torch.float32) * stride - padding).view(ho, 1)
    ws = (torch.arange(wo, dtype=torch.float32) * stride - padding).view(1, wo)
    for b in range(n):
        flat = x[b].reshape(c, h * w)
        cols = []
        for i in range(kh):
            for j in range(kw):
                off_h = offset[b, 2 * (i * kw + j)]
                off_w = offset[b, 2 * (i * kw + j) + 1]
                h_im = hs + i * dilation + off_h
                w_im = ws + j * dilation + off_w
                valid = (h_im > -1) & (w_im > -1) & (h_im < h) & (w_im < w)
                h_low = torch.floor(h_im)
                w_low = torch.floor(w_im)
                lh = h_im - h_low
                lw = w_im - w_low
                hh = 1 - lh
                hw = 1 - lw
                h_low = h_low.long(); w_low = w_low.long()
                h_high = h_low + 1; w_high = w_low + 1

                def corner(yy, xx, ok):
                    ok = ok & valid
                    idx = (yy.clamp(0, h - 1) * w + xx.clamp(0, w - 1)).reshape(-1)
                    v = flat[:, idx].view(c, ho, wo)
                    return torch.where(ok.unsqueeze(0), v, torch.zeros_like(v))

                v1 = corner(h_low, w_low, (h_low >= 0) & (w_low >= 0))
                v2 = corner(h_low, w_high, (h_low >= 0) & (w_high <= w - 1))
                v3 = corner(h_high, w_low, (h_high <= h - 1) & (w_low >= 0))
                v4 = corner(h_high, w_high, (h_high <= h - 1) & (w_high <= w - 1))
                w1 = hh * hw; w2 = hh * lw; w3 = lh * hw; w4 = lh * lw
                cols.append(w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4)          # [c, ho, wo]
        col = torch.stack(cols, dim=1).reshape(c * kh * kw, ho * wo)    # row = c*kh*kw + i*kw + j
        outs.append((weight.reshape(co, -1) @ col).view(co, ho, wo))
    return torch.stack(outs, 0)


# ----------------------------------------------------------------------------------------------------
# NMS. IoU per mmdet/ops/nms/src/nms_kernel.cu:13-21 (== utils/upsnet/nms/nms_kernel.cu devIoU): float32,
# +1 convention; suppression iff IoU > thr (strict); greedy over the score-descending order (:99-123).
# ----------------------------------------------------------------------------------------------------
def _greedy_nms_sorted(boxes, thr):
    """boxes: float32 [n,4] already in descending-score order -> kept positions (ascending)."""
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    b = boxes.astype(np.float32)
    one = np.float32(1.0)
    area = (b[:, 2] - b[:, 0] + one) * (b[:, 3] - b[:, 1] + one)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(thr)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 >= n:
            break
        left = np.maximum(b[i, 0], b[i + 1:, 0]); right = np.minimum(b[i, 2], b[i + 1:, 2])
        top = np.maximum(b[i, 1], b[i + 1:, 1]); bottom = np.minimum(b[i, 3], b[i + 1:, 3])
        ww = np.maximum(right - left + one, np.float32(0)); hh = np.maximum(bottom - top + one, np.float32(0))
        inter = ww * hh
        iou = inter / (area[i] + area[i + 1:] - inter)
        suppressed[i + 1:] |= iou > thr
    return np.asarray(keep, dtype=np.int64)


def nms_mmdet(dets, thr):
    """mmdet/ops/nms/nms_wrapper.py:8-49 + src/nms_kernel.cu:70-130 (nms_cuda): dets torch [n,5].
    Returns (dets[inds], inds) with inds sorted ASCENDING in the original indexing (:127-130)."""
    if dets.shape[0] == 0:
        return dets, dets.new_zeros(0, dtype=torch.long)
    order = torch.sort(dets[:, 4], descending=True, stable=True)[1]
    keep = _greedy_nms_sorted(dets[order, :4].numpy(), thr)
    inds = torch.sort(order[torch.from_numpy(keep)])[0]
    return dets[inds], inds


def nms_upsnet(dets, thr):
    """utils/upsnet/nms/gpu_nms.pyx:23-38 + nms_kernel.cu:97-150: dets numpy float32 [n,5] -> list(order[keep])
    (kept ORIGINAL indices in descending-score order). `scores.argsort()[::-1]` as in the reference."""
    order = dets[:, 4].argsort()[::-1]
    keep = _greedy_nms_sorted(dets[order, :4], thr)
    return list(order[keep])


# ----------------------------------------------------------------------------------------------------
# cv2.resize(src float32 [h0,w0], (w,h)) with the default INTER_LINEAR (used by utils/mask_removal.py:67).
# OpenCV imgproc resize.cpp: scale = src/dst in double, fx = (float)((dx+0.5)*scale-0.5), sx=floor(fx),
# fx-=sx, (sx<0 -> sx=0,fx=0), (sx>=src-1 -> sx=src-1,fx=0); float weights; horizontal pass then vertical.
# ----------------------------------------------------------------------------------------------------
def _cv_lin_coords(dst, src):
    scale = float(src) / float(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0; s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0; s[hi] = src - 1
    s1 = np.minimum(s + 1, src - 1)
    return s, s1, f


def cv2_resize_linear(src, dsize):
    w, h = dsize
    src = np.asarray(src, dtype=np.float32)
    h0, w0 = src.shape
    x0, x1, fx = _cv_lin_coords(w, w0)
    y0, y1, fy = _cv_lin_coords(h, h0)
    one = np.float32(1)
    rows = src[:, x0] * (one - fx)[None, :] + src[:, x1] * fx[None, :]      # [h0, w]
    out = rows[y0, :] * (one - fy)[:, None] + rows[y1, :] * fy[:, None]
    return out.astype(np.float32)
