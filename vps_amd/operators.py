"""Operator-level drop-ins with the reference's class names and (NCHW) signatures, each a thin call into libvpship:
`mmdet.ops.RoIAlign / DeformConv / nms`, `flow_modules.*_package.{Correlation, Resample2d, ChannelNorm}`,
`utils.upsnet.nms.nms.gpu_nms_wrapper`. The detector itself stays NHWC and does not use these wrappers.
"""
import numpy as np
import torch
import torch.nn as nn

from . import hip, nhwc


class Resample2d(nn.Module):
    """resample2d_package/resample2d.py:40-49"""

    def __init__(self, kernel_size=1, bilinear=True):
        super().__init__()
        assert kernel_size == 1 and bilinear, 'only kernel_size=1, bilinear=True is on the path (flownet2.py)'

    def forward(self, input1, input2):
        input1 = input1.contiguous()
        b, c, h, w = input1.shape
        out = torch.empty_like(input1)
        hip.check(hip.load().vps_resample2d(hip.tensor4_nchw(input1), hip.tensor4_nchw(input2), hip.tensor4_nchw(out),
                                            b, c, h, w, hip.stream_ptr()), 'vps_resample2d')
        return out


class ChannelNorm(nn.Module):
    """channelnorm_package/channelnorm.py:31-38"""

    def __init__(self, norm_deg=2):
        super().__init__()
        assert norm_deg == 2

    def forward(self, input1):
        b, c, h, w = input1.shape
        out = torch.empty(b, 1, h, w, dtype=input1.dtype, device=input1.device)
        hip.check(hip.load().vps_channelnorm(hip.tensor4_nchw(input1), hip.tensor4_nchw(out), b, c, h, w, hip.stream_ptr()),
                  'vps_channelnorm')
        return out


class Correlation(nn.Module):
    """correlation_package/correlation.py:47-61 (kernel_size=1, stride1=1, pad_size == max_displacement)"""

    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1):
        super().__init__()
        assert kernel_size == 1 and stride1 == 1 and pad_size == max_displacement
        self.max_displacement, self.stride2 = max_displacement, stride2

    def forward(self, input1, input2):
        n, c, h, w = input1.shape
        a = nhwc.from_nchw(input1.contiguous()); b = nhwc.from_nchw(input2.contiguous())
        d = (2 * (self.max_displacement // self.stride2) + 1) ** 2
        out = nhwc.FMap(torch.empty(n, h, w, (d + 3) // 4 * 4, dtype=torch.float32, device=input1.device), d, 0)
        nhwc.correlation(a, b, out, self.max_displacement, self.stride2)
        return out.to_nchw()


class RoIAlign(nn.Module):
    """mmdet/ops/roi_align/roi_align.py:59-87 (single feature level)"""

    def __init__(self, out_size, spatial_scale, sample_num=0, use_torchvision=False):
        super().__init__()
        self.out_size = out_size if isinstance(out_size, int) else out_size[0]
        self.spatial_scale, self.sample_num = float(spatial_scale), int(sample_num)
        assert self.sample_num > 0

    def forward(self, features, rois):
        fm = nhwc.from_nchw(features.contiguous())
        out = nhwc.roi_align([fm], [1.0 / self.spatial_scale], rois.contiguous(), self.out_size, self.sample_num,
                             finest_scale=1e30)        # one level: every roi maps to level 0
        return out.permute(0, 3, 1, 2).contiguous()


class DeformConv(nn.Module):
    """mmdet/ops/dcn/deform_conv.py DeformConv (groups = deformable_groups = 1, bias=False)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        assert not bias and groups == 1 and deformable_groups == 1 and dilation == 1
        self.stride, self.padding = stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size).normal_(0, 0.01))
        self._packed = None

    def forward(self, x, offset):
        key = (self.weight._version, x.device)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, nhwc.PackedConv(self.weight, None, None, self.stride, self.padding, device=x.device, deform=True))
        ws = nhwc.Workspace(x.device)
        out = self._packed[1](nhwc.from_nchw(x.contiguous()), ws=ws, name='o', offset=nhwc.from_nchw(offset.contiguous()))
        return out.to_nchw()


def nms(dets, iou_thr, device_id=None):
    """mmdet/ops/nms/nms_wrapper.py:8-49: dets [n,5] device tensor -> (dets[inds], inds), inds ascending."""
    if dets.shape[0] == 0:
        return dets, dets.new_zeros(0, dtype=torch.long)
    n = dets.shape[0]
    order = torch.sort(dets[:, 4], descending=True, stable=True)[1]
    sorted_dets = dets[order].contiguous()
    dev = dets.device
    cb = (n + 63) // 64
    mask = torch.empty(n * cb, dtype=torch.int64, device=dev)
    keep = torch.empty(n, dtype=torch.int32, device=dev)
    nkeep = torch.zeros(1, dtype=torch.int32, device=dev)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    hip.check(hip.load().vps_nms_batched(hip.ptr(sorted_dets), 1, n, hip.ptr(cnt), float(iou_thr), hip.ptr(mask), hip.ptr(keep),
                                         hip.ptr(nkeep), hip.stream_ptr()), 'vps_nms_batched')
    k = keep[:int(nkeep.item())].long()
    inds = torch.sort(order[k])[0]
    return dets[inds, :], inds


def gpu_nms_wrapper(thresh, device_id=0):
    """utils/upsnet/nms/nms.py:40-43: numpy float32 dets -> list of kept original indices in descending-score order"""
    def _nms(dets):
        dev = torch.device('cuda', device_id)
        order = dets[:, 4].argsort()[::-1]
        t = torch.from_numpy(np.ascontiguousarray(dets[order])).to(dev)
        n = t.shape[0]
        cb = (n + 63) // 64
        mask = torch.empty(n * cb, dtype=torch.int64, device=dev)
        keep = torch.empty(n, dtype=torch.int32, device=dev)
        nkeep = torch.zeros(1, dtype=torch.int32, device=dev)
        cnt = torch.tensor([n], dtype=torch.int32, device=dev)
        hip.check(hip.load().vps_nms_batched(hip.ptr(t), 1, n, hip.ptr(cnt), float(thresh), hip.ptr(mask), hip.ptr(keep),
                                             hip.ptr(nkeep), hip.stream_ptr()), 'vps_nms_batched')
        return list(order[keep[:int(nkeep.item())].cpu().numpy().astype(np.int64)])
    return _nms
