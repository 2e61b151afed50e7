"""GPU: the three-way pin of the reference's native operators at the BASELINE shapes (SURVEY.md §2.3 launch shapes at
1024x2048):

    oracle/_ref/libvpsref.so  — the REFERENCE's own kernel bodies (correlation / resample2d / channelnorm / RoIAlign /
                                deformable im2col / both NMS kernels) compiled for gfx950 by oracle/build_ref.py
 == oracle/ops.py             — the CPU restatement the rest of the parity suite is built on
 == libvpship (vps_amd)       — the product kernels, through the C-ABI

`_ref` is test infrastructure: it is loaded here and nowhere else. The .so is built where /root/reference is mounted and
travels with the snapshot; a missing library fails the tests loudly on the GPU box.
Tolerances are stated per test; index outputs (NMS) must be identical.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fusetrack as OF
from oracle import ops as O
from oracle import ref_native as RN
from vps_amd import hip, nhwc
from vps_amd import operators as P

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _err(got, ref):
    got = got.detach().cpu().double(); ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float((got - ref).abs().max()), float(ref.abs().max())


def _close(got, ref, rel, what):
    e, m = _err(got, ref)
    print('%s: max abs err %.3e (max |ref| %.3e)' % (what, e, m))
    assert e <= rel * max(m, 1e-30), '%s: max abs err %.3e > %.1e * %.3e' % (what, e, rel, m)


def _rois(n, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    cx = torch.rand(n, generator=g) * W; cy = torch.rand(n, generator=g) * H
    s = torch.exp(torch.rand(n, generator=g) * math.log(512 / 16)) * 16
    ar = torch.exp((torch.rand(n, generator=g) - 0.5))
    w = s * ar; h = s / ar
    b = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    b[:, 0::2] = b[:, 0::2].clamp(0, W - 1); b[:, 1::2] = b[:, 1::2].clamp(0, H - 1)
    return torch.cat([torch.zeros(n, 1), b], 1)


def test_ref_library_is_present():
    assert RN.available(), 'oracle/_ref/libvpsref.so is missing (python oracle/build_ref.py)'
    RN.load()


# FlowNetC: Correlation(pad 20, k 1, maxdisp 20, s1 1, s2 2) on 2x[1,256,128,256] -> 441 ch (FlowNetC.py:31);
# LiteFlowNetCorr: Correlation(pad 4, k 1, maxdisp 4, s1 1, s2 1) on 2x[1,256,256,512] -> 81 ch (flow_modules.py:54-56)
@pytest.mark.parametrize('C,H,W,md,s2', [(256, 128, 256, 20, 2), (256, 256, 512, 4, 1), (64, 9, 11, 4, 1)],
                         ids=['flownetc_441ch_128x256', 'liteflownet_81ch_256x512', 'small_ragged'])
def test_correlation_three_way(dev, C, H, W, md, s2):
    a = _rand(1, C, H, W, seed=1); b = _rand(1, C, H, W, seed=2)
    ref = RN.correlation(a.to(dev), b.to(dev), md, 1, md, 1, s2)
    ora = O.correlation(a, b, md, 1, md, 1, s2)
    ours = P.Correlation(pad_size=md, kernel_size=1, max_displacement=md, stride1=1, stride2=s2)(a.to(dev), b.to(dev))
    # 256-term fp32 dot products in three different summation orders: 32-lane tree (_ref), torch sum (oracle), butterfly (ours)
    _close(ora, ref, 2e-6, 'correlation oracle vs reference kernel')
    _close(ours, ref, 2e-6, 'correlation libvpship vs reference kernel')


@pytest.mark.parametrize('C,H,W,mag', [(3, 1024, 2048, 6.0), (3, 17, 23, 40.0)], ids=['flownet2_1024x2048', 'ragged_large_flow'])
def test_resample2d_three_way(dev, C, H, W, mag):
    img = _rand(1, C, H, W, seed=1); flow = _rand(1, 2, H, W, seed=2, scale=mag)
    ref = RN.resample2d(img.to(dev), flow.to(dev))
    ora = O.resample2d(img, flow)
    ours = P.Resample2d()(img.to(dev), flow.to(dev))
    _close(ora, ref, 1e-6, 'resample2d oracle vs reference kernel')
    _close(ours, ref, 1e-6, 'resample2d libvpship vs reference kernel')


@pytest.mark.parametrize('C', [3, 2])
def test_channelnorm_three_way(dev, C):
    x = _rand(1, C, 1024, 2048, seed=3, scale=4.0)
    ref = RN.channelnorm(x.to(dev))
    _close(O.channelnorm(x), ref, 2e-7, 'channelnorm oracle vs reference kernel')
    _close(P.ChannelNorm()(x.to(dev)), ref, 2e-7, 'channelnorm libvpship vs reference kernel')


# bbox_roi_extractor: <=1000 proposals x 7x7 over P2..P5 (panoptic_fusetrack.py:367-369); mask_roi_extractor: K<=100 x 14x14
@pytest.mark.parametrize('R,Pz', [(1000, 7), (100, 14)], ids=['bbox_1000x7x7', 'mask_100x14x14'])
def test_roi_align_three_way_four_levels(dev, R, Pz):
    H, W = 1024, 2048
    strides = (4, 8, 16, 32)
    feats = [_rand(1, 256, H // s, W // s, seed=s) for s in strides]
    rois = _rois(R, H, W, seed=5)
    # SingleRoIExtractor (single_level.py:54-107) around the reference kernel: one launch per level with >= 1 roi
    lvls = OF.map_roi_levels(rois, 4)
    ref = torch.zeros(R, 256, Pz, Pz)
    fd = [f.to(dev) for f in feats]
    for i, s in enumerate(strides):
        inds = torch.nonzero(lvls == i).flatten()
        if inds.numel():
            ref[inds] = RN.roi_align(fd[i], rois[inds].to(dev), Pz, 1.0 / s, 2).cpu()
    ours = nhwc.roi_align([nhwc.from_nchw(f) for f in fd], list(strides), rois.to(dev), Pz).permute(0, 3, 1, 2)
    _close(ours, ref, 2e-6, 'roi_align libvpship vs reference kernel (%d rois, %dx%d, 4 levels)' % (R, Pz, Pz))
    sub = torch.arange(0, R, max(R // 40, 1))           # the CPU restatement loops over rois: a strided subset
    ora = OF.roi_extract(feats, rois[sub], Pz)
    # the compiled reference kernel contracts `roi_start + ph * bin_size + ...` into FMAs (nvcc and hipcc both do by default):
    # sampling coordinates differ from the unfused CPU restatement in the last bit, values by ~1e-5 of the feature range.
    # libvpship reproduces the compiled kernel to 1 ulp (above); the oracle is within 2e-5.
    _close(ora, ref[sub], 2e-5, 'roi_align oracle vs reference kernel (subset of %d rois)' % sub.numel())


# UPSNetFPN tower (upsnetFPN.py:39-52): DeformConv 3x3 pad 1 on P2 (256x512) / P3; offsets from the 18-channel conv
@pytest.mark.parametrize('cin,cout,H,W', [(256, 256, 256, 512), (256, 128, 128, 256), (128, 128, 33, 47)],
                         ids=['p2_256to256', 'p3_256to128', 'ragged_128to128'])
def test_deform_conv_three_way(dev, cin, cout, H, W):
    x = _rand(1, cin, H, W, seed=1)
    off = _rand(1, 18, H, W, seed=2, scale=1.5)
    w = _rand(cout, cin, 3, 3, seed=3, scale=(2.0 / (cin * 9)) ** 0.5)
    xd, od = x.to(dev), off.to(dev)
    # (1) the sampled columns: reference im2col kernel vs the oracle's sampling (a strided set of output positions)
    col = RN.deformable_im2col(xd, od)                               # [cin*9, H*W]
    ref = (w.to(dev).reshape(cout, -1).double() @ col.double()).float().view(1, cout, H, W)   # deform_conv_cuda.cpp:233-236 (addmm_)
    ora = O.deform_conv(x, off, w, 1, 1)
    _close(ora, ref, 2e-5, 'deform_conv oracle vs reference im2col + GEMM')
    # all three fp32-grade modes, the benchmarked f16x3 included (its 22-bit operands: 3 * 2^-22 per product, same bound as the others)
    for prec, tol, name in ((hip.PREC_F32, 2e-5, 'f32'), (hip.PREC_BF16X6, 2e-5, 'bf16x6'), (hip.PREC_F16X3, 2e-5, 'f16x3')):
        pc = nhwc.PackedConv(w, None, None, 1, 1, device=dev, deform=True, prec=prec)
        out = pc(nhwc.from_nchw(xd), ws=nhwc.Workspace(dev), name='o', offset=nhwc.from_nchw(od)).to_nchw()
        _close(out, ref, tol, 'deform_conv libvpship (%s) vs reference im2col + GEMM' % name)


def _dets(n, seed, H=1024, W=2048, ties=False):
    r = _rois(n, H, W, seed)[:, 1:]
    sc = torch.rand(n, generator=torch.Generator().manual_seed(seed + 100))
    if ties:
        sc = (sc * 50).round() / 50          # many exactly equal scores: the sort order of ties is part of the contract
    return torch.cat([r, sc[:, None]], 1)


# RPN: <=1000 boxes per level, thr 0.7 (rpn_head.py:92); indices returned ascending (nms_kernel.cu:127-130)
@pytest.mark.parametrize('n,ties', [(1000, False), (1000, True), (65, False), (1, False)])
def test_mmdet_nms_three_way(dev, n, ties):
    d = _dets(n, 7, ties=ties)
    _, ref = RN.nms_mmdet(d.to(dev), 0.7)
    _, ora = O.nms_mmdet(d, 0.7)
    _, ours = P.nms(d.to(dev), 0.7)
    assert torch.equal(ora, ref.cpu()), 'oracle nms != reference kernel'
    assert torch.equal(ours.cpu(), ref.cpu()), 'libvpship nms != reference kernel'


# MaskROI: all (roi, class) candidates above 0.6 as one class-agnostic set, thr 0.5 (mask_roi.py:88-103): up to 8000 boxes
@pytest.mark.parametrize('n,ties', [(3000, False), (8000, True), (64, False)])
def test_upsnet_nms_three_way(dev, n, ties):
    d = _dets(n, 11, ties=ties).numpy().astype(np.float32)
    ref = [int(i) for i in RN.nms_upsnet(d, 0.5)]
    ora = [int(i) for i in O.nms_upsnet(d, 0.5)]
    ours = [int(i) for i in P.gpu_nms_wrapper(0.5, 0)(d)]
    assert ora == ref, 'oracle gpu_nms != reference _nms'
    assert ours == ref, 'libvpship gpu_nms != reference _nms'
