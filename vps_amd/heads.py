"""Heads of the FuseTrack detector on libvpship. Registry names and parameter names mirror the reference:
UPSNetFPN (models/panoptic/upsnetFPN.py), RPNHead (anchor_heads/rpn_head.py + anchor_head.py), SingleRoIExtractor
(roi_extractors/single_level.py), SharedFCBBoxHead (bbox_heads/convfc_bbox_head.py), TrackHead
(track_heads/track_head.py), FCNMaskHead (mask_heads/fcn_mask_head.py).
"""
import numpy as np
import torch
import torch.nn as nn

from . import hip, nhwc
from .base import HipModule
from .necks import _ConvModule
from .registry import HEADS, PANOPTIC, ROI_EXTRACTORS


# ------------------------------------------------------------------------------------------------------------
class _DeformConvParams(nn.Module):
    """mmdet.ops.DeformConv parameter container: `.weight` [O, I, 3, 3], no bias (ops/dcn/deform_conv.py)"""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k).normal_(0, 0.01))
        self.stride, self.padding = (1, 1), (1, 1)
        self.bias = None


class _DeformConvWithOffset(nn.Module):
    # utils/deform_conv_with_offset.py:8-37
    def __init__(self, cin, cout):
        super().__init__()
        self.conv_offset = nn.Conv2d(cin, 18, 3, padding=1)
        self.conv = _DeformConvParams(cin, cout, 3)


@PANOPTIC.register_module
class UPSNetFPN(HipModule):
    def __init__(self, in_channels, out_channels, num_levels, num_things_classes, num_classes, ignore_label=255,
                 loss_weight=1.0, conv_cfg=None, norm_cfg=None):
        super().__init__()
        self.in_channels, self.out_channels, self.num_levels = in_channels, out_channels, num_levels
        self.num_things_classes, self.num_classes = num_things_classes, num_classes
        self.num_stuff_classes = num_classes - num_things_classes
        self.deform_convs = nn.ModuleList([nn.Sequential(
            _DeformConvWithOffset(in_channels, in_channels), nn.GroupNorm(32, in_channels), nn.ReLU(),
            _DeformConvWithOffset(in_channels, out_channels), nn.GroupNorm(32, out_channels), nn.ReLU(),
            _DeformConvWithOffset(out_channels, out_channels), nn.GroupNorm(32, out_channels), nn.ReLU())])
        self.conv_pred = _ConvModule(out_channels * 4, num_classes, 1)

    def pack(self, device):
        seq = self.deform_convs[0]
        self._tower = []
        for ci, gi in ((0, 1), (3, 4), (6, 7)):
            d, g = seq[ci], seq[gi]
            self._tower.append(dict(
                off=nhwc.pack_conv_module(d.conv_offset, device=device),
                dcn=nhwc.PackedConv(d.conv.weight, None, None, 1, 1, device=device, deform=True),
                gamma=g.weight.detach().float().to(device), beta=g.bias.detach().float().to(device), eps=g.eps, G=g.num_groups))
        self._pred = nhwc.pack_conv_module(self.conv_pred.conv, device=device)

    def run(self, levels, ws, tag='sem.'):
        """upsnetFPN.py:67-81 -> fcn_score FMap [1,H/4,W/4,19(+1)]. The full-resolution fcn_output (x4 bilinear, 159 MB)
        is never stored: the panoptic combine kernel re-derives it per pixel."""
        self.ensure_packed(levels[0].t.device)
        assert len(levels) == self.num_levels
        l0 = levels[0]
        oc = self.out_channels
        # the tower maps live on this head's stream only: temporaries (the 4 x 128-channel concat included: every window is written
        # each frame); fcn_score is read by the combine kernel on the main stream: persistent
        cat = ws.fmap(tag + 'cat', l0.N, l0.H, l0.W, oc * len(levels), temp=True)
        # GroupNorm sums: one zeroed slot per (level, tower layer); the deformable conv's epilogue adds into it (unsplit launches:
        # the two high-resolution levels), else vps_groupnorm_relu makes its own statistics pass
        stats = ws.get(tag + 'gnstats', (len(levels) * len(self._tower), nhwc.GN_REP, 64), dtype=torch.float64)
        stats.zero_()
        for li, x in enumerate(levels):
            src = x
            for ti, t in enumerate(self._tower):
                n = '%sl%dt%d' % (tag, li, ti)
                off = t['off'](x, ws=ws, name=n + 'off')
                slot = stats[li * len(self._tower) + ti]
                raw = t['dcn'](x, ws=ws, name=n + 'dcn', offset=off, gn=(slot, t['G']), temp=True)
                if x is not src:
                    ws.release(x)
                last = ti == len(self._tower) - 1
                if last and li == 0:
                    dst = cat.window(0, oc)
                else:
                    dst = ws.fmap(n + 'gn', raw.N, raw.H, raw.W, raw.C, temp=True)
                x = nhwc.groupnorm_relu(raw, dst, t['G'], t['gamma'], t['beta'], t['eps'], slot, stats_ready=t['dcn'].gn_fused)
                ws.release(raw)
            if li > 0:
                nhwc.resize(x, cat.window(li * oc, oc), 'bilinear')
                ws.release(x)
        score = self._pred(cat, ws=ws, name=tag + 'fcn_score')
        ws.release(cat)
        return score

    def forward(self, inputs):
        """reference signature: (fcn_output [1,19,H,W], fcn_score [1,19,H/4,W/4]) NCHW."""
        ws = nhwc.Workspace(inputs[0].device)
        ws.pooling = False
        score = self.run([nhwc.from_nchw(t) for t in inputs], ws)
        up = nhwc.resize(score, ws.fmap('fcn_output', score.N, score.H * 4, score.W * 4, score.C), 'bilinear')
        return up.to_nchw(), score.to_nchw()


# ------------------------------------------------------------------------------------------------------------
def gen_base_anchors(base_size, scales, ratios):
    """core/anchor/anchor_generator.py:18-46 (scale_major, ctr None)"""
    scales = torch.tensor(scales, dtype=torch.float32); ratios = torch.tensor(ratios, dtype=torch.float32)
    w = h = float(base_size)
    xc = 0.5 * (w - 1); yc = 0.5 * (h - 1)
    hr = torch.sqrt(ratios); wr = 1 / hr
    ws_ = (w * wr[:, None] * scales[None, :]).view(-1)
    hs_ = (h * hr[:, None] * scales[None, :]).view(-1)
    return torch.stack([xc - 0.5 * (ws_ - 1), yc - 0.5 * (hs_ - 1), xc + 0.5 * (ws_ - 1), yc + 0.5 * (hs_ - 1)], dim=-1).round()


@HEADS.register_module
class RPNHead(HipModule):
    def __init__(self, in_channels, feat_channels=256, anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1.0, 2.0),
                 anchor_strides=(4, 8, 16, 32, 64), anchor_base_sizes=None, target_means=(.0, .0, .0, .0),
                 target_stds=(1.0, 1.0, 1.0, 1.0), loss_cls=None, loss_bbox=None):
        super().__init__()
        self.in_channels, self.feat_channels = in_channels, feat_channels
        self.anchor_scales, self.anchor_ratios = list(anchor_scales), list(anchor_ratios)
        self.anchor_strides = list(anchor_strides)
        self.anchor_base_sizes = list(anchor_strides) if anchor_base_sizes is None else list(anchor_base_sizes)
        self.target_means, self.target_stds = tuple(target_means), tuple(target_stds)
        assert loss_cls is None or loss_cls.get('use_sigmoid', False), 'RPN with sigmoid objectness is the configured path'
        self.num_anchors = len(self.anchor_ratios) * len(self.anchor_scales)
        self.rpn_conv = nn.Conv2d(in_channels, feat_channels, 3, padding=1)
        self.rpn_cls = nn.Conv2d(feat_channels, self.num_anchors, 1)
        self.rpn_reg = nn.Conv2d(feat_channels, self.num_anchors * 4, 1)
        self._anchor_cache = {}

    def pack(self, device):
        self._conv = nhwc.pack_conv_module(self.rpn_conv, act=hip.ACT_RELU, device=device)
        self._cls = nhwc.pack_conv_module(self.rpn_cls, device=device)
        self._reg = nhwc.pack_conv_module(self.rpn_reg, device=device)

    def _anchors(self, lvl, H, W, device):
        key = (lvl, H, W, str(device))
        a = self._anchor_cache.get(key)
        if a is None:   # core/anchor/anchor_generator.py:55-72, generated once per feature size
            base = gen_base_anchors(self.anchor_base_sizes[lvl], self.anchor_scales, self.anchor_ratios)
            s = self.anchor_strides[lvl]
            sx = torch.arange(0, W, dtype=torch.float32) * s
            sy = torch.arange(0, H, dtype=torch.float32) * s
            xx = sx.repeat(H); yy = sy.view(-1, 1).repeat(1, W).view(-1)
            shifts = torch.stack([xx, yy, xx, yy], dim=-1)
            a = (base[None, :, :] + shifts[:, None, :]).view(-1, 4).to(device)
            self._anchor_cache[key] = a
        return a

    def run(self, levels, ws, img_shape, cfg, tag='rpn.'):
        """rpn_head.py:30-35 (forward_single per level) + anchor_head.py:198-223 + rpn_head.py:55-104.
        Returns (proposals [max_num, 5], n int32 [1]) on the device: rows >= n are zero (fewer than max_num boxes survived —
        tiny images only). NHWC makes the reference's permute(1,2,0) a no-op."""
        dev = levels[0].t.device
        self.ensure_packed(dev)
        lib = hip.load()
        A = self.num_anchors
        nlv = len(levels)
        nms_pre = cfg.nms_pre
        # vps_rpn_collect walks every level's keep list in score order and cuts it at nms_post: equal to the reference (which cuts the
        # nms() output of a level that was not top-k'ed in ascending anchor order) only when the cut never bites, and its LDS sort
        # holds 8192 candidates: the shipped configs (1000 / 1000 / 1000) satisfy both
        if not (cfg.nms_post >= nms_pre > 0 and nlv * min(cfg.nms_post, nms_pre) <= 8192):
            raise hip.VpsHipError('RPNHead: test_cfg.rpn needs nms_post >= nms_pre > 0 and levels * nms_pre <= 8192 (got nms_pre=%d nms_post=%d, %d levels)'
                                  % (nms_pre, cfg.nms_post, nlv))
        boxes = ws.get(tag + 'boxes', (nlv, nms_pre, 5), zero=False)
        cls_l, reg_l, counts = [], [], []
        for li, x in enumerate(levels):
            t = self._conv(x, ws=ws, name='%sconv%d' % (tag, li), temp=True)
            cls_l.append(self._cls(t, ws=ws, name='%scls%d' % (tag, li)))
            reg_l.append(self._reg(t, ws=ws, name='%sreg%d' % (tag, li)))
            ws.release(t)
            counts.append(min(x.H * x.W * A, nms_pre))
        # rpn_head.py:62-91 for all levels in ONE launch (sigmoid, top nms_pre by score, gathers, delta2bbox): vps_rpn_select.
        # Base anchors as the reference rounds them (anchor_generator.py:28-53), once per device
        key = str(dev)
        if self._anchor_cache.get('base') is None or self._anchor_cache['base'][0] != key:
            base = torch.stack([gen_base_anchors(self.anchor_base_sizes[li], self.anchor_scales, self.anchor_ratios) for li in range(nlv)])
            self._anchor_cache['base'] = (key, base.to(dev).contiguous())
        base_d = self._anchor_cache['base'][1]
        assert base_d.shape == (nlv, A, 4)
        from ctypes import c_float, c_int32, c_void_p
        cp = (c_void_p * nlv)(*[c.t.data_ptr() for c in cls_l]); cl = (c_int32 * nlv)(*[c.ld for c in cls_l])
        rp = (c_void_p * nlv)(*[r.t.data_ptr() for r in reg_l]); rl = (c_int32 * nlv)(*[r.ld for r in reg_l])
        for c, r in zip(cls_l, reg_l):
            assert c.coff == 0 and r.coff == 0
        Hs = (c_int32 * nlv)(*[x.H for x in levels]); Ws = (c_int32 * nlv)(*[x.W for x in levels])
        st = (c_float * nlv)(*[float(v) for v in self.anchor_strides[:nlv]])
        stds = (c_float * 4)(*[float(v) for v in self.target_stds])
        assert all(float(v) == 0.0 for v in self.target_means)
        nkeys = sum(x.H * x.W * A for x in levels)
        keys = ws.get(tag + 'keys', (nkeys,), dtype=torch.int32, zero=False)
        hist = ws.get(tag + 'hist', (nlv * 4096,), dtype=torch.int32, zero=True)        # zero once: every launch leaves it zero
        hip.check(lib.vps_rpn_select(cp, cl, rp, rl, Hs, Ws, st, nlv, A, hip.ptr(base_d), nms_pre, stds, float(img_shape[0]), float(img_shape[1]),
                                     hip.ptr(keys), hip.ptr(hist), hip.ptr(boxes), hip.stream_ptr()), 'vps_rpn_select')
        assert cfg.min_bbox_size == 0 and not cfg.nms_across_levels
        cb = (nms_pre + 63) // 64
        ck = (tuple(counts), str(dev))
        if getattr(self, '_counts_key', None) != ck:          # uploaded once per frame size
            self._counts_d, self._counts_key = torch.tensor(counts, dtype=torch.int32, device=dev), ck
        counts_d = self._counts_d
        mask = ws.get(tag + 'nmsmask', (nlv * nms_pre * cb,), dtype=torch.int64, zero=False)
        keep = ws.get(tag + 'keep', (nlv, nms_pre), dtype=torch.int32, zero=False)
        nkeep = ws.get(tag + 'nkeep', (nlv,), dtype=torch.int32)
        hip.check(lib.vps_nms_batched(hip.ptr(boxes), nlv, nms_pre, hip.ptr(counts_d), float(cfg.nms_thr), hip.ptr(mask),
                                      hip.ptr(keep), hip.ptr(nkeep), hip.stream_ptr()), 'vps_nms_batched')
        # rpn_head.py:94-104 on the device (no host sync): kept boxes of every level, `[:nms_post]` each, the `max_num` best by
        # score. The per-level order the reference returns them in (ascending anchor index for the levels it hands to nms()
        # unsorted) is irrelevant behind the final top-k by score.
        props = ws.get(tag + 'props', (cfg.max_num, 5), zero=False)
        nprops = ws.get(tag + 'nprops', (1,), dtype=torch.int32, zero=False)
        hip.check(lib.vps_rpn_collect(hip.ptr(boxes), hip.ptr(keep), hip.ptr(nkeep), nlv, nms_pre, int(cfg.nms_post), int(cfg.max_num),
                                      hip.ptr(props), hip.ptr(nprops), hip.stream_ptr()), 'vps_rpn_collect')
        return props, nprops


# ------------------------------------------------------------------------------------------------------------
@ROI_EXTRACTORS.register_module
class SingleRoIExtractor(HipModule):
    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56):
        super().__init__()
        cfg = dict(roi_layer)
        assert cfg.pop('type') == 'RoIAlign'
        self.out_size = cfg.get('out_size', 7)
        self.sample_num = cfg.get('sample_num', 2)
        self.out_channels, self.featmap_strides, self.finest_scale = out_channels, list(featmap_strides), finest_scale

    num_inputs = property(lambda s: len(s.featmap_strides))

    def pack(self, device):
        pass

    def run(self, levels, rois):
        """single_level.py:89-107: ONE launch for all levels -> [R, P, P, C] NHWC device tensor."""
        n = len(self.featmap_strides)
        return nhwc.roi_align(levels[:n], self.featmap_strides, rois, self.out_size, self.sample_num, float(self.finest_scale))

    def forward(self, feats, rois, roi_scale_factor=None):
        assert roi_scale_factor is None
        out = self.run([nhwc.from_nchw(f) for f in feats], rois)
        return out.permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------------------------------------------
@HEADS.register_module
class SharedFCBBoxHead(HipModule):
    def __init__(self, num_fcs=2, fc_out_channels=1024, in_channels=256, roi_feat_size=7, num_classes=81,
                 target_means=(0., 0., 0., 0.), target_stds=(0.1, 0.1, 0.2, 0.2), reg_class_agnostic=False, **unused):
        super().__init__()
        assert num_fcs == 2 and not reg_class_agnostic
        self.in_channels, self.roi_feat_size, self.num_classes = in_channels, roi_feat_size, num_classes
        d = in_channels * roi_feat_size * roi_feat_size
        self.shared_fcs = nn.ModuleList([nn.Linear(d, fc_out_channels), nn.Linear(fc_out_channels, fc_out_channels)])
        self.fc_cls = nn.Linear(fc_out_channels, num_classes)
        self.fc_reg = nn.Linear(fc_out_channels, 4 * num_classes)

    def pack(self, device):
        S = self.roi_feat_size ** 2
        self._fc1 = nhwc.pack_linear(self.shared_fcs[0].weight, self.shared_fcs[0].bias, hip.ACT_RELU, device, chw=(self.in_channels, S))
        self._fc2 = nhwc.pack_linear(self.shared_fcs[1].weight, self.shared_fcs[1].bias, hip.ACT_RELU, device)
        # fc_cls and fc_reg share their input: one GEMM with the two weight blocks stacked
        w = torch.cat([self.fc_cls.weight, self.fc_reg.weight], 0); b = torch.cat([self.fc_cls.bias, self.fc_reg.bias], 0)
        self._out = nhwc.pack_linear(w, b, hip.ACT_NONE, device)

    def run(self, roi_feats, ws, tag='bbox.'):
        """convfc_bbox_head.py:132-168 on NHWC roi features [R,7,7,C] -> (cls_score [R,nc], bbox_pred [R,4nc])"""
        self.ensure_packed(roi_feats.device)
        R = roi_feats.shape[0]
        x = nhwc.FMap(roi_feats.view(1, 1, R, -1))
        x = self._fc1(x, ws=ws, name=tag + 'fc1')
        x = self._fc2(x, ws=ws, name=tag + 'fc2')
        o = self._out(x, ws=ws, name=tag + 'out').t.view(R, -1)
        nc = self.num_classes
        return o[:, :nc].contiguous(), o[:, nc:nc * 5].contiguous()

    def forward(self, x):
        return self.run(x.permute(0, 2, 3, 1).contiguous(), nhwc.Workspace(x.device))


@HEADS.register_module
class TrackHead(HipModule):
    def __init__(self, with_avg_pool=False, num_fcs=2, in_channels=256, roi_feat_size=7, fc_out_channels=1024,
                 match_coeff=None, bbox_dummy_iou=0, dynamic=True, loss_match=None):
        super().__init__()
        assert not with_avg_pool and num_fcs == 2 and dynamic
        self.in_channels, self.roi_feat_size = in_channels, roi_feat_size
        self.match_coeff, self.bbox_dummy_iou = match_coeff, bbox_dummy_iou
        d = in_channels * roi_feat_size * roi_feat_size
        self.fcs = nn.ModuleList([nn.Linear(d, fc_out_channels), nn.Linear(fc_out_channels, fc_out_channels)])

    def pack(self, device):
        S = self.roi_feat_size ** 2
        self._fc1 = nhwc.pack_linear(self.fcs[0].weight, self.fcs[0].bias, hip.ACT_RELU, device, chw=(self.in_channels, S))
        self._fc2 = nhwc.pack_linear(self.fcs[1].weight, self.fcs[1].bias, hip.ACT_NONE, device)

    def embed(self, roi_feats, ws, tag='trk.'):
        """track_head.py:104-111: fc -> relu -> fc. The memory side applies the same fcs to stored features every frame
        in the reference; embeddings are cached instead (identical values)."""
        self.ensure_packed(roi_feats.device)
        R = roi_feats.shape[0]
        x = nhwc.FMap(roi_feats.reshape(1, 1, R, -1))
        x = self._fc1(x, ws=ws, name=tag + 'fc1')
        return self._fc2(x, ws=ws, name=tag + 'fc2').t.view(R, -1).clone()

    def match_scores(self, emb, mem_emb, ws, tag='trk.'):
        """track_head.py:112-130: x @ ref_x^T with a prepended zero column -> [K, M+1] (the product is the conv GEMM with
        the memory embeddings as the weight panel)."""
        K, D = emb.shape
        M = mem_emb.shape[0]
        # exact-fp32 MFMA kernel: a [K x M x 1024] product is ~0.1 GFLOP, and the split modes would re-split the memory matrix
        # into operand planes every frame (~30 small elementwise launches)
        pc = nhwc.PackedConv.from_matrix(mem_emb, prec=hip.PREC_F32)
        prod = pc(nhwc.FMap(emb.contiguous().view(1, 1, K, D)), ws=ws, name=tag + 'prod').t.view(K, -1)[:, :M]
        return torch.cat([prod.new_zeros(K, 1), prod], dim=1)

    def compute_comp_scores(self, match_ll, bbox_scores, bbox_ious, label_delta, add_bbox_dummy=False):
        """track_head.py:73-91"""
        if add_bbox_dummy:
            bbox_ious = torch.cat((torch.ones(bbox_ious.size(0), 1, device=bbox_ious.device) * self.bbox_dummy_iou, bbox_ious), dim=1)
            label_delta = torch.cat((torch.ones(bbox_ious.size(0), 1, device=bbox_ious.device), label_delta), dim=1)
        if self.match_coeff is None:
            return match_ll
        assert len(self.match_coeff) == 3
        return (match_ll + self.match_coeff[0] * torch.log(bbox_scores) + self.match_coeff[1] * bbox_ious
                + self.match_coeff[2] * label_delta)


@HEADS.register_module
class FCNMaskHead(HipModule):
    def __init__(self, num_convs=4, roi_feat_size=14, in_channels=256, conv_kernel_size=3, conv_out_channels=256,
                 upsample_method='deconv', upsample_ratio=2, num_classes=81, class_agnostic=False, **unused):
        super().__init__()
        assert upsample_method == 'deconv' and upsample_ratio == 2 and not class_agnostic and conv_kernel_size == 3
        self.num_classes = num_classes
        self.convs = nn.ModuleList([_ConvModule(in_channels if i == 0 else conv_out_channels, conv_out_channels, 3, 1)
                                    for i in range(num_convs)])
        self.upsample = nn.ConvTranspose2d(conv_out_channels, conv_out_channels, 2, stride=2)
        self.conv_logits = nn.Conv2d(conv_out_channels, num_classes, 1)

    def pack(self, device):
        self._convs = [nhwc.pack_conv_module(m.conv, act=hip.ACT_RELU, device=device) for m in self.convs]
        self._up = nhwc.pack_conv_module(self.upsample, act=hip.ACT_RELU, device=device)
        self._logits = nhwc.pack_conv_module(self.conv_logits, device=device)

    def run(self, mask_feats, ws, tag='mask.'):
        """fcn_mask_head.py:95-103 on NHWC roi features [K,14,14,C] -> NHWC logits FMap [K,28,28,num_classes(+pad)]"""
        self.ensure_packed(mask_feats.device)
        x = nhwc.FMap(mask_feats)
        for i, pc in enumerate(self._convs):
            x = pc(x, ws=ws, name='%sc%d' % (tag, i))
        x = self._up(x, ws=ws, name=tag + 'up')
        return self._logits(x, ws=ws, name=tag + 'logits')

    def forward(self, x):
        return self.run(x.permute(0, 2, 3, 1).contiguous(), nhwc.Workspace(x.device)).to_nchw()
