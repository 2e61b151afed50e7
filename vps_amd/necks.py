"""FPN neck and the BFP-TCEA temporal fusion neck on libvpship (registry names `FPN`, `BFPTcea`).

Mirrors mmdet/models/necks/fpn.py:10-139, extra_necks/bfp_tcea.py:13-149, utils/tcea_modules.py:17-78 and
flow_modules/flow_modules.py:37-148 (OpticalFlowEstimatorCorr / LiteFlowNetCorr / WarpingLayer) at the
parameter-name level.
"""
import os

import torch
import torch.nn as nn

from . import hip, nhwc
from .base import HipModule
from .registry import EXTRA_NECKS, NECKS

# VPS_SCATTER_ALL=0: BFP scatter level by level (five passes over the refined map) instead of vps_bfp_scatter_all (A/B; bitwise equal)
SCATTER_ALL = os.environ.get('VPS_SCATTER_ALL', '1') != '0'


class _ConvModule(nn.Module):
    """parameter container named like mmdet's ConvModule (`.conv`)"""

    def __init__(self, cin, cout, k, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding)


@NECKS.register_module
class FPN(HipModule):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 **unused):
        super().__init__()
        assert start_level == 0 and end_level == -1 and not add_extra_convs, 'only the fusetrack.py FPN is on the path'
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        self.lateral_convs = nn.ModuleList([_ConvModule(c, out_channels, 1) for c in in_channels])
        self.fpn_convs = nn.ModuleList([_ConvModule(out_channels, out_channels, 3, 1) for _ in in_channels])

    def pack(self, device):
        self._lat = [nhwc.pack_conv_module(m.conv, device=device) for m in self.lateral_convs]
        self._out = [nhwc.pack_conv_module(m.conv, device=device) for m in self.fpn_convs]

    def run(self, feats, ws, tag):
        """feats C2..C5 -> P2..P6. The top-down `laterals[i-1] += nearest_x2(laterals[i])` (fpn.py:108-111) is the
        residual input of the lateral 1x1 conv (res_shift=1): no separate upsample/add pass."""
        self.ensure_packed(feats[0].t.device)
        n = len(feats)
        lats = [None] * n
        for i in range(n - 1, -1, -1):
            lats[i] = self._lat[i](feats[i], ws=ws, name='%slat%d' % (tag, i),
                                   res=lats[i + 1] if i + 1 < n else None, res_shift=1 if i + 1 < n else 0, temp=True)
        # the levels are what the rest of the frame (another stream, the next frame) reads: kept (`out` target of the workspace)
        outs = [self._out[i](lats[i], ws=ws, name='%sp%d' % (tag, i + 2), keep=True) for i in range(n)]
        ws.release(*lats)
        for j in range(self.num_outs - n):   # F.max_pool2d(k=1, stride=2) == stride-2 subsample (fpn.py:124-126)
            src = outs[-1]
            dst = ws.fmap('%sp%d' % (tag, n + 2 + j), src.N, (src.H + 1) // 2, (src.W + 1) // 2, src.C, out=True)
            outs.append(nhwc.resize(src, dst, 'nearest'))
        return outs

    def forward(self, inputs):
        ws = nhwc.Workspace(inputs[0].device)
        ws.pooling = False
        return tuple(o.to_nchw() for o in self.run([nhwc.from_nchw(t) for t in inputs], ws, 'fpn.'))


# ------------------------------------------------------------------------------------------------------------
class _FlowEstimator(nn.Module):
    # flow_modules.py:37-48: convs = Sequential(conv(ch,64), conv(64,64), conv(64,32), Conv2d(32,2))
    def __init__(self, ch_in):
        super().__init__()
        mk = lambda a, b: nn.Sequential(nn.Conv2d(a, b, 3, padding=1), nn.LeakyReLU(0.1))
        self.convs = nn.Sequential(mk(ch_in, 64), mk(64, 64), mk(64, 32), nn.Conv2d(32, 2, 3, padding=1))


class _LiteFlowNetCorr(nn.Module):
    def __init__(self, in_ch, search_range=4):
        super().__init__()
        self.search_range = search_range
        self.flow_estimator = _FlowEstimator(in_ch + (2 * search_range + 1) ** 2)


class _TCEAFusion(nn.Module):
    # utils/tcea_modules.py:22-39
    def __init__(self, nf, nframes, center):
        super().__init__()
        self.center = center
        self.tAtt_1 = nn.Conv2d(nf, nf, 3, 1, 1)
        self.tAtt_2 = nn.Conv2d(nf, nf, 3, 1, 1)
        self.fea_fusion = nn.Conv2d(nframes * nf, nf, 1, 1)
        self.sAtt_1 = nn.Conv2d(nframes * nf, nf, 1, 1)
        self.sAtt_2 = nn.Conv2d(nf * 2, nf, 1, 1)
        self.sAtt_3 = nn.Conv2d(nf, nf, 3, 1, 1)
        self.sAtt_4 = nn.Conv2d(nf, nf, 3, 1, 1)
        self.sAtt_add_1 = nn.Conv2d(nf, nf, 1, 1)
        self.sAtt_add_2 = nn.Conv2d(nf, nf, 1, 1)


@EXTRA_NECKS.register_module
class BFPTcea(HipModule):
    def __init__(self, in_channels, num_levels, refine_level=1, refine_type=None, nframes=3, center=None,
                 stack_type='add', conv_cfg=None, norm_cfg=None):
        super().__init__()
        assert refine_level == 0 and refine_type == 'conv' and nframes == 2 and center == 0, \
            'only the fusetrack.py BFPTcea configuration is on the path'
        self.in_channels, self.num_levels = in_channels, num_levels
        self.liteflownet = _LiteFlowNetCorr(in_channels + 2, 4)
        self.tcea_fusion = _TCEAFusion(in_channels, nframes, center)
        self.refine = _ConvModule(in_channels, in_channels, 3, 1)

    def pack(self, device):
        P = nhwc.pack_conv_module
        L, R = hip.ACT_LEAKY, hip.ACT_RELU
        fe = self.liteflownet.flow_estimator.convs
        self._est = [P(fe[0][0], act=L, device=device), P(fe[1][0], act=L, device=device),
                     P(fe[2][0], act=L, device=device), P(fe[3], device=device)]
        t = self.tcea_fusion
        self._t = dict(tAtt_1=P(t.tAtt_1, device=device), tAtt_2=P(t.tAtt_2, device=device),
                       # fea_fusion and sAtt_1 are both 1x1 convolutions + LeakyReLU of the SAME 512-channel input: ONE launch with the two
                       # weight blocks stacked (the 268 MB input is read once; per output channel the arithmetic is unchanged)
                       fea_att1=nhwc.PackedConv(torch.cat([t.fea_fusion.weight, t.sAtt_1.weight], 0), torch.cat([t.fea_fusion.bias, t.sAtt_1.bias], 0),
                                                None, 1, 0, act=L, device=device),
                       sAtt_2=P(t.sAtt_2, act=L, device=device), sAtt_3=P(t.sAtt_3, act=L, device=device),
                       sAtt_4=P(t.sAtt_4, device=device), sAtt_add_1=P(t.sAtt_add_1, act=L, device=device),
                       sAtt_add_2=P(t.sAtt_add_2, device=device))
        self._refine = P(self.refine.conv, act=R, device=device)

    # --- pieces ------------------------------------------------------------------------------------------
    def gather(self, levels, ws, name):
        """bfp_tcea.py:96-109 -> the 339(+1)-channel LiteFlowNet input buffer [bsf | corr | flow_init]; the bsf window is
        also what the NEXT frame needs as ref_bsf (bfp_tcea.py:117), so the detector keeps the buffer alive."""
        C = self.in_channels
        l0 = levels[0]
        cat = ws.fmap(name, l0.N, l0.H, l0.W, C + 81 + 2, out=True)
        nhwc.bfp_gather(levels, cat.window(0, C))
        return cat

    def run(self, levels, cat, ref_bsf, ws, tag):
        """levels: P2..P6 of the target frame; cat: gather(levels) buffer whose window [C+81, C+83) already holds
        flow_init (written by the detector's x0.25 flow resize); ref_bsf: gathered reference-frame feature."""
        self.ensure_packed(levels[0].t.device)
        C = self.in_channels
        N, H, W = cat.N, cat.H, cat.W
        bsf = cat.window(0, C)
        flow_init = cat.window(C + 81, 2)
        # everything between the gathered feature and the five outputs lives on the current stream only: temporaries, released after
        # their last consumer (the aux dict hands flow_fine / warp / fused / refined out: valid until the next call on this workspace
        # when pooling is on - tools that inspect them switch `ws.pooling` off)
        T = lambda nm, c: ws.fmap(tag + nm, N, H, W, c, temp=True)
        warp1 = nhwc.flow_warp(ref_bsf, flow_init, T('warp1', C))
        nhwc.correlation(bsf, warp1, cat.window(C, 81), 4, 1, prec=self._refine.prec)
        x = cat
        for i, pc in enumerate(self._est):
            y = pc(x, ws=ws, name='%sest%d' % (tag, i), temp=True)
            if x is not cat:
                ws.release(x)
            x = y
        flow_fine = x
        warp2 = nhwc.flow_warp(warp1, flow_fine, T('warp2', C))
        ws.release(warp1)
        # TCEA_Fusion (tcea_modules.py:50-78), frames = [bsf, warp2], center 0
        P = self._t
        emb = T('emb', 2 * C)
        P['tAtt_1'](bsf, out=emb.window(0, C), ws=ws)
        P['tAtt_1'](warp2, out=emb.window(C, C), ws=ws)
        emb_ref = P['tAtt_2'](bsf, ws=ws, name=tag + 'emb_ref', temp=True)
        al = nhwc.tcea_temporal(emb, emb_ref, bsf, warp2, T('aligned', 2 * C))
        ws.release(emb, emb_ref)
        fa = P['fea_att1'](al, ws=ws, name=tag + 'fea_att1', temp=True)
        fea, att = fa.window(0, C), fa.window(C, C)
        ws.release(al)
        Hh, Wh = (H + 1) // 2, (W + 1) // 2
        pooled = ws.fmap(tag + 'attpool', N, Hh, Wh, 2 * C, temp=True)
        nhwc.pool3x3s2(att, pooled.window(0, C), 'max')
        nhwc.pool3x3s2(att, pooled.window(C, C), 'avg')
        att = P['sAtt_2'](pooled, ws=ws, name=tag + 'att2', temp=True)
        ws.release(pooled)
        att3 = P['sAtt_3'](att, ws=ws, name=tag + 'att3', temp=True)
        ws.release(att)
        att_up = nhwc.resize(att3, ws.fmap(tag + 'attup', N, 2 * Hh, 2 * Wh, C, temp=True), 'bilinear')
        ws.release(att3)
        att = P['sAtt_4'](att_up, ws=ws, name=tag + 'att4', temp=True)
        ws.release(att_up)
        add1 = P['sAtt_add_1'](att, ws=ws, name=tag + 'add1', temp=True)
        add = P['sAtt_add_2'](add1, ws=ws, name=tag + 'add2', temp=True)
        ws.release(add1)
        fused = nhwc.tcea_modulate(fea, att, add, T('fused', C))
        ws.release(fa, att, add)
        refined = self._refine(fused, ws=ws, name=tag + 'refined', temp=True)
        outs = [ws.fmap('%sout%d' % (tag, i), lv.N, lv.H, lv.W, C) for i, lv in enumerate(levels)]
        if not (SCATTER_ALL and nhwc.bfp_scatter_all(refined, levels, outs)):                  # one pass over `refined` (round 6)
            for lv, o in zip(levels, outs):
                nhwc.bfp_scatter(refined, lv, o)
        ws.release(flow_fine, warp2, fused, refined)
        return outs, dict(flow_fine=flow_fine, warp=warp2, fused=fused, refined=refined)

    def forward(self, inputs, ref_inputs, flow_init, next_inputs=None, next_flow_init=None):
        """NCHW operator-level API with the reference signature (bfp_tcea.py:111)."""
        assert next_inputs is None
        dev = inputs[0].device
        ws = nhwc.Workspace(dev)
        ws.pooling = False
        lv = [nhwc.from_nchw(t) for t in inputs]
        cat = self.gather(lv, ws, 'cat')
        refcat = self.gather([nhwc.from_nchw(t) for t in ref_inputs], ws, 'refcat')
        fi = nhwc.from_nchw(flow_init)
        nhwc.resize(fi, cat.window(self.in_channels + 81, 2), 'nearest')
        outs, _ = self.run(lv, cat, refcat.window(0, self.in_channels), ws, 'neck.')
        return tuple(o.to_nchw() for o in outs)
