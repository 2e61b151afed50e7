"""FlowNet2 on libvpship. Parameter names mirror mmdet/models/flow_modules/{flownet2,FlowNetC,FlowNetS,FlowNetSD,
FlowNetFusion,submodules}.py so `FlowNet2_checkpoint.pth.tar` loads unchanged (`flownetc.conv1.0.weight`, ...).

MI355X-first structure: every decoder concat is a pre-allocated NHWC buffer that the encoder conv, the transposed
conv and the 2-channel flow up-conv write into directly (no torch.cat); the correlation writes its 441 channels
next to conv_redir; the full-resolution inter-network stages (x4 upsample + Resample2d + ChannelNorm + concat)
are ONE fused pass (vps_flow_stage) instead of 6-8 full-resolution kernels.
"""
import torch
import torch.nn as nn

from . import hip, nhwc
from .base import HipModule


def _conv(cin, cout, k=3, stride=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride=stride, padding=(k - 1) // 2, bias=True), nn.LeakyReLU(0.1))


def _iconv(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=True))


def _deconv(cin, cout):
    return nn.Sequential(nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=True), nn.LeakyReLU(0.1))


def _pred(cin):
    return nn.Conv2d(cin, 2, 3, padding=1, bias=True)


class _Net(nn.Module):
    """packs every Conv/ConvTranspose child once; LeakyReLU(0.1) iff the container is Sequential(conv, LeakyReLU)"""

    def pack(self, device):
        self._p = {}
        for name, m in self.named_children():
            if isinstance(m, nn.Sequential):
                act = hip.ACT_LEAKY if len(m) > 1 else hip.ACT_NONE
                self._p[name] = nhwc.pack_conv_module(m[0], act=act, slope=0.1, device=device)
            elif isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                self._p[name] = nhwc.pack_conv_module(m, device=device)


class _RefineS(_Net):
    """encoder tail + decoder shared by FlowNetC and FlowNetS (FlowNetC.py:104-123 == FlowNetS.py:69-90)"""

    def _mk_decoder(self, up_bias):
        self.deconv5 = _deconv(1024, 512); self.deconv4 = _deconv(1026, 256)
        self.deconv3 = _deconv(770, 128); self.deconv2 = _deconv(386, 64)
        self.predict_flow6 = _pred(1024); self.predict_flow5 = _pred(1026); self.predict_flow4 = _pred(770)
        self.predict_flow3 = _pred(386); self.predict_flow2 = _pred(194)
        for n in ('6_to_5', '5_to_4', '4_to_3', '3_to_2'):
            setattr(self, 'upsampled_flow' + n, nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=up_bias))

    def _cats(self, ws, tag, N, h2, w2):
        """the four decoder concat buffers [encoder | deconv | flow] (194 / 386 / 770 / 1026 channels, zero pad channels: persistent).
        FlowNetC, FlowNetS_1 and FlowNetS_2 run one after the other on ONE stream and rewrite every window: they share one set
        (`FlowNet2.run` passes them the same `cat_tag`; FlowNetSD may run beside them and has its own)"""
        ct = getattr(self, 'cat_tag', None) or tag
        return (ws.fmap(ct + 'cat2', N, h2, w2, 194), ws.fmap(ct + 'cat3', N, h2 // 2, w2 // 2, 386),
                ws.fmap(ct + 'cat4', N, h2 // 4, w2 // 4, 770), ws.fmap(ct + 'cat5', N, h2 // 8, w2 // 8, 1026))

    def _decode(self, ws, tag, cats, c6):
        """cats = [cat2, cat3, cat4, cat5] buffers whose first window already holds the encoder feature"""
        p = self._p
        cat2, cat3, cat4, cat5 = cats
        feat = c6
        for lvl, cat, enc_c, dec_c in ((5, cat5, 512, 512), (4, cat4, 512, 256), (3, cat3, 256, 128), (2, cat2, 128, 64)):
            flow = p['predict_flow%d' % (lvl + 1)](feat, ws=ws, name='%sflow%d' % (tag, lvl + 1))
            p['upsampled_flow%d_to_%d' % (lvl + 1, lvl)](flow, out=cat.window(enc_c + dec_c, 2), ws=ws)
            p['deconv%d' % lvl](feat, out=cat.window(enc_c, dec_c), ws=ws)
            feat = cat
        return p['predict_flow2'](cat2, ws=ws, name=tag + 'flow2')


class FlowNetC(_RefineS):
    def __init__(self):
        super().__init__()
        self.conv1 = _conv(3, 64, 7, 2); self.conv2 = _conv(64, 128, 5, 2); self.conv3 = _conv(128, 256, 5, 2)
        self.conv_redir = _conv(256, 32, 1, 1)
        self.conv3_1 = _conv(473, 256)
        self.conv4 = _conv(256, 512, stride=2); self.conv4_1 = _conv(512, 512)
        self.conv5 = _conv(512, 512, stride=2); self.conv5_1 = _conv(512, 512)
        self.conv6 = _conv(512, 1024, stride=2); self.conv6_1 = _conv(1024, 1024)
        self._mk_decoder(up_bias=True)

    def run(self, x6, ws, tag):
        """FlowNetC.py:71-128. x6: [1,H,W,8] (ch 0-2 img, 3-5 ref)."""
        p = self._p
        N, H, W = x6.N, x6.H, x6.W
        h2, w2 = H // 4, W // 4
        cat2, cat3, cat4, cat5 = self._cats(ws, tag, N, h2, w2)
        # the stem reads 3 channels out of the 6(+2)-channel pair buffer: pack once with cin padded to 4, second image via
        # a 4-aligned copy window
        xa = ws.fmap(tag + 'xa', N, H, W, 3)
        xb = ws.fmap(tag + 'xb', N, H, W, 3)
        hip.check(hip.load().vps_axpb(x6.ptr(), x6.ld, 0, xa.ptr(), xa.ld, 0, x6.npix, 3, 1.0, 0.0, hip.stream_ptr()), 'axpb')
        hip.check(hip.load().vps_axpb(x6.ptr(), x6.ld, 3, xb.ptr(), xb.ld, 0, x6.npix, 3, 1.0, 0.0, hip.stream_ptr()), 'axpb')
        # single-consumer encoder maps: temporaries of the caller's scope (FlowNet2.run wraps every sub-network in `ws.scope()`)
        c1a = p['conv1'](xa, ws=ws, name=tag + 'c1a', temp=True)
        c2a = p['conv2'](c1a, out=cat2.window(0, 128), ws=ws)
        ws.release(c1a)
        c3a = p['conv3'](c2a, ws=ws, name=tag + 'c3a', temp=True)
        c1b = p['conv1'](xb, ws=ws, name=tag + 'c1b', temp=True)
        c2b = p['conv2'](c1b, ws=ws, name=tag + 'c2b', temp=True)
        ws.release(c1b)
        c3b = p['conv3'](c2b, ws=ws, name=tag + 'c3b', temp=True)
        ws.release(c2b)
        in31 = ws.fmap(tag + 'in31', N, c3a.H, c3a.W, 473)
        p['conv_redir'](c3a, out=in31.window(0, 32), ws=ws)
        nhwc.correlation(c3a, c3b, in31.window(32, 441), 20, 2, hip.ACT_LEAKY, 0.1, prec=p['conv_redir'].prec)   # corr + corr_activation
        ws.release(c3a, c3b)
        p['conv3_1'](in31, out=cat3.window(0, 256), ws=ws)
        t = p['conv4'](cat3.window(0, 256), ws=ws, name=tag + 'c4', temp=True)
        p['conv4_1'](t, out=cat4.window(0, 512), ws=ws)
        t = p['conv5'](cat4.window(0, 512), ws=ws, name=tag + 'c5', temp=True)
        p['conv5_1'](t, out=cat5.window(0, 512), ws=ws)
        t = p['conv6'](cat5.window(0, 512), ws=ws, name=tag + 'c6', temp=True)
        c6 = p['conv6_1'](t, ws=ws, name=tag + 'c61', temp=True)
        return self._decode(ws, tag, [cat2, cat3, cat4, cat5], c6)


class FlowNetS(_RefineS):
    def __init__(self, input_channels=12):
        super().__init__()
        self.conv1 = _conv(input_channels, 64, 7, 2); self.conv2 = _conv(64, 128, 5, 2); self.conv3 = _conv(128, 256, 5, 2)
        self.conv3_1 = _conv(256, 256)
        self.conv4 = _conv(256, 512, stride=2); self.conv4_1 = _conv(512, 512)
        self.conv5 = _conv(512, 512, stride=2); self.conv5_1 = _conv(512, 512)
        self.conv6 = _conv(512, 1024, stride=2); self.conv6_1 = _conv(1024, 1024)
        self._mk_decoder(up_bias=False)

    def run(self, x12, ws, tag):
        """FlowNetS.py:59-94."""
        p = self._p
        N, H, W = x12.N, x12.H, x12.W
        h2, w2 = H // 4, W // 4
        cat2, cat3, cat4, cat5 = self._cats(ws, tag, N, h2, w2)
        c1 = p['conv1'](x12, ws=ws, name=tag + 'c1', temp=True)
        p['conv2'](c1, out=cat2.window(0, 128), ws=ws)
        ws.release(c1)
        t = p['conv3'](cat2.window(0, 128), ws=ws, name=tag + 'c3', temp=True)
        p['conv3_1'](t, out=cat3.window(0, 256), ws=ws)
        t = p['conv4'](cat3.window(0, 256), ws=ws, name=tag + 'c4', temp=True)
        p['conv4_1'](t, out=cat4.window(0, 512), ws=ws)
        t = p['conv5'](cat4.window(0, 512), ws=ws, name=tag + 'c5', temp=True)
        p['conv5_1'](t, out=cat5.window(0, 512), ws=ws)
        t = p['conv6'](cat5.window(0, 512), ws=ws, name=tag + 'c6', temp=True)
        c6 = p['conv6_1'](t, ws=ws, name=tag + 'c61', temp=True)
        return self._decode(ws, tag, [cat2, cat3, cat4, cat5], c6)


class FlowNetSD(_Net):
    def __init__(self):
        super().__init__()
        self.conv0 = _conv(6, 64)
        self.conv1 = _conv(64, 64, stride=2); self.conv1_1 = _conv(64, 128)
        self.conv2 = _conv(128, 128, stride=2); self.conv2_1 = _conv(128, 128)
        self.conv3 = _conv(128, 256, stride=2); self.conv3_1 = _conv(256, 256)
        self.conv4 = _conv(256, 512, stride=2); self.conv4_1 = _conv(512, 512)
        self.conv5 = _conv(512, 512, stride=2); self.conv5_1 = _conv(512, 512)
        self.conv6 = _conv(512, 1024, stride=2); self.conv6_1 = _conv(1024, 1024)
        self.deconv5 = _deconv(1024, 512); self.deconv4 = _deconv(1026, 256)
        self.deconv3 = _deconv(770, 128); self.deconv2 = _deconv(386, 64)
        self.inter_conv5 = _iconv(1026, 512); self.inter_conv4 = _iconv(770, 256)
        self.inter_conv3 = _iconv(386, 128); self.inter_conv2 = _iconv(194, 64)
        self.predict_flow6 = _pred(1024); self.predict_flow5 = _pred(512); self.predict_flow4 = _pred(256)
        self.predict_flow3 = _pred(128); self.predict_flow2 = _pred(64)
        for n in ('6_to_5', '5_to_4', '4_to_3', '3_to_2'):
            setattr(self, 'upsampled_flow' + n, nn.ConvTranspose2d(2, 2, 4, 2, 1))

    def run(self, x6, ws, tag):
        """FlowNetSD.py:66-105."""
        p = self._p
        N, H, W = x6.N, x6.H, x6.W
        h2, w2 = H // 4, W // 4
        cat2 = ws.fmap(tag + 'cat2', N, h2, w2, 194)
        cat3 = ws.fmap(tag + 'cat3', N, h2 // 2, w2 // 2, 386)
        cat4 = ws.fmap(tag + 'cat4', N, h2 // 4, w2 // 4, 770)
        cat5 = ws.fmap(tag + 'cat5', N, h2 // 8, w2 // 8, 1026)
        c0 = p['conv0'](x6.window(0, 6), ws=ws, name=tag + 'c0', temp=True)
        t = p['conv1'](c0, ws=ws, name=tag + 'c1', temp=True)
        ws.release(c0)                                   # 537 MB at 1024x2048: the largest map of the frame
        c1 = p['conv1_1'](t, ws=ws, name=tag + 'c11', temp=True)
        ws.release(t)
        t = p['conv2'](c1, ws=ws, name=tag + 'c2', temp=True)
        ws.release(c1)
        p['conv2_1'](t, out=cat2.window(0, 128), ws=ws)
        ws.release(t)
        t = p['conv3'](cat2.window(0, 128), ws=ws, name=tag + 'c3', temp=True)
        p['conv3_1'](t, out=cat3.window(0, 256), ws=ws)
        t = p['conv4'](cat3.window(0, 256), ws=ws, name=tag + 'c4', temp=True)
        p['conv4_1'](t, out=cat4.window(0, 512), ws=ws)
        t = p['conv5'](cat4.window(0, 512), ws=ws, name=tag + 'c5', temp=True)
        p['conv5_1'](t, out=cat5.window(0, 512), ws=ws)
        t = p['conv6'](cat5.window(0, 512), ws=ws, name=tag + 'c6', temp=True)
        c6 = p['conv6_1'](t, ws=ws, name=tag + 'c61', temp=True)
        flow = p['predict_flow6'](c6, ws=ws, name=tag + 'flow6')
        feat = c6
        for lvl, cat, enc_c, dec_c in ((5, cat5, 512, 512), (4, cat4, 512, 256), (3, cat3, 256, 128), (2, cat2, 128, 64)):
            p['upsampled_flow%d_to_%d' % (lvl + 1, lvl)](flow, out=cat.window(enc_c + dec_c, 2), ws=ws)
            p['deconv%d' % lvl](feat, out=cat.window(enc_c, dec_c), ws=ws)
            inter = p['inter_conv%d' % lvl](cat, ws=ws, name='%sinter%d' % (tag, lvl), temp=True)
            flow = p['predict_flow%d' % lvl](inter, ws=ws, name='%sflow%d' % (tag, lvl))
            ws.release(inter)
            feat = cat
        return flow


class FlowNetFusion(_Net):
    def __init__(self):
        super().__init__()
        self.conv0 = _conv(11, 64)
        self.conv1 = _conv(64, 64, stride=2); self.conv1_1 = _conv(64, 128)
        self.conv2 = _conv(128, 128, stride=2); self.conv2_1 = _conv(128, 128)
        self.deconv1 = _deconv(128, 32); self.deconv0 = _deconv(162, 16)
        self.inter_conv1 = _iconv(162, 32); self.inter_conv0 = _iconv(82, 16)
        self.predict_flow2 = _pred(128); self.predict_flow1 = _pred(32); self.predict_flow0 = _pred(16)
        self.upsampled_flow2_to_1 = nn.ConvTranspose2d(2, 2, 4, 2, 1)
        self.upsampled_flow1_to_0 = nn.ConvTranspose2d(2, 2, 4, 2, 1)

    def run(self, x11, ws, tag):
        """FlowNetFusion.py:47-66."""
        p = self._p
        N, H, W = x11.N, x11.H, x11.W
        cat0 = ws.fmap(tag + 'cat0', N, H, W, 82)
        cat1 = ws.fmap(tag + 'cat1', N, H // 2, W // 2, 162)
        p['conv0'](x11, out=cat0.window(0, 64), ws=ws)
        t = p['conv1'](cat0.window(0, 64), ws=ws, name=tag + 'c1', temp=True)
        p['conv1_1'](t, out=cat1.window(0, 128), ws=ws)
        ws.release(t)
        t = p['conv2'](cat1.window(0, 128), ws=ws, name=tag + 'c2', temp=True)
        c2 = p['conv2_1'](t, ws=ws, name=tag + 'c21', temp=True)
        ws.release(t)
        flow2 = p['predict_flow2'](c2, ws=ws, name=tag + 'flow2')
        p['upsampled_flow2_to_1'](flow2, out=cat1.window(160, 2), ws=ws)
        p['deconv1'](c2, out=cat1.window(128, 32), ws=ws)
        ws.release(c2)
        inter1 = p['inter_conv1'](cat1, ws=ws, name=tag + 'inter1', temp=True)
        flow1 = p['predict_flow1'](inter1, ws=ws, name=tag + 'flow1')
        ws.release(inter1)
        p['upsampled_flow1_to_0'](flow1, out=cat0.window(80, 2), ws=ws)
        p['deconv0'](cat1, out=cat0.window(64, 16), ws=ws)
        inter0 = p['inter_conv0'](cat0, ws=ws, name=tag + 'inter0', temp=True)
        return p['predict_flow0'](inter0, ws=ws, name=tag + 'flow0', keep=True)       # the network's output: the caller's `out` workspace


class FlowNet2(HipModule):
    """flownet2.py:32-198 (batchNorm=False, div_flow=20, rgb_max=255)."""

    def __init__(self, args=None, batchNorm=False, div_flow=20., requires_grad=False):
        super().__init__()
        assert not batchNorm
        self.div_flow = float(div_flow)
        self.rgb_max = float(getattr(args, 'rgb_max', 255.0)) if args is not None else 255.0
        self.flownetc = FlowNetC()
        self.flownets_1 = FlowNetS()
        self.flownets_2 = FlowNetS()
        self.flownets_d = FlowNetSD()
        self.flownetfusion = FlowNetFusion()
        for prm in self.parameters():
            prm.requires_grad = False

    def pack(self, device):
        for n in (self.flownetc, self.flownets_1, self.flownets_2, self.flownets_d, self.flownetfusion):
            n.pack(device)
        for n in (self.flownetc, self.flownets_1, self.flownets_2):
            n.cat_tag = 'fn2.CSS.'
        self._nblk = 512

    def run(self, img, ref, mean, std, ws, tag='fn2.', sd_stream=None):
        """img/ref: NCHW [1,3,H,W] normalised device tensors (the detector inputs); mean/std: device [3] tensors.
        Returns the full-resolution flow FMap [1,H,W,2(+2)] (compute_flow before the x0.25 resize).
        sd_stream: FlowNetSD needs the image pair alone (flownet2.py:166-169 runs it after FlowNetS_2, but nothing of the
        C -> S1 -> S2 chain feeds it): with a second stream it runs BESIDE that chain - its low-resolution layers (<= 64x128: a few
        dozen workgroups each) fill CUs the chain's leave idle and vice versa. Forked behind the pair buffer, joined in front of
        the fusion stage; disjoint workspace names, one split-K scratch per stream: results are bitwise those of the serial order."""
        self.ensure_packed(img.device)
        assert abs(self.rgb_max - 255.0) < 1e-6
        _, _, H0, W0 = img.shape
        # panoptic_fusetrack.py:125-128: the two sizes the reference pads with zeros (bottom / right) before FlowNet2 ...
        H, W = {(800, 1600): (832, 1664), (200, 400): (256, 448)}.get((H0, W0), (H0, W0))
        assert H % 64 == 0 and W % 64 == 0, 'Flownet input must be divisible by 64.'
        lib = hip.load()
        sp = hip.stream_ptr
        x6 = ws.fmap(tag + 'x6', 1, H, W, 6, ld=8)
        partial = ws.get(tag + 'partial', (3 * self._nblk,), dtype=torch.float64)
        rgb_mean = ws.get(tag + 'rgb_mean', (4,))
        img = img.contiguous(); ref = ref.contiguous()
        hip.check(lib.vps_flow_prep_pad(hip.ptr(img), hip.ptr(ref), hip.ptr(mean), hip.ptr(std), x6.ptr(), x6.ld, H0, W0, H, W,
                                        hip.ptr(partial), self._nblk, hip.ptr(rgb_mean), sp()), 'vps_flow_prep_pad')
        D = self.div_flow

        def full(mode, flow_a, flow_b, out):
            fb = flow_a if flow_b is None else flow_b
            hip.check(lib.vps_flow_stage_full(x6.ptr(), x6.ld, flow_a.ptr(), flow_a.ld, flow_a.coff, fb.ptr(), fb.ld, fb.coff, H, W, mode, D,
                                              out.ptr(), out.ld, sp()), 'vps_flow_stage_full')

        # Every sub-network runs inside a workspace scope: its single-consumer maps are temporaries that go back to the stream's pool at
        # the end, so FlowNetC / S1 / S2 / Fusion reuse ONE set of blocks (FlowNetSD on its own stream has its own)
        cur = sd_flow2 = None
        if sd_stream is not None:
            cur = torch.cuda.current_stream(img.device)
            sd_stream.wait_stream(cur)            # the pair buffer is written (and this workspace's previous SD pass was joined below)
            with torch.cuda.stream(sd_stream), ws.scope():
                sd_flow2 = self.flownets_d.run(x6, ws, tag + 'SD.')
        # whole 12-float pixels per thread (vps_flow_stage_full); the channel-wise vps_flow_stage + vps_axpb pair is the general form
        with ws.scope():
            c_flow2 = self.flownetc.run(x6, ws, tag + 'C.')
        concat1 = ws.fmap(tag + 'concat1', 1, H, W, 12, temp=True)
        full(0, c_flow2, None, concat1)                                # flownet2.py:142-151
        with ws.scope():
            s1_flow2 = self.flownets_1.run(concat1, ws, tag + 'S1.')
        ws.release(concat1)
        concat2 = ws.fmap(tag + 'concat2', 1, H, W, 12, temp=True)
        full(0, s1_flow2, None, concat2)                               # :154-163
        with ws.scope():
            s2_flow2 = self.flownets_2.run(concat2, ws, tag + 'S2.')
        ws.release(concat2)
        if sd_stream is None:
            with ws.scope():
                sd_flow2 = self.flownets_d.run(x6, ws, tag + 'SD.')
        else:
            cur.wait_stream(sd_stream)
        concat3 = ws.fmap(tag + 'concat3', 1, H, W, 11)
        full(1, s2_flow2, sd_flow2, concat3)                           # :166-187 nearest x4 of flow*20 resp. flow/20 (sic)
        with ws.scope():
            flow = self.flownetfusion.run(concat3, ws, tag + 'F.')
        if (H, W) != (H0, W0):
            # ... and trims afterwards (:135-138, index_select of the first H0 rows / W0 columns)
            trimmed = ws.fmap(tag + 'flow_trim', 1, H0, W0, flow.C, ld=flow.ld, out=True)
            trimmed.t.copy_(flow.t[:, :H0, :W0, :])
            flow = trimmed
        # (debugging: concat1 / concat2 are temporaries - their contents are valid until the next launch on this stream's pool)
        self._last = dict(x6=x6, c_flow2=c_flow2, concat1=concat1, s1_flow2=s1_flow2, concat2=concat2,
                          s2_flow2=s2_flow2, sd_flow2=sd_flow2, concat3=concat3)
        return flow

    def forward(self, inputs):
        """reference signature: inputs [B,3,2,H,W] RGB 0..255 -> flow [B,2,H,W] (flownet2.py:133)."""
        assert inputs.shape[0] == 1
        dev = inputs.device
        ws = nhwc.Workspace(dev)
        ws.pooling = False
        one = torch.ones(3, device=dev); zero = torch.zeros(3, device=dev)
        flow = self.run(inputs[:, :, 0].contiguous(), inputs[:, :, 1].contiguous(), zero, one, ws)
        return flow.to_nchw()
