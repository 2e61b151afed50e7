"""ResNet backbone on libvpship (registry name `ResNet`).

Mirrors mmdet/models/backbones/resnet.py (Bottleneck :86-266, ResNet :332-526; style='pytorch', eval BatchNorm)
at the parameter-name level; every conv+BN(+ReLU)(+residual add) is ONE fused implicit-GEMM launch.
"""
import torch
import torch.nn as nn

from . import hip, nhwc
from .base import HipModule
from .registry import BACKBONES


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)   # stride on the 3x3 ('pytorch')
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))


@BACKBONES.register_module
class ResNet(HipModule):
    arch_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3),
                 style='pytorch', frozen_stages=-1, norm_eval=True, zero_init_residual=True, **unused):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError('invalid depth {} for resnet'.format(depth))
        assert style == 'pytorch' and tuple(dilations) == (1, 1, 1, 1), 'only the fusetrack.py configuration is on the path'
        self.depth, self.out_indices = depth, tuple(out_indices)
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        self.res_layers = []
        for i, nb in enumerate(self.arch_settings[depth][:num_stages]):
            planes = 64 * 2 ** i
            blocks = [Bottleneck(inplanes, planes, strides[i], downsample=True)]
            inplanes = planes * 4
            blocks += [Bottleneck(inplanes, planes) for _ in range(1, nb)]
            name = 'layer%d' % (i + 1)
            self.add_module(name, nn.Sequential(*blocks))
            self.res_layers.append(name)
        self.feat_dim = inplanes

    def pack(self, device):
        P = nhwc.pack_conv_module
        self._stem = P(self.conv1, self.bn1, hip.ACT_RELU, device=device)
        self._blocks = []
        for name in self.res_layers:
            stage = []
            for b in getattr(self, name):
                stage.append(dict(
                    c1=P(b.conv1, b.bn1, hip.ACT_RELU, device=device),
                    c2=P(b.conv2, b.bn2, hip.ACT_RELU, device=device),
                    c3=P(b.conv3, b.bn3, hip.ACT_RELU, device=device),     # ReLU applied after the residual add
                    ds=P(b.downsample[0], b.downsample[1], hip.ACT_NONE, device=device) if b.downsample is not None else None))
            self._blocks.append(stage)

    def run(self, x, ws, tag):
        """x: FMap [1,H,W,3(+pad)] normalised RGB. Returns [C2..C5] FMaps (resnet.py:506-517)."""
        self.ensure_packed(x.t.device)
        # every map below is produced and consumed on the current stream: temporaries of the workspace, released after their last
        # consumer is enqueued (the stage outputs are the caller's to release: FPN's lateral convolutions read them)
        t = self._stem(x, ws=ws, name=tag + 'stem', temp=True)
        p = ws.fmap(tag + 'pool', t.N, (t.H + 1) // 2, (t.W + 1) // 2, 64, temp=True)
        x = nhwc.pool3x3s2(t, p, 'max')
        ws.release(t)
        outs = []
        for si, stage in enumerate(self._blocks):
            for bi, b in enumerate(stage):
                n = '%sl%d.%d.' % (tag, si + 1, bi)
                idn = x if b['ds'] is None else b['ds'](x, ws=ws, name=n + 'ds', temp=True)
                y1 = b['c1'](x, ws=ws, name=n + 'c1', temp=True)
                y2 = b['c2'](y1, ws=ws, name=n + 'c2', temp=True)
                ws.release(y1)
                xn = b['c3'](y2, ws=ws, name=n + 'c3', res=idn, temp=True)
                ws.release(y2, idn if idn is not x else None)
                if not outs or x is not outs[-1]:
                    ws.release(x)                      # the previous block's output (not a stage output: those stay for the caller)
                x = xn
            if si in self.out_indices:
                outs.append(x)
        if not outs or x is not outs[-1]:
            ws.release(x)
        return outs

    def forward(self, x):
        """NCHW operator-level API (reference call signature): tuple of NCHW stage outputs."""
        ws = nhwc.Workspace(x.device)
        ws.pooling = False                  # one-off operator call: plain buffers
        return tuple(o.to_nchw() for o in self.run(nhwc.from_nchw(x), ws, 'bb.'))
