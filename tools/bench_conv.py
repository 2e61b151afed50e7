"""Micro-benchmark of vps_conv2d on the layer shapes of the FuseTrack path at 1024x2048 (GPU box only)."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vps_amd import hip, nhwc

SHAPES = [  # name, cin, cout, k, stride, pad, H, W, transposed, deform
    ('resnet.stem 3->64 7x7s2 @1024x2048', 3, 64, 7, 2, 3, 1024, 2048, False, False),
    ('resnet.l1 64->64 3x3 @256x512', 64, 64, 3, 1, 1, 256, 512, False, False),
    ('resnet.l1 64->256 1x1 @256x512', 64, 256, 1, 1, 0, 256, 512, False, False),
    ('resnet.l1 256->64 1x1 @256x512', 256, 64, 1, 1, 0, 256, 512, False, False),
    ('resnet.l2 128->128 3x3 @128x256', 128, 128, 3, 1, 1, 128, 256, False, False),
    ('resnet.l3 256->256 3x3 @64x128', 256, 256, 3, 1, 1, 64, 128, False, False),
    ('resnet.l3 1024->256 1x1 @64x128', 1024, 256, 1, 1, 0, 64, 128, False, False),
    ('resnet.l4 512->512 3x3 @32x64', 512, 512, 3, 1, 1, 32, 64, False, False),
    ('resnet.l4 512->2048 1x1 @32x64', 512, 2048, 1, 1, 0, 32, 64, False, False),
    ('fpn/tcea 256->256 3x3 @256x512', 256, 256, 3, 1, 1, 256, 512, False, False),
    ('tcea 512->256 1x1 @256x512', 512, 256, 1, 1, 0, 256, 512, False, False),
    ('flownetSD conv0 6->64 3x3 @1024x2048', 6, 64, 3, 1, 1, 1024, 2048, False, False),
    ('fusion conv0 11->64 3x3 @1024x2048', 11, 64, 3, 1, 1, 1024, 2048, False, False),
    ('flownetS conv1 12->64 7x7s2 @1024x2048', 12, 64, 7, 2, 3, 1024, 2048, False, False),
    ('flownet conv2 64->128 5x5s2 @512x1024', 64, 128, 5, 2, 2, 512, 1024, False, False),
    ('fusion conv1_1 64->128 3x3 @512x1024', 64, 128, 3, 1, 1, 512, 1024, False, False),
    ('flownet conv3 128->256 5x5s2 @256x512', 128, 256, 5, 2, 2, 256, 512, False, False),
    ('flownet conv3_1 256->256 3x3 @128x256', 256, 256, 3, 1, 1, 128, 256, False, False),
    ('flownet conv4_1 512->512 3x3 @64x128', 512, 512, 3, 1, 1, 64, 128, False, False),
    ('flownet conv5_1 512->512 3x3 @32x64', 512, 512, 3, 1, 1, 32, 64, False, False),
    ('flownet conv6_1 1024->1024 3x3 @16x32', 1024, 1024, 3, 1, 1, 16, 32, False, False),
    ('flownet deconv5 1024->512 4x4 @16x32', 1024, 512, 4, 2, 1, 16, 32, True, False),
    ('flownet deconv2 386->64 4x4 @128x256', 386, 64, 4, 2, 1, 128, 256, True, False),
    ('flownet predict_flow2 194->2 3x3 @256x512', 194, 2, 3, 1, 1, 256, 512, False, False),
    ('fusion deconv0 162->16 4x4 @512x1024', 162, 16, 4, 2, 1, 512, 1024, True, False),
    ('fusion interconv0 82->16 3x3 @1024x2048', 82, 16, 3, 1, 1, 1024, 2048, False, False),
    ('fusion interconv1 162->32 3x3 @512x1024', 162, 32, 3, 1, 1, 512, 1024, False, False),
    ('fusion deconv1 128->32 4x4 @256x512', 128, 32, 4, 2, 1, 256, 512, True, False),
    ('fusion conv1 64->64 3x3s2 @1024x2048', 64, 64, 3, 2, 1, 1024, 2048, False, False),
    ('flownet predict_flow3 386->2 3x3 @128x256', 386, 2, 3, 1, 1, 128, 256, False, False),
    ('fusion predict_flow0 16->2 3x3 @1024x2048', 16, 2, 3, 1, 1, 1024, 2048, False, False),
    ('rpn_cls narrow 256->3 1x1 @256x512', 256, 3, 1, 1, 0, 256, 512, False, False),
    ('rpn_cls narrow 256->3 1x1 @128x256', 256, 3, 1, 1, 0, 128, 256, False, False),
    ('rpn_cls narrow 256->3 1x1 @32x64', 256, 3, 1, 1, 0, 32, 64, False, False),
    ('upflow narrow 2->2 4x4 @512x1024', 2, 2, 4, 2, 1, 512, 1024, True, False),
    ('upflow narrow 2->2 4x4 @128x256', 2, 2, 4, 2, 1, 128, 256, True, False),
    ('upflow narrow 2->2 4x4 @16x32', 2, 2, 4, 2, 1, 16, 32, True, False),
    ('upsnet dcn 256->256 3x3 @256x512', 256, 256, 3, 1, 1, 256, 512, False, True),
    ('upsnet dcn 128->128 3x3 @256x512', 128, 128, 3, 1, 1, 256, 512, False, True),
    ('bbox fc 12544->1024 M=1000', 12544, 1024, 1, 1, 0, 1, 1000, False, False),
    ('mask conv 256->256 3x3 100x14x14', 256, 256, 3, 1, 1, 14, 14, False, False),
]


def main():
    prec = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    dev = torch.device('cuda:0')
    ws = nhwc.Workspace(dev)
    rows = []
    flt = os.environ.get('BENCH_CONV_FILTER')
    for name, cin, cout, k, s, p, H, W, tr, df in SHAPES:
        if flt and flt not in name:
            continue
        N = 100 if 'mask conv' in name else 1
        g = torch.Generator().manual_seed(0)
        w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), generator=g) * 0.05
        pc = nhwc.PackedConv(w, torch.zeros(cout), None, stride=s, padding=p, act=hip.ACT_LEAKY, transposed=tr, deform=df, device=dev, prec=prec)
        x = nhwc.FMap(torch.randn(N, H, W, (cin + 3) // 4 * 4, device=dev), cin, 0)
        off = None
        if df:
            off = nhwc.FMap(torch.randn(N, H, W, 20, device=dev) * 2.0, 18, 0)
        out = pc(x, ws=ws, name='bench_out', offset=off)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        reps = int(os.environ.get('BENCH_CONV_REPS', '5'))
        e0.record()
        for _ in range(reps):
            pc(x, out=out, ws=ws, offset=off)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = pc.flops(N, H, W)
        rows.append(dict(layer=name, ms=round(ms, 4), tflops=round(fl / ms / 1e9, 2), gflop=round(fl / 1e9, 2)))
        print('%-48s %8.3f ms  %7.2f TFLOP/s  (%.1f GFLOP)' % (name, ms, fl / ms / 1e9, fl / 1e9), flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(rows, open('gpurun_out/bench_conv_p%d.json' % prec, 'w'), indent=1)


if __name__ == '__main__':
    main()
