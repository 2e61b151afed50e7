"""The 1x1 (pointwise) layers of the FuseTrack path at 1024x2048 on vps_conv2d: persistent pointwise kernel (conv_pw.hip) against the
uniform-lead kernel (VPS_PW=0) in ONE process - time per launch, bitwise comparison of the two, and both against a torch fp32 GEMM."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vps_amd import hip, nhwc

SHAPES = [  # name, cin, cout, stride, H, W, residual (0 none, 1 same size, 2 upsampled x2)
    ('resnet.l1 64->256 @256x512 +res', 64, 256, 1, 256, 512, 1),
    ('resnet.l1 64->256 @256x512', 64, 256, 1, 256, 512, 0),
    ('resnet.l1 256->64 @256x512', 256, 64, 1, 256, 512, 0),
    ('resnet.l1 64->64 @256x512', 64, 64, 1, 256, 512, 0),
    ('resnet.l2 256->128 @256x512', 256, 128, 1, 256, 512, 0),
    ('resnet.l2 256->512 s2 @256x512', 256, 512, 2, 256, 512, 0),
    ('fpn lateral 256->256 @256x512 +up', 256, 256, 1, 256, 512, 2),
    ('256->256 @256x512', 256, 256, 1, 256, 512, 0),
    ('tcea 512->512 @256x512', 512, 512, 1, 256, 512, 0),
    ('resnet.l2 128->512 @128x256 +res', 128, 512, 1, 128, 256, 1),
    ('resnet.l2 512->128 @128x256', 512, 128, 1, 128, 256, 0),
    ('512->256 @128x256', 512, 256, 1, 128, 256, 0),
    ('resnet.l3 256->1024 @64x128 +res', 256, 1024, 1, 64, 128, 1),
    ('resnet.l3 1024->256 @64x128', 1024, 256, 1, 64, 128, 0),
    ('resnet.l4 512->2048 @32x64 +res', 512, 2048, 1, 32, 64, 1),
]


def run(pc, x, out, ws, res, rs, reps):
    pc(x, out=out, ws=ws, res=res, res_shift=rs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        pc(x, out=out, ws=ws, res=res, res_shift=rs)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device('cuda:0')
    ws = nhwc.Workspace(dev)
    flt = os.environ.get('BENCH_PW_FILTER')
    rows = []
    for name, cin, cout, s, H, W, rk in SHAPES:
        if flt and flt not in name:
            continue
        g = torch.Generator().manual_seed(1)
        w = torch.randn(cout, cin, 1, 1, generator=g) * 0.05
        bn = dict(weight=torch.rand(cout, generator=g) + 0.5, bias=torch.randn(cout, generator=g) * 0.1,
                  running_mean=torch.randn(cout, generator=g) * 0.1, running_var=torch.rand(cout, generator=g) + 0.5)
        pc = nhwc.PackedConv(w, None, bn, stride=s, padding=0, act=hip.ACT_RELU, device=dev, prec=hip.PREC_F16X3)
        x = nhwc.FMap(torch.randn(1, H, W, cin, device=dev), cin, 0)
        Ho, Wo = H // s, W // s
        res, rs = None, 0
        if rk == 1:
            res = nhwc.FMap(torch.randn(1, Ho, Wo, cout, device=dev), cout, 0)
        elif rk == 2:
            res, rs = nhwc.FMap(torch.randn(1, Ho // 2, Wo // 2, cout, device=dev), cout, 0), 1
        outs, us = {}, {}
        for mode in ('0', '1'):
            os.environ['VPS_PW'] = mode
            pc.__dict__.pop('_dcache', None)
            out = nhwc.FMap(torch.zeros(1, Ho, Wo, cout, device=dev), cout, 0)
            us[mode] = run(pc, x, out, ws, res, rs, 20)
            outs[mode] = out.t.clone()
        # torch reference: fp32 GEMM + folded BN + residual + ReLU
        xs = x.t[:, ::s, ::s, :].reshape(-1, cin)
        sc = bn['weight'] / torch.sqrt(bn['running_var'] + 1e-5)
        sh = (0 - bn['running_mean']) * sc + bn['bias']
        ref = (xs.double() @ w.view(cout, cin).t().double().to(dev)) * sc.double().to(dev) + sh.double().to(dev)
        if res is not None:
            r = res.t
            if rs:
                r = r.repeat_interleave(2, 1).repeat_interleave(2, 2)
            ref = ref + r.reshape(-1, cout).double()
        ref = ref.clamp_min(0).float().view(1, Ho, Wo, cout)
        err = {m: float((outs[m] - ref).abs().max() / ref.abs().max()) for m in outs}
        same = bool(torch.equal(outs['0'], outs['1']))
        byts = pc.bytes(1, H, W, res is not None)
        rows.append(dict(layer=name, us_q=round(us['0'], 1), us_pw=round(us['1'], 1), bitwise_equal=same, err_q=err['0'], err_pw=err['1'],
                         floor_us_6p3=round(byts / 6.3e6, 1), mfma_floor_us=round(3 * pc.flops(1, H, W) / 2.5e9, 1)))
        print('%-40s q %7.1f us  pw %7.1f us  x%.2f  hbm floor(6.3TB/s) %6.1f  mfma floor %5.1f  bitwise %s  err %.1e / %.1e' % (
            name, us['0'], us['1'], us['0'] / us['1'], byts / 6.3e6, 3 * pc.flops(1, H, W) / 2.5e9, same, err['0'], err['1']), flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(rows, open('gpurun_out/bench_pw.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
