"""The stride-1 3x3 layers with 128-column tiles on vps_conv2d: pipelined 8-wave halo kernel (conv_h8p.hip, round 6) against
conv_mfma_h8_kernel (VPS_H8P=0) in ONE process - time per launch, bitwise comparison, both against a torch fp64 reference on a crop."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from vps_amd import hip, nhwc

SHAPES = [  # name, cin, cout, H, W, residual
    ('fpn/tcea 256->256 3x3 @256x512', 256, 256, 256, 512, 0),
    ('256->256 3x3 @128x256', 256, 256, 128, 256, 0),
    ('128->128 3x3 @256x512', 128, 128, 256, 512, 0),
    ('473->256 3x3 @128x256', 473, 256, 128, 256, 0),
    ('256->256 3x3 @256x512 +res', 256, 256, 256, 512, 1),
    ('256->256 3x3 @250x500 (ragged patches)', 256, 256, 250, 500, 0),
]


def run(pc, x, out, ws, res, reps):
    pc(x, out=out, ws=ws, res=res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        pc(x, out=out, ws=ws, res=res)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device('cuda:0')
    ws = nhwc.Workspace(dev)
    rows = []
    for name, cin, cout, H, W, rk in SHAPES:
        g = torch.Generator().manual_seed(1)
        w = torch.randn(cout, cin, 3, 3, generator=g) * 0.03
        b = torch.randn(cout, generator=g) * 0.1
        pc = nhwc.PackedConv(w, b, None, stride=1, padding=1, act=hip.ACT_LEAKY, device=dev, prec=hip.PREC_F16X3)
        cp = (cin + 3) // 4 * 4
        xt = torch.zeros(1, H, W, cp, device=dev)
        xt[..., :cin] = torch.randn(1, H, W, cin, device=dev)
        x = nhwc.FMap(xt, cin, 0)
        res = nhwc.FMap(torch.randn(1, H, W, cout, device=dev), cout, 0) if rk else None
        outs, us = {}, {}
        for mode in ('0', '1'):
            os.environ['VPS_H8P'] = mode
            pc.__dict__.pop('_dcache', None)
            out = nhwc.FMap(torch.zeros(1, H, W, cout, device=dev), cout, 0)
            us[mode] = run(pc, x, out, ws, res, 10)
            outs[mode] = out.t.clone()
        # reference on the top-left 40 x 72 crop (fp64 on the device)
        ch, cw = min(H, 40), min(W, 72)
        xin = xt[0, :ch + 1, :cw + 1, :cin].permute(2, 0, 1)[None].double()
        ref = F.conv2d(xin, w.double().to(dev), b.double().to(dev), padding=1)[0, :, :ch, :cw].permute(1, 2, 0)
        if res is not None:
            ref = ref + res.t[0, :ch, :cw].double()
        ref = F.leaky_relu(ref, 0.1).float()
        err = {m: float((outs[m][0, :ch, :cw] - ref).abs().max() / ref.abs().max()) for m in outs}
        same = bool(torch.equal(outs['0'], outs['1']))
        fl = pc.flops(1, H, W)
        rows.append(dict(layer=name, us_h8=round(us['0'], 1), us_h8p=round(us['1'], 1), bitwise_equal=same, err_h8=err['0'], err_h8p=err['1']))
        print('%-42s h8 %7.1f us (%.0f TF)  h8p %7.1f us (%.0f TF)  x%.2f  bitwise %s  err %.1e / %.1e' % (
            name, us['0'], fl / us['0'] / 1e6, us['1'], fl / us['1'] / 1e6, us['0'] / us['1'], same, err['0'], err['1']), flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(rows, open('gpurun_out/bench_h8p.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
