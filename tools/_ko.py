import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vps_amd import hip, nhwc
dev = torch.device('cuda:0')
ws = nhwc.Workspace(dev)
SH = [(256, 256, False), (512, 256, False), (64, 256, True)]
for cin, cout, res in SH:
    H, W = (256, 512) if cout != 1024 else (64, 128)
    g = torch.Generator().manual_seed(0)
    w = torch.randn(cout, cin, 1, 1, generator=g) * 0.05
    pc = nhwc.PackedConv(w, torch.zeros(cout), None, stride=1, padding=0, act=hip.ACT_RELU, device=dev, prec=hip.PREC_F16X3)
    x = nhwc.FMap(torch.randn(1, H, W, cin, device=dev), cin, 0)
    r = nhwc.FMap(torch.randn(1, H, W, cout, device=dev), cout, 0) if res else None
    out = ws.fmap('o%d_%d' % (cin, cout), 1, H, W, cout)
    for ko, ul in [('0', '3'), ('2', '3'), ('4', '3'), ('8', '3'), ('16', '3'), ('32', '3'), ('6', '3'), ('14', '3'), ('30', '3'), ('62', '3'), ('0', '0')]:
        os.environ['VPS_KO'] = ko
        os.environ['VPS_UNIFORM_LEAD'] = ul
        pc.__dict__.pop('_dcache', None)
        for _ in range(3):
            pc(x, out=out, ws=ws, res=r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        reps = 20
        e0.record()
        for _ in range(reps):
            pc(x, out=out, ws=ws, res=r)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        mb = pc.bytes(1, H, W, res) / 1e6
        print('%4d->%4d res=%d ko=%s ul=%s: %7.1f us  %6.2f TB/s(algorithmic %.0f MB)' % (cin, cout, res, ko, ul, us, mb / us, mb), flush=True)
