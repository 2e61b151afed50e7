"""Times the thin-input conv layers of the frame alone (20 launches each, HIP events): python tools/bench_thin.py  (VPS_THIN=0|1 in the environment)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vps_amd import hip, nhwc

dev = torch.device('cuda:0')
ws = nhwc.Workspace(dev)
for cin, k, s, H, W in ((6, 3, 1, 1024, 2048), (11, 3, 1, 1024, 2048), (3, 7, 2, 1024, 2048), (12, 7, 2, 1024, 2048)):
    g = torch.Generator().manual_seed(cin)
    w = torch.randn(64, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    pc = nhwc.PackedConv(w, torch.zeros(64), None, s, k // 2, act=hip.ACT_LEAKY, device=dev, prec=hip.PREC_F16X3)
    x = nhwc.from_nchw(torch.randn(1, cin, H, W, generator=g).to(dev))
    out = pc(x, ws=ws, name='o%d_%d' % (cin, k))
    for _ in range(3):
        pc(x, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        pc(x, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50.0
    print('VPS_THIN=%s %d->64 k%d s%d @%dx%d: %.1f us  (%.2f TB/s algorithmic)' % (os.environ.get('VPS_THIN', '1'), cin, k, s, H, W, us, pc.bytes(1, H, W) / us / 1e6), flush=True)
