"""runs a handful of big vps_conv2d launches (for rocprofv3 --pmc passes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vps_amd import hip, nhwc
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device('cuda:0')
ws = nhwc.Workspace(dev)
for cin, cout, k, s, p, H, W in [(256, 256, 3, 1, 1, 256, 512), (512, 256, 1, 1, 0, 256, 512), (64, 128, 5, 2, 2, 512, 1024),
                                  (64, 64, 3, 1, 1, 256, 512), (6, 64, 3, 1, 1, 1024, 2048), (512, 512, 3, 1, 1, 64, 128)]:
    w = torch.randn(cout, cin, k, k) * 0.05
    pc = nhwc.PackedConv(w, torch.zeros(cout), None, stride=s, padding=p, act=hip.ACT_LEAKY, device=dev, prec=prec)
    x = nhwc.FMap(torch.randn(1, H, W, (cin + 3) // 4 * 4, device=dev), cin, 0)
    for _ in range(3):
        pc(x, ws=ws, name='o%d%d%d' % (cin, cout, k))
torch.cuda.synchronize()
print('done')
