import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vps_amd import hip, nhwc
dev = torch.device('cuda:0')
ws = nhwc.Workspace(dev)
cin, cout, H, W = int(sys.argv[1]), int(sys.argv[2]), 256, 512
res = len(sys.argv) > 3 and sys.argv[3] == 'res'
g = torch.Generator().manual_seed(0)
w = torch.randn(cout, cin, 1, 1, generator=g) * 0.05
pc = nhwc.PackedConv(w, torch.zeros(cout), None, stride=1, padding=0, act=hip.ACT_RELU, device=dev, prec=hip.PREC_F16X3)
x = nhwc.FMap(torch.randn(1, H, W, cin, device=dev), cin, 0)
r = nhwc.FMap(torch.randn(1, H, W, cout, device=dev), cout, 0) if res else None
out = ws.fmap('o', 1, H, W, cout)
for _ in range(4):
    pc(x, out=out, ws=ws, res=r)
torch.cuda.synchronize()
