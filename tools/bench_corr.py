"""The two correlation shapes of the path (FlowNetC: 441 channels @128x256, stride-2 displacements; LiteFlowNetCorr: 81 channels @256x512)
on vps_correlation_f16 (split fp16 on the matrix cores) and vps_correlation (exact vector-ALU kernels): us per call, max difference.
VPS_CORR_MFMA=3 enables the matrix-core instance for both shapes (default: the stride-2 shape only)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vps_amd import hip, nhwc


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device('cuda:0')
    for name, H, W, md, s2 in (('FlowNetC 441ch @128x256', 128, 256, 20, 2), ('LiteFlowNetCorr 81ch @256x512', 256, 512, 4, 1)):
        g = torch.Generator().manual_seed(1)
        a = nhwc.FMap(torch.randn(1, H, W, 256, generator=g).to(dev)); b = nhwc.FMap(torch.randn(1, H, W, 256, generator=g).to(dev))
        D = (2 * (md // s2) + 1) ** 2
        o1 = nhwc.FMap(torch.zeros(1, H, W, (D + 3) // 4 * 4, device=dev), D, 0); o2 = nhwc.FMap(torch.zeros(1, H, W, (D + 3) // 4 * 4, device=dev), D, 0)
        t_f16 = timed(lambda: nhwc.correlation(a, b, o1, md, s2, prec=hip.PREC_F16X3))
        t_ex = timed(lambda: nhwc.correlation(a, b, o2, md, s2, prec=hip.PREC_F32))
        err = float((o1.t[..., :D] - o2.t[..., :D]).abs().max() / o2.t[..., :D].abs().max())
        print('%-32s f16x3 path %7.1f us   exact %7.1f us   max |diff| / max %.1e' % (name, t_f16, t_ex, err), flush=True)


if __name__ == '__main__':
    main()
