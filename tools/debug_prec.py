"""GPU debug: run the same frame with two conv arithmetic modes and compare every workspace buffer"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
import numpy as np, torch
import vps_amd
from vps_amd import synth, nhwc, hip
dev = torch.device('cuda:0')
cfg = vps_amd.Config.fromfile('configs/cityscapes/fusetrack.py')
H, W = 128, 256
fr = synth.synth_clip(H, W, 3, 0)
res = {}
for name, prec in (('f32', hip.PREC_F32), ('x6', hip.PREC_BF16X6), ('x3', hip.PREC_BF16X3)):
    nhwc.DEFAULT_PREC = prec
    m = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    synth.load_synth(m, 0)
    snaps = []
    for t in range(3):
        out = m(return_loss=False, rescale=True, img=[fr[t].to(dev)], img_meta=[[synth.img_meta(H, W, 10001 + t)]], ref_img=[fr[t - 1 if t else 0].to(dev)])
        torch.cuda.synchronize()
        snaps.append({k: v.detach().float().cpu().clone() for k, v in m._ws.bufs.items() if v.dtype in (torch.float32,)})
    res[name] = snaps
with open('gpurun_out/debug_prec.txt', 'w') as f:
    for other in ('x6', 'x3'):
        for t in range(3):
            f.write('==== %s vs f32, frame %d\n' % (other, t))
            rows = []
            for k, a in res['f32'][t].items():
                b = res[other][t].get(k)
                if b is None or b.shape != a.shape or k.startswith('__'):
                    continue
                d = float((a - b).abs().max()); mx = float(a.abs().max())
                rows.append((d / max(mx, 1e-12), k, mx))
            for r, k, mx in sorted(rows, reverse=True)[:25]:
                f.write('  %-28s rel %.3e (max %.3e)\n' % (k, r, mx))
with open('gpurun_out/debug_prec.txt', 'a') as f:
    for key in ('sem.l0t0dcn', 'sem.l0t0off', 'neck.out0'):
        a = res['f32'][0][key]; b = res['x6'][0][key]
        err = (a - b).abs()
        thr = 1e-3 * float(a.abs().max())
        bad = (err > thr).nonzero()
        f.write('---- %s shape %s: %d elements above %.3e\n' % (key, tuple(a.shape), bad.shape[0], thr))
        if bad.shape[0]:
            f.write('   y unique: %s\n   x unique: %s\n   c unique count %d (first %s)\n' % (
                bad[:, 1].unique().tolist()[:40], bad[:, 2].unique().tolist()[:70], bad[:, 3].unique().numel(), bad[:, 3].unique().tolist()[:20]))
            for row in bad[:8].tolist():
                f.write('   %s f32 %.5f x6 %.5f\n' % (row, float(a[tuple(row)]), float(b[tuple(row)])))
print(open('gpurun_out/debug_prec.txt').read())
