"""MaskRemoval's box walk at 1024x2048 on crowded synthetic lists: us per call of the modes of vps_amd.panoptic_ops.MaskRemoval
(hist = vps_mask_removal_hist, dep = vps_mask_removal_dep, level = the per-level launches), kept lists compared. GPU box only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vps_amd import hip, nhwc, panoptic_ops as P


def main():
    dev = torch.device('cuda:0')
    H, W, S = 1024, 2048, 28
    P.HIST_MAX_PAIRS = 10 ** 9       # time the pattern kernels on every list (the detector keeps lists with > 200 intersecting pairs on `dep`)
    for n, ncls, seed in ((100, 8, 3), (100, 2, 4), (100, 1, 5), (60, 8, 2), (45, 3, 7)):
        rg = np.random.default_rng(seed)
        cx = rg.uniform(0, W, n); cy = rg.uniform(0, H, n)
        bw = np.exp(rg.uniform(np.log(2), np.log(900), n)); bh = np.exp(rg.uniform(np.log(2), np.log(600), n))
        rows = np.zeros((n, 8), dtype=np.float32)
        rows[:, 1] = cx - bw / 2; rows[:, 2] = cy - bh / 2; rows[:, 3] = cx + bw / 2; rows[:, 4] = cy + bh / 2
        rows[:, 5] = np.sort(rg.uniform(0.6, 1.0, n))[::-1]
        rows[:, 6] = rg.integers(1, ncls + 1, n)
        rows[:, 7] = np.arange(n)
        rows_d = torch.from_numpy(rows).to(dev)
        masks = (torch.from_numpy(rg.standard_normal((n, S, S)).astype(np.float32)) * 2 + 0.6).to(dev)
        cm = {c: 10 + c for c in range(1, ncls + 1)}
        res = {}
        for mode in ('hist', 'dep', 'level'):
            P.MASK_REMOVAL_MODE = mode
            ws = nhwc.Workspace(dev)
            mr = P.MaskRemoval(0.3)
            for _ in range(3):
                out = mr(rows, rows_d, masks, (H, W), ws, cm)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(20):
                out = mr(rows, rows_d, masks, (H, W), ws, cm)
            e1.record(); torch.cuda.synchronize()
            k = int(out['kinfo'].cpu().numpy()[0])
            res[mode] = (e0.elapsed_time(e1) / 20 * 1e3, k, out['keep'][:k].cpu().numpy().copy(), 'mr.hist' in ws.bufs)
        same = all(res[m][1] == res['level'][1] and np.array_equal(res[m][2], res['level'][2]) for m in ('hist', 'dep'))
        print('n %3d classes %d (max per class %d): hist %7.1f us (pattern kernel ran: %s)  dep %7.1f us  level %7.1f us  kept %d  equal %s' % (
            n, ncls, int(np.bincount(rows[:, 6].astype(np.int64)).max()), res['hist'][0], res['hist'][3], res['dep'][0], res['level'][0], res['hist'][1], same), flush=True)


if __name__ == '__main__':
    main()
