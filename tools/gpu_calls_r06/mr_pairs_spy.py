import numpy as np, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vps_amd import panoptic_ops as P
orig = P.MaskRemoval.forward
def pairs_of(rows_h, H, W):
    b = rows_h[:, 1:5].astype(np.int32).astype(np.int64); c = rows_h[:, 6].astype(np.int64)
    x0 = np.maximum(b[:, 0], 0); x1 = np.minimum(b[:, 2] + 1, W); y0 = np.maximum(b[:, 1], 0); y1 = np.minimum(b[:, 3] + 1, H)
    inter = (c[:, None] == c[None, :]) & (x0[:, None] < x1[None, :]) & (x0[None, :] < x1[:, None]) & (y0[:, None] < y1[None, :]) & (y0[None, :] < y1[:, None])
    area = np.maximum(x1 - x0, 0) * np.maximum(y1 - y0, 0)
    return int((inter.sum() - len(b)) // 2), int(area.sum())
def spy(self, rows_h, rows_d, mask_prob, im_shape, *a, **k):
    print('detections', rows_h.shape[0], 'pairs, box area', pairs_of(rows_h, int(im_shape[0]), int(im_shape[1])), flush=True)
    return orig(self, rows_h, rows_d, mask_prob, im_shape, *a, **k)
P.MaskRemoval.forward = spy
P.MASK_REMOVAL_MODE = 'dep'
sys.argv = ['bench.py', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-extras']
try:
    bench.main()
except SystemExit:
    pass
H, W = 1024, 2048
for n, ncls, seed in ((100, 8, 3), (100, 2, 4), (100, 1, 5), (60, 8, 2), (45, 3, 7)):
    rg = np.random.default_rng(seed)
    cx = rg.uniform(0, W, n); cy = rg.uniform(0, H, n)
    bw = np.exp(rg.uniform(np.log(2), np.log(900), n)); bh = np.exp(rg.uniform(np.log(2), np.log(600), n))
    rows = np.zeros((n, 8), dtype=np.float32)
    rows[:, 1] = cx - bw / 2; rows[:, 2] = cy - bh / 2; rows[:, 3] = cx + bw / 2; rows[:, 4] = cy + bh / 2
    rows[:, 6] = rg.integers(1, ncls + 1, n)
    print('synthetic', n, ncls, pairs_of(rows, H, W))
