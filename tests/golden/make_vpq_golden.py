"""Generates tests/golden/vpq_cases.npz by running the REAL reference function /root/reference/tools/eval_vpq.py:74-209
(vpq_compute_single_core; the module imports numpy / PIL / stdlib only) on synthetic ground-truth / prediction clips.
    python tests/golden/make_vpq_golden.py
The GPU box has no /root/reference: tests read the committed .npz only."""
import importlib
import json
import os
import sys

import numpy as np


def id2rgb(i):
    return [i % 256, (i // 256) % 256, i // 65536]


def make_clip(rng, H, W, nfr, ninst, crowd=False, void=True, unlisted=False):
    """gt / pred frames of a clip: stuff background in blocks, instances that move a little, predictions that are shifted /
    missing / of the wrong class, optional crowd region and void pixels. Returns the per-frame tuples the reference takes."""
    categories = {c: {'id': c, 'isthing': 1 if c >= 11 else 0} for c in range(19)}
    stuff_ids = {c: 1000 + c for c in range(11)}
    gt_inst = [(5000 + 97 * i, 11 + i % 8) for i in range(ninst)]              # (segment id, category)
    frames = []
    base = rng.integers(0, 11, size=((H + 15) // 16, (W + 15) // 16))
    boxes = [(int(rng.integers(0, H - 12)), int(rng.integers(0, W - 16)), int(rng.integers(6, 12)), int(rng.integers(8, 16))) for _ in range(ninst)]
    for f in range(nfr):
        gt = np.zeros((H, W), np.int64); pr = np.zeros((H, W), np.int64)
        st = base.repeat(16, 0).repeat(16, 1)[:H, :W]
        for c in range(11):
            gt[st == c] = stuff_ids[c]; pr[st == c] = stuff_ids[c] + 100 * (c % 3 == 0)     # some stuff ids differ between gt / pred
        gt_seg = {}; pr_seg = {}
        for c in np.unique(st):
            gt_seg[stuff_ids[c]] = int(c); pr_seg[stuff_ids[c] + 100 * (c % 3 == 0)] = int(c) if c != 4 else 5   # one wrong stuff class
        for i, (sid, cat) in enumerate(gt_inst):
            y, x, h, w = boxes[i]
            y = min(max(y + f, 0), H - h); x = min(max(x + 2 * f, 0), W - w)
            gt[y:y + h, x:x + w] = sid; gt_seg[sid] = cat
            if i % 5 == 3:
                continue                                                         # missed instance
            dy, dx = (0, 0) if i % 3 else (2, 3)
            pid = 9000 + 31 * i
            yy, xx = min(y + dy, H - h), min(x + dx, W - w)
            pr[yy:yy + h, xx:xx + w] = pid; pr_seg[pid] = cat if i % 7 != 6 else 11 + (cat - 10) % 8      # some wrong classes
        if void:
            gt[:3, :7] = 0
            pr[H - 2:, :5] = 0
        gt_info = [{'id': int(k), 'category_id': int(v), 'iscrowd': 0, 'area': int((gt == k).sum())} for k, v in gt_seg.items() if (gt == k).any()]
        if crowd and gt_info:
            gt_info[-1]['iscrowd'] = 1
        if unlisted and f % 2 == 1 and ninst:
            # inconsistent ground truth: instance 0 stays painted in this frame's PNG but its JSON entry is missing here (listed
            # by the other frames of the window): the reference still counts its pixels in the window's confusion map
            gt_info = [el for el in gt_info if el['id'] != gt_inst[0][0]]
        pr_info = [{'id': int(k), 'category_id': int(v), 'iscrowd': 0, 'area': int((pr == k).sum())} for k, v in pr_seg.items() if (pr == k).any()]
        to_rgb = lambda m: np.stack([m % 256, (m // 256) % 256, m // 65536], -1).astype(np.uint8)
        frames.append(({'segments_info': gt_info}, {'segments_info': pr_info}, to_rgb(gt), to_rgb(pr), {}))
    return frames, categories


def main():
    sys.path.insert(0, '/root/reference')
    ref = importlib.import_module('tools.eval_vpq')
    rng = np.random.default_rng(0)
    out = {}
    specs = [dict(H=48, W=80, nfr=5, ninst=9, crowd=False), dict(H=64, W=96, nfr=6, ninst=14, crowd=True), dict(H=40, W=64, nfr=4, ninst=0, crowd=False),
             dict(H=48, W=80, nfr=5, ninst=6, crowd=False, unlisted=True)]
    for ci, sp in enumerate(specs):
        frames, categories = make_clip(rng, **sp)
        out['clip%d_json' % ci] = np.frombuffer(json.dumps([[f[0], f[1]] for f in frames]).encode(), dtype=np.uint8)
        out['clip%d_gt' % ci] = np.stack([f[2] for f in frames]); out['clip%d_pred' % ci] = np.stack([f[3] for f in frames])
        for nf in (1, 2, 3):
            stat = ref.vpq_compute_single_core(frames, categories, nframes=nf)
            rows = [[c, stat[c].tp, stat[c].fp, stat[c].fn] for c in sorted(categories)]
            out['clip%d_nf%d_counts' % (ci, nf)] = np.array(rows, dtype=np.int64)
            out['clip%d_nf%d_iou' % (ci, nf)] = np.array([stat[c].iou for c in sorted(categories)], dtype=np.float64)
    out['nclips'] = np.int64(len(specs))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'vpq_cases.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
