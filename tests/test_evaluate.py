"""VPQ tube statistics. CPU: the oracle restatement against golden statistics of the REAL reference
function (tests/golden/make_vpq_golden.py). GPU: the device-counted evaluator against the golden statistics, exactly."""
import json
import os

import numpy as np
import pytest

from oracle import evaluate as oev

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'vpq_cases.npz')
CATS = {c: {'id': c, 'isthing': 1 if c >= 11 else 0} for c in range(19)}


def _clips():
    z = np.load(GOLD)
    for ci in range(int(z['nclips'])):
        js = json.loads(bytes(z['clip%d_json' % ci]).decode())
        gt, pred = z['clip%d_gt' % ci], z['clip%d_pred' % ci]
        frames = [(js[f][0], js[f][1], gt[f], pred[f], {}) for f in range(len(js))]
        for nf in (1, 2, 3):
            yield ci, nf, frames, z['clip%d_nf%d_counts' % (ci, nf)], z['clip%d_nf%d_iou' % (ci, nf)]


def _check(stat, counts, iou, what):
    for row, v in zip(counts, iou):
        c = int(row[0])
        assert [stat[c].tp, stat[c].fp, stat[c].fn] == [int(row[1]), int(row[2]), int(row[3])], (what, c)
        assert stat[c].iou == float(v), (what, c, stat[c].iou, float(v))          # same sums in the same order: bitwise


def test_oracle_matches_reference_function():
    n = 0
    for ci, nf, frames, counts, iou in _clips():
        _check(oev.vpq_compute_single_core(frames, CATS, nframes=nf), counts, iou, (ci, nf)); n += 1
    assert n == 12 and int(np.load(GOLD)['clip1_nf2_counts'][:, 1].sum()) > 0        # the golden clips do contain matches


def test_host_matching_logic_with_numpy_counts(monkeypatch):
    """CPU coverage of vps_amd/evaluate.py's host side (segment bookkeeping, window sums, matching): the per-frame device count
    is replaced by its NumPy definition and the result must still equal the reference function's statistics"""
    from vps_amd import evaluate as ev

    def count(self, gt_json, pred_json, gt_pan, pred_pan, categories, extra_gt_ids=None):
        ids = lambda q: (lambda u: u[:, :, 0] + u[:, :, 1] * 256 + u[:, :, 2] * 65536)(np.asarray(q).astype(np.int64))
        g, p = ids(gt_pan), ids(pred_pan)
        gt_segms, pred_segms = ev._merged(gt_json), ev._merged(pred_json)
        labels, cnt = np.unique(p, return_counts=True)
        pred_set = set(el['id'] for el in pred_json['segments_info'])
        for label, c in zip(labels, cnt):
            label = int(label)
            if label not in pred_segms:
                assert label == 0
                continue
            pred_segms[label]['area'] = int(c); pred_set.remove(label)
        assert not pred_set
        listed_g = set([0] + [el['id'] for el in gt_json['segments_info']] + list(extra_gt_ids or ()))
        lab, c2 = np.unique(g * (1 << 24) + p, return_counts=True)
        pairs = {(int(l >> 24), int(l & ((1 << 24) - 1))): int(c) for l, c in zip(lab, c2) if int(l >> 24) in listed_g}
        return gt_segms, pred_segms, pairs
    monkeypatch.setattr(ev.FrameCounts, 'count', count)
    monkeypatch.setattr(ev.FrameCounts, '__init__', lambda self, device='cuda': None)
    for ci, nf, frames, counts, iou in _clips():
        _check(ev.vpq_compute_single_core(frames, CATS, nframes=nf, device='cpu'), counts, iou, (ci, nf))


@pytest.mark.gpu
def test_device_counted_vpq_matches_reference_function(dev):
    from vps_amd import evaluate as ev
    for ci, nf, frames, counts, iou in _clips():
        _check(ev.vpq_compute_single_core(frames, CATS, nframes=nf, device=dev), counts, iou, (ci, nf))


@pytest.mark.gpu
def test_device_counts_full_size_and_error_paths(dev):
    import torch
    from vps_amd import evaluate as ev
    rng = np.random.default_rng(0)
    H, W = 1024, 2048
    ids = np.array([0, 7, 300, 70000, 1 << 20, (1 << 24) - 1], dtype=np.int64)
    gt = ids[rng.integers(0, len(ids), size=(H // 32, W // 32))].repeat(32, 0).repeat(32, 1)
    pr = ids[rng.integers(1, len(ids), size=(H // 16, W // 16))].repeat(16, 0).repeat(16, 1)
    rgb = lambda m: np.stack([m % 256, (m // 256) % 256, m // 65536], -1).astype(np.uint8)
    info = lambda m: {'segments_info': [{'id': int(i), 'category_id': int(i % 19), 'iscrowd': 0, 'area': int((m == i).sum())} for i in np.unique(m) if i]}
    g, p, pairs = ev.FrameCounts(dev).count(info(gt), info(pr), torch.from_numpy(rgb(gt)).to(dev), rgb(pr), CATS)
    key = gt.astype(np.uint64) * (1 << 24) + pr.astype(np.uint64)
    lab, cnt = np.unique(key, return_counts=True)
    assert pairs == {(int(l >> 24), int(l & ((1 << 24) - 1))): int(c) for l, c in zip(lab, cnt)}
    assert all(p[int(i)]['area'] == int((pr == i).sum()) for i in np.unique(pr))
    bad = info(pr); bad['segments_info'] = bad['segments_info'][1:]               # a PNG id that the JSON does not list
    with pytest.raises(KeyError):
        ev.FrameCounts(dev).count(info(gt), bad, rgb(gt), rgb(pr), CATS)
    extra = info(pr); extra['segments_info'].append({'id': 12345, 'category_id': 3, 'iscrowd': 0, 'area': 1})
    with pytest.raises(KeyError):
        ev.FrameCounts(dev).count(info(gt), extra, rgb(gt), rgb(pr), CATS)
