"""Tube-matching VPQ statistics with the confusion counts taken on the device.

`vpq_compute_single_core(gt_pred_set, categories, nframes)` has the reference's signature and return value
(`tools/eval_vpq.py:74-209`, a `PQStat`). The reference stacks the id maps of every window of `nframes` frames and calls
`np.unique` on nframes x H x W uint64 keys — for every window and again for every window length (k = 1, 2, 3, 4 in
`eval_vpq.py:main`). Here `FrameCounts` counts every frame ONCE on the device (`vps_pair_count`: dense table over the segment
ids the two JSONs list for that frame) and a window's confusion map is the sum of its frames' sparse tables; the segment
bookkeeping and the matching rules are the reference's statements on the host. No CPU path for the counting."""
import copy
from collections import defaultdict

import numpy as np
import torch

from . import hip

VOID = 0


class PQStatCat:
    """eval_vpq.py:20-32"""

    def __init__(self):
        self.iou, self.tp, self.fp, self.fn = 0.0, 0, 0, 0

    def __iadd__(self, other):
        self.iou += other.iou; self.tp += other.tp; self.fp += other.fp; self.fn += other.fn
        return self


class PQStat:
    """eval_vpq.py:34-71"""

    def __init__(self):
        self.pq_per_cat = defaultdict(PQStatCat)

    def __getitem__(self, i):
        return self.pq_per_cat[i]

    def __iadd__(self, other):
        for label, cat in other.pq_per_cat.items():
            self.pq_per_cat[label] += cat
        return self

    def pq_average(self, categories, isthing, exclude=()):
        """tools/eval_vpq.py:40-71; `exclude`: category ids left out of the average - tools/dataset/viper.py:64-90 is the same
        function with `if label == 11: continue` (VIPER's "mobilebarrier"): `postprocess.DATASETS['viper']['pq_exclude']`"""
        pq, sq, rq, n = 0, 0, 0, 0
        per_class = {}
        for label, info in categories.items():
            if isthing is not None and isthing != (info['isthing'] == 1):
                continue
            if label in exclude:
                continue
            c = self.pq_per_cat[label]
            if c.tp + c.fp + c.fn == 0:
                per_class[label] = {'pq': 0.0, 'sq': 0.0, 'rq': 0.0, 'iou': 0.0, 'tp': 0, 'fp': 0, 'fn': 0}
                continue
            n += 1
            pq_c = c.iou / (c.tp + 0.5 * c.fp + 0.5 * c.fn)
            sq_c = c.iou / c.tp if c.tp != 0 else 0
            rq_c = c.tp / (c.tp + 0.5 * c.fp + 0.5 * c.fn)
            per_class[label] = {'pq': pq_c, 'sq': sq_c, 'rq': rq_c, 'iou': c.iou, 'tp': c.tp, 'fp': c.fp, 'fn': c.fn}
            pq += pq_c; sq += sq_c; rq += rq_c
        return {'pq': pq / n, 'sq': sq / n, 'rq': rq / n, 'n': n}, per_class


def _merged(json_entry):
    segms = {}
    for el in json_entry['segments_info']:                     # eval_vpq.py:93-104
        if el['id'] in segms:
            segms[el['id']]['area'] += el['area']
        else:
            segms[el['id']] = copy.deepcopy(el)
    return segms


class FrameCounts:
    """Per-frame device work, done once per frame whatever the window length: the (gt id, pred id) -> pixels table and the
    per-frame segment tables of eval_vpq.py:89-119 (with its consistency checks between PNG and JSON)."""

    def __init__(self, device='cuda'):
        self.device = torch.device(device)

    def count(self, gt_json, pred_json, gt_pan, pred_pan, categories, extra_gt_ids=None):
        lib = hip.load()
        dev = self.device

        def up(a):
            t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
            t = t.to(dev).contiguous()
            assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3, 'panoptic PNG as uint8 [H,W,3]'
            return t
        g, p = up(gt_pan), up(pred_pan)
        assert g.shape == p.shape
        # ground-truth ids counted in this frame: its own JSON's plus `extra_gt_ids` (the ids any frame of the clip lists): the
        # reference takes np.unique over the WINDOW's pixel pairs (eval_vpq.py:150-157), so an id painted in this frame's PNG but
        # listed only by another frame of the window still takes part in the matching
        gt_ids = np.unique(np.array([VOID] + [el['id'] for el in gt_json['segments_info']] + list(extra_gt_ids or ()), dtype=np.int64))
        pred_ids = np.unique(np.array([VOID] + [el['id'] for el in pred_json['segments_info']], dtype=np.int64))
        gi = torch.from_numpy(gt_ids.astype(np.int32)).to(dev); pi = torch.from_numpy(pred_ids.astype(np.int32)).to(dev)   # ids < 2^24
        counts = torch.empty((len(gt_ids) + 1) * (len(pred_ids) + 1), dtype=torch.int32, device=dev)
        hip.check(lib.vps_pair_count(hip.ptr(g), hip.ptr(p), g.shape[0] * g.shape[1], hip.ptr(gi), len(gt_ids), hip.ptr(pi), len(pred_ids),
                                     hip.ptr(counts), hip.stream_ptr()), 'vps_pair_count')
        tab = counts.view(len(gt_ids) + 1, len(pred_ids) + 1).cpu().numpy().astype(np.int64)
        # ---- eval_vpq.py:105-119: predicted areas from the PNG, PNG <-> JSON consistency
        gt_segms, pred_segms = _merged(gt_json), _merged(pred_json)
        if tab[:, -1].sum() > 0:                               # a predicted id that is neither listed nor VOID: name it like the reference
            ids = (lambda q: q[:, :, 0] + q[:, :, 1] * 256 + q[:, :, 2] * 65536)(p.cpu().numpy().astype(np.int64))
            bad = [int(v) for v in np.unique(ids) if v not in pred_segms and v != VOID]
            raise KeyError('Segment with ID {} is presented in PNG and not presented in JSON.'.format(bad[0]))
        col = tab.sum(0)
        pred_labels_set = set(el['id'] for el in pred_json['segments_info'])
        for j, label in enumerate(pred_ids):
            if col[j] == 0:
                continue                                       # not in the PNG
            label = int(label)
            if label not in pred_segms:
                continue                                       # VOID that the JSON does not list
            pred_segms[label]['area'] = int(col[j])
            pred_labels_set.remove(label)
            if pred_segms[label]['category_id'] not in categories:
                raise KeyError('Segment with ID {} has unknown category_id {}.'.format(label, pred_segms[label]['category_id']))
        if len(pred_labels_set) != 0:
            raise KeyError('The following segment IDs {} are presented in JSON and not presented in PNG.'.format(list(pred_labels_set)))
        # sparse table of the listed ids (unlisted ground-truth ids never take part in the matching)
        pairs = {}
        rows, cols = np.nonzero(tab[:-1, :-1])
        for r, c in zip(rows, cols):
            pairs[(int(gt_ids[r]), int(pred_ids[c]))] = int(tab[r, c])
        return gt_segms, pred_segms, pairs


def vpq_compute_single_core(gt_pred_set, categories, nframes=2, device='cuda', _cache=None):
    """eval_vpq.py:74-209, same arguments and result. `_cache` (optional dict) keeps the per-frame device results across
    calls with different `nframes`; entries are keyed by the clip object as well, so one dict may serve several videos."""
    fc = FrameCounts(device)
    cache = {} if _cache is None else _cache
    stat = PQStat()
    clip_gt_ids = sorted({el['id'] for item in gt_pred_set for el in item[0]['segments_info']})
    for idx in range(0, len(gt_pred_set) - nframes + 1):
        gts, preds, gt_pred_map = [], [], {}
        for off, (gt_json, pred_json, gt_pan, pred_pan, _) in enumerate(gt_pred_set[idx:idx + nframes]):
            # keyed by the clip OBJECT: the entry keeps a reference to it and a hit is accepted only for the same object, so an
            # id() recycled by CPython for another (freed and re-allocated) clip list can never return stale counts
            key = (id(gt_pred_set), idx + off)
            ent = cache.get(key)
            if ent is None or ent[0] is not gt_pred_set:
                ent = cache[key] = (gt_pred_set, fc.count(gt_json, pred_json, gt_pan, pred_pan, categories, clip_gt_ids))
            g, p, pairs = ent[1]
            gts.append(copy.deepcopy(g)); preds.append(copy.deepcopy(p))        # the reference rebuilds them per window
            for k, v in pairs.items():
                gt_pred_map[k] = gt_pred_map.get(k, 0) + v
        vid_gt, vid_pred = {}, {}
        for segms, vid in [(s, vid_gt) for s in gts] + [(s, vid_pred) for s in preds]:      # eval_vpq.py:129-141
            for k in segms.keys():
                if k not in vid:
                    vid[k] = segms[k]
                else:
                    vid[k]['area'] += segms[k]['area']
        gt_pred_map = dict(sorted(gt_pred_map.items()))        # np.unique order: ascending gt id, then pred id
        gt_matched, pred_matched = set(), set()
        for (gt_label, pred_label), intersection in gt_pred_map.items():                    # eval_vpq.py:166-189
            if gt_label not in vid_gt or pred_label not in vid_pred:
                continue
            if vid_gt[gt_label]['iscrowd'] == 1:
                continue
            if vid_gt[gt_label]['category_id'] != vid_pred[pred_label]['category_id']:
                continue
            union = vid_pred[pred_label]['area'] + vid_gt[gt_label]['area'] - intersection - gt_pred_map.get((VOID, pred_label), 0)
            iou = intersection / union
            assert iou <= 1.0, 'INVALID IOU VALUE : %d' % (gt_label)
            if iou > 0.5:
                stat[vid_gt[gt_label]['category_id']].tp += 1
                stat[vid_gt[gt_label]['category_id']].iou += iou
                gt_matched.add(gt_label); pred_matched.add(pred_label)
        crowd = {}
        for gt_label, info in vid_gt.items():                                               # :191-200
            if gt_label in gt_matched:
                continue
            if info['iscrowd'] == 1:
                crowd[info['category_id']] = gt_label
                continue
            stat[info['category_id']].fn += 1
        for pred_label, info in vid_pred.items():                                           # :202-213
            if pred_label in pred_matched:
                continue
            inter = gt_pred_map.get((VOID, pred_label), 0)
            if info['category_id'] in crowd:
                inter += gt_pred_map.get((crowd[info['category_id']], pred_label), 0)
            if inter / info['area'] > 0.5:
                continue
            stat[info['category_id']].fp += 1
    return stat
