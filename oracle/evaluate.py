"""CPU restatement (NumPy) of the reference's tube-matching VPQ statistics — TEST INFRASTRUCTURE ONLY.

Follows /root/reference/tools/eval_vpq.py:20-71 (PQStatCat / PQStat) and :74-209 (vpq_compute_single_core). Pinned against the
real function: tests/golden/make_vpq_golden.py imports tools/eval_vpq.py (numpy + PIL only) and stores its per-category
statistics for synthetic clips in tests/golden/vpq_cases.npz; tests/test_evaluate.py checks this restatement against them
exactly (integers equal, iou sums bitwise)."""
import copy
from collections import defaultdict

import numpy as np

OFFSET = 256 * 256 * 256
VOID = 0


class PQStatCat:
    def __init__(self):
        self.iou, self.tp, self.fp, self.fn = 0.0, 0, 0, 0

    def __iadd__(self, o):
        self.iou += o.iou; self.tp += o.tp; self.fp += o.fp; self.fn += o.fn
        return self


class PQStat:
    def __init__(self):
        self.pq_per_cat = defaultdict(PQStatCat)

    def __getitem__(self, i):
        return self.pq_per_cat[i]

    def __iadd__(self, o):
        for label, c in o.pq_per_cat.items():
            self.pq_per_cat[label] += c
        return self


def ids_of(pan_rgb):
    p = np.uint32(pan_rgb)
    return p[:, :, 0] + p[:, :, 1] * 256 + p[:, :, 2] * 256 * 256           # eval_vpq.py:91-92


def merged_segments(json_entry):
    """eval_vpq.py:93-104: segments_info -> {id: copy}, areas of repeated ids summed"""
    segms = {}
    for el in json_entry['segments_info']:
        if el['id'] in segms:
            segms[el['id']]['area'] += el['area']
        else:
            segms[el['id']] = copy.deepcopy(el)
    return segms


def frame_segments(gt_json, pred_json, pan_pred, categories):
    """eval_vpq.py:93-119: per-frame segment tables; predicted areas are re-counted from the PNG and checked against the JSON"""
    gt_segms, pred_segms = merged_segments(gt_json), merged_segments(pred_json)
    pred_labels_set = set(el['id'] for el in pred_json['segments_info'])
    labels, labels_cnt = np.unique(pan_pred, return_counts=True)
    for label, label_cnt in zip(labels, labels_cnt):
        if label not in pred_segms:
            if label == VOID:
                continue
            raise KeyError('Segment with ID {} is presented in PNG and not presented in JSON.'.format(label))
        pred_segms[label]['area'] = label_cnt
        pred_labels_set.remove(label)
        if pred_segms[label]['category_id'] not in categories:
            raise KeyError('Segment with ID {} has unknown category_id {}.'.format(label, pred_segms[label]['category_id']))
    if len(pred_labels_set) != 0:
        raise KeyError('The following segment IDs {} are presented in JSON and not presented in PNG.'.format(list(pred_labels_set)))
    return gt_segms, pred_segms


def tube_segments(per_frame):
    """eval_vpq.py:129-141: first occurrence kept (the dict object itself), areas of later frames added"""
    vid = {}
    for segms in per_frame:
        for k in segms.keys():
            if k not in vid:
                vid[k] = segms[k]
            else:
                vid[k]['area'] += segms[k]['area']
    return vid


def match_tubes(stat, gt_pred_map, vid_gt_segms, vid_pred_segms):
    """eval_vpq.py:158-207 for one window; gt_pred_map must iterate in ascending (gt id, pred id) order like np.unique"""
    gt_matched, pred_matched = set(), set()
    for (gt_label, pred_label), intersection in gt_pred_map.items():
        if gt_label not in vid_gt_segms or pred_label not in vid_pred_segms:
            continue
        if vid_gt_segms[gt_label]['iscrowd'] == 1:
            continue
        if vid_gt_segms[gt_label]['category_id'] != vid_pred_segms[pred_label]['category_id']:
            continue
        union = vid_pred_segms[pred_label]['area'] + vid_gt_segms[gt_label]['area'] - intersection - gt_pred_map.get((VOID, pred_label), 0)
        iou = intersection / union
        assert iou <= 1.0, 'INVALID IOU VALUE : %d' % (gt_label)
        if iou > 0.5:
            stat[vid_gt_segms[gt_label]['category_id']].tp += 1
            stat[vid_gt_segms[gt_label]['category_id']].iou += iou
            gt_matched.add(gt_label)
            pred_matched.add(pred_label)
    crowd_labels_dict = {}
    for gt_label, gt_info in vid_gt_segms.items():
        if gt_label in gt_matched:
            continue
        if gt_info['iscrowd'] == 1:
            crowd_labels_dict[gt_info['category_id']] = gt_label
            continue
        stat[gt_info['category_id']].fn += 1
    for pred_label, pred_info in vid_pred_segms.items():
        if pred_label in pred_matched:
            continue
        intersection = gt_pred_map.get((VOID, pred_label), 0)
        if pred_info['category_id'] in crowd_labels_dict:
            intersection += gt_pred_map.get((crowd_labels_dict[pred_info['category_id']], pred_label), 0)
        if intersection / pred_info['area'] > 0.5:
            continue
        stat[pred_info['category_id']].fp += 1


def vpq_compute_single_core(gt_pred_set, categories, nframes=2):
    """eval_vpq.py:74-209. gt_pred_set: list of (gt_json, pred_json, gt_pan, pred_pan, gt_image_json) per frame."""
    stat = PQStat()
    for idx in range(0, len(gt_pred_set) - nframes + 1):
        pans_gt, pans_pred, gts, preds = [], [], [], []
        for gt_json, pred_json, gt_pan, pred_pan, _ in gt_pred_set[idx:idx + nframes]:
            pan_gt, pan_pred = ids_of(gt_pan), ids_of(pred_pan)
            g, p = frame_segments(gt_json, pred_json, pan_pred, categories)
            pans_gt.append(pan_gt); pans_pred.append(pan_pred); gts.append(g); preds.append(p)
        vid_gt, vid_pred = tube_segments(gts), tube_segments(preds)
        key = np.stack(pans_gt).astype(np.uint64) * OFFSET + np.stack(pans_pred).astype(np.uint64)      # :150
        labels, labels_cnt = np.unique(key, return_counts=True)
        gt_pred_map = {(label // OFFSET, label % OFFSET): inter for label, inter in zip(labels, labels_cnt)}
        match_tubes(stat, gt_pred_map, vid_gt, vid_pred)
    return stat
