"""Shared comparison of one frame's outputs with a golden file of the REAL reference detector (tests/golden/make_golden.py).
Used by the GPU test of the second-seed clip (HIP path) and, on the CPU, with the oracle's outputs and with deliberately
perturbed copies — so that the tolerant matching itself is tested where no GPU is needed.

Tolerances: stage tensors within `tol` of max|ref|; semantic map < 0.1 % differing pixels; every kept detection of
the golden frame found with the same class and a score within 2e-3 (at most `max_unmatched` per frame on either side: the
synthetic heads put up to 100 detections into a narrow score band, a borderline one may flip); track ids of the matched detections
equal up to ONE relabelling over the clip (`id_map` / `id_back` are carried from frame to frame by the caller); the panoptic map is
compared pixel by pixel when the listing is identical, and as a map of (stuff class | instance class + relabelled track id)
when it is not (instance numbers are listing positions)."""
import numpy as np

NSTUFF = 11      # panoptic ids below are stuff classes, NSTUFF + j = j-th listed instance


def relmax(got, ref):
    got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got - ref).max() / max(float(np.abs(ref).max()), 1e-12))


def compare_frame(rec, g, p, id_map, id_back, tol=2e-3, max_unmatched=1, pan_tol=1e-3, max_id_violations=0):
    """rec: numpy arrays of one frame — 'fcn_outputs', 'panoptic_outputs', 'panoptic_cls_inds', 'panoptic_cls_prob',
    'panoptic_det_obj_ids' and optionally stage tensors under the golden file's names ('flow_full', 'fpn_p2', 'fpn_p5',
    'neck_out_p2', 'fcn_score'). g: the loaded .npz, p: 'f%d.' prefix. Returns a report dict; raises AssertionError."""
    rep = {}
    for k in ('flow_full', 'fpn_p2', 'fpn_p5', 'neck_out_p2', 'fcn_score'):
        if k in rec:
            rep[k] = relmax(rec[k], g[p + k])
            assert rep[k] < tol, (k, rep[k])
    rep['sem_mismatch'] = float((np.asarray(rec['fcn_outputs']).reshape(-1) != g[p + 'fcn_outputs'].reshape(-1)).mean())
    assert rep['sem_mismatch'] < 1e-3, rep
    gc, gp, gid = g[p + 'panoptic_cls_inds'], g[p + 'panoptic_cls_prob'], g[p + 'panoptic_det_obj_ids']
    oc, op, oid = (np.asarray(rec[k]) for k in ('panoptic_cls_inds', 'panoptic_cls_prob', 'panoptic_det_obj_ids'))
    rep['strict'] = bool(np.array_equal(oc, gc) and np.array_equal(oid, gid))
    used, unmatched = set(), 0
    for i in range(len(oc)):
        d = np.abs(gp - op[i]) + 1e6 * (gc != oc[i])
        for j in used:
            d[j] = 1e9
        j = int(np.argmin(d)) if len(d) else -1
        if j < 0 or d[j] >= 2e-3:
            unmatched += 1
            continue
        used.add(j)
        a, b = int(oid[i]), int(gid[j])
        if not (id_map.setdefault(a, b) == b and id_back.setdefault(b, a) == a):
            # two same-class detections whose scores are closer than the matching tolerance may be paired crosswise
            id_map['violations'] = id_map.get('violations', 0) + 1
            assert id_map['violations'] <= max_id_violations, ('track ids are not one relabelling of the golden ids', p, i, a, b)
    rep['unmatched'] = (unmatched, len(gc) - len(used))
    assert unmatched <= max_unmatched and len(gc) - len(used) <= max_unmatched, rep
    if rep['strict']:
        rep['pan_mismatch'] = float((np.asarray(rec['panoptic_outputs']).reshape(-1) != g[p + 'panoptic_outputs'].reshape(-1)).mean())
        assert rep['pan_mismatch'] < pan_tol, rep
    else:
        # the listing differs (instance numbers are listing positions): compare the maps as maps of (stuff class | instance class,
        # track id through the clip's relabelling). An unmatched detection occupies pixels the other side gives to something else.
        def labels(pan, cls, ids, relabel):
            lut = np.arange(256, dtype=np.int64)
            for j in range(len(cls)):
                i = int(ids[j])
                lut[NSTUFF + j] = 1000 + 100000 * int(cls[j]) + (relabel.get(i, 50000 + i) if relabel is not None else i)
            return lut[np.asarray(pan).astype(np.int64).reshape(-1)]
        mine = labels(rec['panoptic_outputs'], oc, oid, {k: v for k, v in id_map.items() if k != 'violations'})
        gold = labels(g[p + 'panoptic_outputs'], gc, gid, None)
        rep['pan_mismatch'] = float((mine != gold).mean())
        assert rep['pan_mismatch'] < (pan_tol if rep['unmatched'] == (0, 0) else 2e-2), rep
    return rep
