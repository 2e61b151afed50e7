"""CPU: pin the oracle's head-logic EDGE paths against the REAL reference functions (VERDICT r2 "Missing" #3).

tests/golden/inject_cases.npz was written by tests/golden/make_inject_golden.py, which runs the reference's own
`MaskROI.forward` (dummy row, ties at the `max_det` cap), the tracking block of `simple_test_bboxes` (undo branch, memory growth),
`MaskRemoval.forward` (overlap rule, keep-nothing) and `SegTerm.forward` (`cls == 0` skip) on the injected head inputs of
tests/inject_cases.py at 1024x2048. Here the oracle's restatements (oracle/fusetrack.py: mask_roi, track_scores, greedy_assign,
mask_removal, seg_term) run on the SAME inputs and must reproduce every integer exactly.

The expensive cases (K = 100: a [1,111,1024,2048] logit tensor per frame on the CPU) run with VPS_SLOW_TESTS=1; the default
set covers every edge path once.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import inject_cases as IC
import vps_amd
from oracle import fusetrack as OF
from vps_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden', 'inject_cases.npz')
FAST = ['K32_M32', 'K32_M100_undo', 'ties_at_cap_M0', 'dummy_row_M32', 'dummy_row_M0', 'keep_nothing', 'overlap_skip']
CASES = list(IC.CASES) if os.environ.get('VPS_SLOW_TESTS') else FAST


@pytest.fixture(scope='module')
def ctx():
    cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'cityscapes', 'fusetrack.py'))
    model = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    sd = synth.synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, 0)
    return sd, IC.neck_features(), np.load(GOLD)


def _canonical(scores, rois, keep, pano, bbox_ids, pan_map, first_frame):
    """Detections with EXACTLY equal scores have no defined order in the reference: `scores.argsort()[::-1]` (gpu_nms.pyx:27) and
    `np.argsort(cls_prob)[::-1]` (mask_removal.py:49) are numpy's unstable sort, whose tie order depends on the numpy build and the
    CPU (the golden's tied block is neither ascending nor descending). Both sides are therefore brought into ONE canonical order
    before they are compared: detections by (-score, x1, y1), kept instances by that rank — positions, first-frame ids (= positions)
    and the instance numbers of the panoptic map are relabelled accordingly. Without ties this is the identity."""
    scores, rois, keep = np.asarray(scores), np.asarray(rois), np.asarray(keep).astype(np.int64)
    perm = np.lexsort((rois[:, 2], rois[:, 1], -scores.astype(np.float64)))          # canonical order of the detections
    rank = np.empty_like(perm); rank[perm] = np.arange(len(perm))
    kc = rank[keep]                                                                   # canonical rank of every kept detection
    ko = np.argsort(kc, kind='stable')                                                # kept list in canonical order
    pano = {k: np.asarray(pano[k])[ko] for k in ('panoptic_cls_inds', 'panoptic_cls_prob', 'panoptic_det_labels', 'panoptic_det_obj_ids')}
    if first_frame and 'panoptic_det_obj_ids' in pano:                                # ids = positions in the first frame
        pano['panoptic_det_obj_ids'] = rank[pano['panoptic_det_obj_ids']]
    inv = np.empty_like(ko); inv[ko] = np.arange(len(ko))
    lut = np.arange(256, dtype=np.uint8); nst = 11
    lut[nst:nst + len(ko)] = (nst + inv).astype(np.uint8)
    return perm, np.sort(kc), pano, lut[np.asarray(pan_map).astype(np.uint8)]


def check_against_golden(g, p, mask_roi, comp_scores, det_obj_ids, keep_inds, pano, bbox_ids, pan_map, sem_map, score_tol=1e-6, comp_tol=2e-3,
                         map_tol=1e-4):
    """shared by the CPU (oracle) and GPU (HIP) tests. mask_roi = (scores, rois, cls_idx) as numpy."""
    gs = g[p + 'mask_roi_scores']
    assert len(mask_roi[0]) == len(gs), 'number of detections %d != %d' % (len(mask_roi[0]), len(gs))
    first = int(g[p + 'M_before']) == 0
    gpano = {k: g[p + k] for k in ('panoptic_cls_inds', 'panoptic_cls_prob', 'panoptic_det_labels', 'panoptic_det_obj_ids')}
    if len(np.unique(gs)) == len(gs):
        perm_x = perm_g = np.arange(len(gs))
        keep_x, keep_g = np.asarray(keep_inds), g[p + 'keep_inds']
        pano_x, pano_g = {k: np.asarray(v) for k, v in pano.items()}, gpano
        pan_x, gs4, ghist = np.asarray(pan_map).astype(np.uint8).reshape(IC.H, IC.W), g[p + 'panoptic_outputs_s4'], g[p + 'panoptic_outputs_hist']
    else:
        assert comp_scores is None and first, 'tied scores are canonicalised for single-frame cases only'
        perm_x, keep_x, pano_x, pan_x = _canonical(mask_roi[0], mask_roi[1], keep_inds, pano, bbox_ids, np.asarray(pan_map).reshape(IC.H, IC.W), first)
        perm_g, keep_g, pano_g, gs4 = _canonical(gs, g[p + 'mask_roi_rois'], g[p + 'keep_inds'], gpano, None, g[p + 'panoptic_outputs_s4'], first)
        ghist = None                                                     # the stride-4 map carries the relabelled comparison
    assert np.array_equal(mask_roi[2][perm_x], g[p + 'mask_roi_cls_idx'][perm_g]), 'MaskROI classes / order'
    assert np.allclose(mask_roi[1][perm_x], g[p + 'mask_roi_rois'][perm_g], rtol=0, atol=1e-3), 'MaskROI boxes'
    assert np.allclose(mask_roi[0][perm_x], gs[perm_g], rtol=1e-5, atol=score_tol), 'MaskROI scores'
    if p + 'comp_scores' in g.files and comp_scores is not None:
        gc = g[p + 'comp_scores']
        assert comp_scores.shape == gc.shape
        assert np.abs(comp_scores - gc).max() <= comp_tol * max(1.0, np.abs(gc).max()), np.abs(comp_scores - gc).max()
    assert np.array_equal(keep_x, keep_g), 'MaskRemoval kept list'
    for k in ('panoptic_cls_inds', 'panoptic_det_labels', 'panoptic_det_obj_ids'):
        assert np.array_equal(pano_x[k], pano_g[k]), k
    assert np.allclose(pano_x['panoptic_cls_prob'], pano_g['panoptic_cls_prob'], rtol=1e-5, atol=score_tol)
    if bbox_ids is not None:
        assert np.array_equal(np.asarray(sorted(bbox_ids)), g[p + 'bbox_ids'])
    HW = float(IC.H * IC.W)
    assert float((pan_x[::4, ::4] != gs4).mean()) < map_tol, 'panoptic_outputs'
    if ghist is not None:
        assert np.abs(np.bincount(pan_x.reshape(-1), minlength=256) - ghist).sum() / HW < 2 * map_tol, 'panoptic_outputs histogram'
    m = np.asarray(sem_map).astype(np.uint8).reshape(IC.H, IC.W)
    assert float((m[::4, ::4] != g[p + 'fcn_outputs_s4']).mean()) < map_tol, 'fcn_outputs'
    assert np.abs(np.bincount(m.reshape(-1), minlength=256) - g[p + 'fcn_outputs_hist']).sum() / HW < 2 * map_tol, 'fcn_outputs histogram'


@pytest.mark.parametrize('case', CASES)
def test_oracle_head_logic_equals_real_reference_functions(ctx, case):
    sd, x, g = ctx
    o = OF.FuseTrackOracle(sd)
    n_undo = 0
    for t, inj in enumerate(IC.frames_of(case)):
        pub = IC.public(inj)
        p = '%s.f%d.' % (case, t)
        with torch.no_grad():
            fcn_output = F.interpolate(pub['fcn_score'], scale_factor=4, mode='bilinear', align_corners=False)
            M_before = 0 if o.prev_bboxes is None else o.prev_bboxes.size(0)
            assert M_before == int(g[p + 'M_before'])
            det = o.detect(x, (IC.H, IC.W), t == 0, pub)
            r = o.panoptic(x, fcn_output, det, pub)
        comp = None if det['comp_scores'] is None else det['comp_scores'].numpy()
        ids = np.asarray(det['det_obj_ids'])
        check_against_golden(g, p, (det['cls_prob'].numpy(), det['det_rois'].numpy(), det['cls_idx'].numpy()), comp, ids, r['keep_inds'],
                             {k: r[k].numpy() for k in ('panoptic_cls_inds', 'panoptic_cls_prob', 'panoptic_det_labels', 'panoptic_det_obj_ids')},
                             [int(i) for i in ids if i >= 0], r['panoptic_outputs'].numpy(), r['fcn_outputs'].numpy())
        # the case exercised its path IN THE REFERENCE RUN (properties of the golden, not of the oracle)
        spec = IC.CASES[case][t]
        K = g[p + 'mask_roi_cls_idx'].shape[0]
        if spec['K'] == 0:
            assert K == 1 and int(g[p + 'mask_roi_cls_idx'][0]) == 0 and float(g[p + 'mask_roi_scores'][0]) == 1.0     # mask_roi.py:136-142
            assert np.array_equal(g[p + 'keep_inds'], [0]) and not bool(g[p + 'mask_energy_nonzero'])
        if spec.get('tie_from') is not None:
            assert K > 100, 'ties at the cap keep more than max_det detections (mask_roi.py:111-116)'
        if spec.get('masks') == 'negative':
            assert np.array_equal(g[p + 'keep_inds'], [0]) and not bool(g[p + 'mask_energy_nonzero'])                    # mask_removal.py:89-91
        if case == 'overlap_skip':
            assert len(g[p + 'keep_inds']) < K
        if p + 'comp_scores' in g.files:
            _, updates = OF.greedy_assign(torch.from_numpy(g[p + 'comp_scores']), M_before)
            seen = set()
            for u in updates:
                if u[0] == 'set':
                    n_undo += u[1] in seen
                    seen.add(u[1])
    if case == 'K32_M100_undo':
        assert n_undo >= 1, 'the golden must contain the tracker undo branch (panoptic_fusetrack.py:449-453)'
