"""GPU: SURVEY.md §8(d) config 2 at the BASELINE frame size (1024x2048) — head-side inputs INJECTED at the operator boundaries
(`simple_test(..., inject=...)`: fcn_score, proposals, cls_score, bbox_pred, mask_score), so that the order-defining host
logic of the reference runs on controlled detection sets:

    K in {0 -> MaskROI dummy row, 32, 100, >100 with ties at the max_det cap, >244 -> refusal}, tracker memory M in {0, 32, 100},
    MaskRemoval keeping nothing, SegTerm's cls==0 skip, the tracker's "undo" branch.

Checker: the oracle's head functions (oracle/fusetrack.py: mask_roi / track_scores / greedy_assign / mask_removal / seg_term,
restating mask_roi.py:37-147, panoptic_fusetrack.py:400-469,585-597, mask_removal.py:29-92, unary_logits.py:81-108) fed with
the SAME injected tensors and with the neck features of the HIP run (the convolutional stages are compared elsewhere), so the
CPU cost is seconds per frame. Everything integer must be identical; maps may differ in < 0.1 % of the pixels (bilinear x4 +
argmax at logit ties).
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import vps_amd
from oracle import fusetrack as OF
from vps_amd import hip, nhwc, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W = 1024, 2048
NPROP = 1000


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def make_inject(seed, K, tie_from=None, masks='random', jitter_of=None, extra_same_object=0, extra_offset=0.6):
    """-> dict of injected tensors producing (about) K candidates above the 0.6 score threshold.
    tie_from: candidates tie_from.. share one score (the max_det cap `>=` keeps all of them, mask_roi.py:106-121).
    jitter_of: a previous inject dict — object boxes are that frame's boxes moved by a few pixels (same classes) so the tracker
    has matches; extra_same_object: additional detections of the same class overlapping object 0..n (forces the undo branch)."""
    g = _gen(seed)
    ncell = max(K + extra_same_object, 1)
    gy = max(int(math.sqrt(ncell / 2.0)), 1); gx = (ncell + gy - 1) // gy
    ch, cw = H / gy, W / gx
    size = min(ch, cw) * 0.42
    props = torch.zeros(NPROP, 5)
    # background rows: random boxes, score column only orders them
    cx = torch.rand(NPROP, generator=g) * W; cy = torch.rand(NPROP, generator=g) * H
    s = torch.exp(torch.rand(NPROP, generator=g) * math.log(512 / 16)) * 16
    props[:, 0] = (cx - s / 2).clamp(0, W - 1); props[:, 1] = (cy - s / 2).clamp(0, H - 1)
    props[:, 2] = (cx + s / 2).clamp(0, W - 1); props[:, 3] = (cy + s / 2).clamp(0, H - 1)
    props[:, 4] = torch.rand(NPROP, generator=g)
    cls = torch.randint(1, 9, (NPROP,), generator=g)
    if jitter_of is not None:
        n_old = min(K, jitter_of['_K'])
        props[:n_old, :4] = jitter_of['proposals'][:n_old, :4] + torch.randn(n_old, 4, generator=g) * 2.0
        cls[:n_old] = jitter_of['_cls'][:n_old]
    else:
        n_old = 0
    for i in range(n_old, K + extra_same_object):
        r, c = divmod(i, gx)
        w = size * (0.6 + 0.4 * float(torch.rand(1, generator=g))); h = size * (0.6 + 0.4 * float(torch.rand(1, generator=g)))
        x0 = c * cw + (cw - w) * float(torch.rand(1, generator=g)); y0 = r * ch + (ch - h) * float(torch.rand(1, generator=g))
        props[i, :4] = torch.tensor([x0, y0, x0 + w, y0 + h])
    for j in range(extra_same_object):
        # a second detection of object j's class, half a box away: IoU with it < 0.5 (survives NMS) but it competes for the same
        # memory entry through the label term of compute_comp_scores (track_head.py:73-91)
        i = K + j
        b = props[j, :4].clone(); wj = b[2] - b[0]
        props[i, :4] = torch.stack([b[0] + extra_offset * wj, b[1], b[2] + extra_offset * wj, b[3]])
        cls[i] = cls[j]
    nobj = K + extra_same_object
    props[:, 0::2] = props[:, 0::2].clamp(0, W - 1); props[:, 1::2] = props[:, 1::2].clamp(0, H - 1)
    # scores: distinct probabilities in (0.62, 0.99) for the objects, background rows below the threshold
    p = 0.62 + 0.37 * torch.rand(NPROP, generator=g)
    if tie_from is not None:
        # one class for the tied rows: the softmax of identical logit vectors is bitwise identical, whatever the summation order
        cls[tie_from:nobj] = cls[tie_from]
        p[tie_from:nobj] = 0.75
        p[:tie_from] = 0.80 + 0.19 * torch.rand(tie_from, generator=g)
    cls_score = torch.randn(NPROP, 9, generator=g)
    cls_score[:, 0] += 6.0
    for i in range(nobj):
        cls_score[i] = 0.0
        cls_score[i, int(cls[i])] = math.log(8 * float(p[i]) / (1 - float(p[i])))
    bbox_pred = torch.randn(NPROP, 36, generator=g) * 0.3
    mask_score = torch.randn(320, 9, 28, 28, generator=g) * 2.0
    if masks == 'negative':
        mask_score = -mask_score.abs() - 0.1
    elif masks == 'positive_bias':
        mask_score = mask_score + 2.0
    fcn_score = torch.randn(1, 19, H // 4, W // 4, generator=g)
    return dict(proposals=props, cls_score=cls_score, bbox_pred=bbox_pred, mask_score=mask_score, fcn_score=fcn_score, _K=nobj, _cls=cls)


CASES = {
    # name: [(frame inject kwargs)...]  — frame 1's boxes jitter frame 0's
    'K32_M32': [dict(K=32), dict(K=32, jitter=True)],
    'K100_M100': [dict(K=100), dict(K=100, jitter=True)],
    'K32_M100_undo': [dict(K=100), dict(K=32, jitter=True, extra_same_object=12)],
    'K100_M32': [dict(K=32), dict(K=100, jitter=True)],
    'ties_at_cap_M0': [dict(K=130, tie_from=90)],
    'dummy_row_M32': [dict(K=32), dict(K=0)],
    'dummy_row_M0': [dict(K=0)],
    'keep_nothing': [dict(K=32, masks='negative')],
    # same-class boxes with IoU ~0.43 (survive NMS 0.5) and mostly-positive masks: the later one has > 30 % of its mask already
    # claimed and is dropped by MaskRemoval (mask_removal.py:75-80)
    'overlap_skip': [dict(K=40, extra_same_object=20, extra_offset=0.4, masks='positive_bias')],
}


@pytest.fixture(scope='module')
def model(dev):
    old = nhwc.DEFAULT_PREC
    nhwc.DEFAULT_PREC = hip.PREC_BF16X6                    # the benchmarked arithmetic
    try:
        cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'cityscapes', 'fusetrack.py'))
        m = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
        sd = synth.load_synth(m, 0)
        m.ensure_packed(dev)
    finally:
        nhwc.DEFAULT_PREC = old
    frames = [f.to(dev) for f in synth.synth_clip(H, W, 2, 0)]
    return m, sd, frames


def _public(inj):
    return {k: v for k, v in inj.items() if not k.startswith('_')}


@pytest.mark.parametrize('case', list(CASES))
def test_injected_heads_match_oracle_at_full_size(dev, model, case):
    m, sd, frames = model
    m._cache = None; m.reset_tracker()
    o = OF.FuseTrackOracle(sd)
    prev = None
    n_undo = 0
    for t, spec in enumerate(CASES[case]):
        inj = make_inject(100 * t + 7, spec['K'], spec.get('tie_from'), spec.get('masks', 'random'),
                          prev if spec.get('jitter') else None, spec.get('extra_same_object', 0), spec.get('extra_offset', 0.6))
        prev = inj
        out = m.simple_test(frames[t], [synth.img_meta(H, W, 10000 + t + 1)], ref_img=[frames[t - 1 if t else 0]], inject=_public(inj))
        torch.cuda.synchronize()
        x = [l.to_nchw().cpu() for l in m._aux['neck_out']]
        with torch.no_grad():
            fcn_output = F.interpolate(inj['fcn_score'], scale_factor=4, mode='bilinear', align_corners=False)
            M_before = 0 if o.prev_bboxes is None else o.prev_bboxes.size(0)
            det = o.detect(x, (H, W), t == 0, _public(inj))
            if det['comp_scores'] is not None:
                _, updates = OF.greedy_assign(det['comp_scores'], M_before)
                seen = set()
                for u in updates:
                    if u[0] == 'set':
                        n_undo += u[1] in seen
                        seen.add(u[1])
            ref = o.panoptic(x, fcn_output, det, _public(inj))
        hd = m._aux['det']
        K = det['cls_idx'].numel()
        print('%s frame %d: K=%d (HIP %d) M=%d kept %d' % (case, t, K, hd['cls_idx'].numel(), M_before, len(ref['keep_inds'])))
        # ---- MaskROI: same detections in the same order
        assert np.array_equal(hd['cls_idx'].cpu().numpy(), det['cls_idx'].numpy())
        assert np.allclose(hd['det_rois'].cpu().numpy(), det['det_rois'].numpy(), rtol=0, atol=1e-3)
        assert np.allclose(hd['cls_prob'].cpu().numpy(), det['cls_prob'].numpy(), rtol=1e-5, atol=1e-6)
        # ---- tracker: identical ids (greedy assignment incl. undo), MaskRemoval: identical kept list
        assert np.array_equal(np.asarray(hd['det_obj_ids']), np.asarray(det['det_obj_ids'])), (hd['det_obj_ids'], det['det_obj_ids'])
        assert np.array_equal(np.asarray(m._aux['keep_inds']), np.asarray(ref['keep_inds']))
        r = {k: v.cpu().numpy() for k, v in out[2].items()}
        for key in ('panoptic_cls_inds', 'panoptic_det_labels', 'panoptic_det_obj_ids'):
            assert np.array_equal(r[key], ref[key].numpy()), key
        assert sorted(int(k) for k in out[0].keys()) == sorted(int(i) for i in det['det_obj_ids'] if i >= 0)
        dpan = float((r['panoptic_outputs'] != ref['panoptic_outputs'].numpy().astype(np.uint8)).mean())
        dsem = float((r['fcn_outputs'] != ref['fcn_outputs'].numpy().astype(np.uint8)).mean())
        print('   maps: pan mismatch %.5f%%, sem mismatch %.5f%%' % (100 * dpan, 100 * dsem))
        assert dpan < 1e-3 and dsem < 1e-3
        # ---- the case really exercised its path
        if spec['K'] == 0:
            assert K == 1 and int(det['cls_idx'][0]) == 0 and float(det['cls_prob'][0]) == 1.0       # mask_roi.py:136-142 dummy row
            assert np.array_equal(ref['keep_inds'], [0])
        if spec.get('tie_from') is not None:
            assert K > 100, 'ties at the cap must keep more than max_det detections (mask_roi.py:111-116)'
        if spec.get('masks') == 'negative':
            assert np.array_equal(ref['keep_inds'], [0]) and not m._aux.get('masks_valid', True)         # mask_removal.py:89-91
        if case == 'overlap_skip':
            assert len(ref['keep_inds']) < K, 'the case must exercise the MaskRemoval overlap rule'
    if case == 'K32_M100_undo':
        assert n_undo >= 1, 'the case must exercise the tracker undo branch (panoptic_fusetrack.py:449-453)'
        print('   tracker undo events: %d' % n_undo)


def test_more_than_244_instances_is_refused_loudly(dev, model):
    """the uint8 panoptic map names at most 244 instances (11 + k <= 255, 255 = VOID, test_vpq.py:51-56); the reference
    silently wraps, vps_panoptic_combine refuses"""
    m, sd, frames = model
    m._cache = None; m.reset_tracker()
    inj = make_inject(5, 300, tie_from=10)
    o = OF.FuseTrackOracle(sd)
    x = [torch.zeros(1, 256, H // s, W // s) for s in (4, 8, 16, 32, 64)]
    with torch.no_grad():
        det = o.detect(x, (H, W), True, _public(inj))
    assert det['cls_idx'].numel() > 244, 'the case must produce more than 244 detections (got %d)' % det['cls_idx'].numel()
    with pytest.raises(hip.VpsHipError):
        m.simple_test(frames[0], [synth.img_meta(H, W, 10001)], ref_img=[frames[0]], inject=_public(inj))
        torch.cuda.synchronize()
    m._cache = None; m.reset_tracker()
