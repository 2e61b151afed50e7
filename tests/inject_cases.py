"""Injected head inputs of SURVEY.md 8(d) config 2 at 1024x2048 (shared by tests/golden/make_inject_golden.py — which runs
the REAL reference head code on them — tests/test_inject_golden.py and tests/test_config2_inject_gpu.py).

`make_inject` builds the tensors the detector's heads would produce (proposals, cls_score, bbox_pred, mask_score, fcn_score)
so that the order-defining host logic runs on controlled detection sets; `neck_features` is the synthetic neck output the
RoI extractors read (track embeddings). Everything is drawn from seeded CPU generators: identical on every machine.
"""
import math

import torch

H, W = 1024, 2048
NPROP = 1000


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def make_inject(seed, K, tie_from=None, masks='random', jitter_of=None, extra_same_object=0, extra_offset=0.6, extra_first=False):
    """-> dict of injected tensors producing (about) K candidates above the 0.6 score threshold.
    tie_from: candidates tie_from.. share one score (the max_det cap `>=` keeps all of them, mask_roi.py:106-121).
    jitter_of: a previous inject dict — object boxes are that frame's boxes moved by a few pixels (same classes) so the tracker
    has matches; extra_same_object: additional detections of the same class overlapping object 0..n; extra_first: each of them scores
    0.05 above its object (listed first: forces the undo branch)."""
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
    if extra_first:
        # every extra detection is listed BEFORE the object it competes with (higher class score, weaker match): it takes the memory
        # entry first and is undone when the object itself comes (panoptic_fusetrack.py:446-453)
        for j in range(extra_same_object):
            p[j] = 0.62 + 0.28 * float(p[j] - 0.62) / 0.37
            p[K + j] = p[j] + 0.05
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
    if tie_from is not None:
        # the mask bank is indexed by LIST POSITION and the order of exactly tied detections is undefined in the reference
        # (numpy's unstable argsort): every position a tied detection can land on holds the same mask
        mask_score[tie_from:nobj] = mask_score[tie_from].clone()
    fcn_score = torch.randn(1, 19, H // 4, W // 4, generator=g)
    return dict(proposals=props, cls_score=cls_score, bbox_pred=bbox_pred, mask_score=mask_score, fcn_score=fcn_score, _K=nobj, _cls=cls)


CASES = {
    # name: [(frame inject kwargs)...]  — frame 1's boxes jitter frame 0's
    'K32_M32': [dict(K=32), dict(K=32, jitter=True)],
    'K100_M100': [dict(K=100), dict(K=100, jitter=True)],
    'K32_M100_undo': [dict(K=100), dict(K=32, jitter=True, extra_same_object=12, extra_first=True)],
    'K100_M32': [dict(K=32), dict(K=100, jitter=True)],
    'ties_at_cap_M0': [dict(K=130, tie_from=90)],
    'dummy_row_M32': [dict(K=32), dict(K=0)],
    'dummy_row_M0': [dict(K=0)],
    'keep_nothing': [dict(K=32, masks='negative')],
    # same-class boxes with IoU ~0.43 (survive NMS 0.5) and mostly-positive masks: the later one has > 30 % of its mask already
    # claimed and is dropped by MaskRemoval (mask_removal.py:75-80)
    'overlap_skip': [dict(K=40, extra_same_object=20, extra_offset=0.4, masks='positive_bias')],
}


def neck_features(seed=11):
    """five FPN-neck levels [1,256,H/s,W/s], s = 4..64: the tensors the box / track / mask RoI extractors read. SMOOTH fields
    (noise drawn at 1/8 of each level's resolution, bilinearly upsampled) plus a little white noise: RoIs that overlap see
    correlated features, distant ones independent features — like a real feature pyramid, and what makes a second detection of
    an object compete for that object's memory entry (the tracker's undo case)."""
    import torch.nn.functional as F
    g = _gen(seed)
    out = []
    for s_ in (4, 8, 16, 32, 64):
        h, w = H // s_, W // s_
        base = torch.randn(1, 256, max(h // 8, 1), max(w // 8, 1), generator=g)
        lvl = F.interpolate(base, size=(h, w), mode='bilinear', align_corners=False) + 0.1 * torch.randn(1, 256, h, w, generator=g)
        out.append(lvl.contiguous())
    return out


def frames_of(case):
    """-> [inject dict per frame] of a CASES entry (frame t+1 jitters frame t where the spec says so)"""
    out, prev = [], None
    for t, spec in enumerate(CASES[case]):
        inj = make_inject(100 * t + 7, spec['K'], spec.get('tie_from'), spec.get('masks', 'random'),
                          prev if spec.get('jitter') else None, spec.get('extra_same_object', 0), spec.get('extra_offset', 0.6),
                          spec.get('extra_first', False))
        prev = inj
        out.append(inj)
    return out


def public(inj):
    return {k: v for k, v in inj.items() if not k.startswith('_')}
