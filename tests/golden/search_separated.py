"""Find synthetic HEAD weights whose every listing decision at 1024x2048 has a margin above the arithmetic noise
(VERDICT r2 "Next round" #1a). Test infrastructure; runs on the CPU with the oracle only (no reference, no GPU).

The backbone / FlowNet2 / neck / RPN / box-regression / track / mask weights stay `synth.synth_state_dict(seed 0)`. Only the
classification layer of the box head is replaced — `bbox_head.fc_cls.{weight,bias}` — because the listing
(`panoptic_cls_inds`, the score order that defines the first frame's ids, the 0.6 score threshold of MaskROI, the NMS winner of
every cluster) is a function of its scores. With random weights the ~8000 candidate scores of a frame are dense around every
threshold, so the layer is FITTED instead: 8..12 (RoI, class) pairs with mutually disjoint refined boxes are chosen in frame 0
and followed through the clip, and a weighted ridge regression on the shared-FC features of all frames maps them to logits of
evenly spaced probabilities in (0.64, 0.97) and everything else far below the threshold (RoIs overlapping an object in its
class with extra weight). The fitted layer is an ordinary fc_cls tensor pair, stored in tests/golden/separated_fc_cls.npz.

A candidate is accepted when, on a 4-frame clip,
  (a) 4..40 detections per frame (no `max_det` cap in play),
  (b) every candidate probability is >= MARGIN_P away from the 0.6 threshold, every pair of adjacent kept scores is >= MARGIN_P
      apart, every IoU that the class-agnostic NMS compares with 0.5 between two above-threshold candidates is >= MARGIN_IOU
      away from it,
  (c) and the COMPLETE outputs (classes, labels, track ids over the clip, kept list; panoptic map < 3 %) are invariant under
      NPERT random perturbations of the neck output and of the semantic logits by N(0, (SIGMA * max|x|)^2) — max-norm error
      ~5.5 * SIGMA * max|x| = 2.2e-3 max|x|: 4x the largest error the HIP path's neck output was measured at
      (5.4e-4 * max|ref| at 1024x2048, profiles/r02_fullsize_golden_report.txt); the logits move by ~3e-2 under it.

    python tests/golden/search_separated.py            # ~10 min of CPU; writes tests/golden/separated_fc_cls.npz
"""
import math
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import vps_amd                                     # noqa: E402
from oracle import fusetrack as OF                 # noqa: E402
from oracle import ops as OO                       # noqa: E402
from vps_amd import synth                          # noqa: E402

H, W, NFRAMES, SEED = 1024, 2048, 4, 0
# BASELINE config 5 (round 4): VPS_SEP_CONFIG5=1 -> the ResNet-101 model on 3 frames at 1088x1920; the FIRST trial that passes the
# margin filter is written to tests/golden/config5_fc_cls.npz (oracle margins only, no GPU in the selection)
CONFIG5 = bool(os.environ.get('VPS_SEP_CONFIG5'))
RPN_KEYS, RPN_SCALE = ('rpn_head.rpn_cls.weight', 'rpn_head.rpn_cls.bias'), 0.4
DEPTH = 50
if CONFIG5:
    H, W, NFRAMES, DEPTH = 1088, 1920, 3, 101
SIGMA = 4.0e-4
NPERT = 4
MARGIN_P, MARGIN_IOU = 1.0e-2, 2.0e-2
CACHE = os.environ.get('VPS_SEP_CACHE', '/tmp/vps_sep_cache%s.pt' % ('_c5' if CONFIG5 else ''))


def build_sd():
    cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'viper', 'fusetrack_r101.py') if CONFIG5 else os.path.join(ROOT, 'configs', 'cityscapes', 'fusetrack.py'))
    model = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    sd = synth.synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, SEED)
    if CONFIG5:
        # at 1088x1920 the 101-layer synthetic network drives the objectness logits to 20: 250 .. 290 of the 1000 proposals of a frame
        # come out with a score of exactly 1.0f, and which of them survive top-k and NMS is then decided by how torch orders TIED scores
        # (unspecified; the first version of this fixture lost a detection to it in the exact-fp32 kernels as in the split modes).
        # The objectness layer is rescaled so that the scores are distinct (largest logit ~8); it travels with the fitted layer.
        for k in RPN_KEYS:
            sd[k] = sd[k] * RPN_SCALE
    return sd


def stage_cache(sd):
    """neck output + semantic logits of every frame (the expensive, head-independent part), cached on disk"""
    if os.path.exists(CACHE):
        return torch.load(CACHE)
    o = OF.FuseTrackOracle(sd, depth=DEPTH)
    frames = synth.synth_clip(H, W, NFRAMES, SEED)
    out, prev = [], None
    with torch.no_grad():
        for t in range(NFRAMES):
            t0 = time.time()
            pre = o.extract_feat(frames[t])
            flow = o.compute_flow(frames[t], frames[t - 1] if t else frames[0], 0.25)
            x = OF.bfp_tcea(o.sd, 'extra_neck.', pre, prev if t else pre, flow)
            _, fcn_score = OF.upsnet_fpn(o.sd, 'panopticFPN.', x[0:4])
            out.append(dict(x=[l.clone() for l in x], fcn_score=fcn_score.clone()))
            prev = pre
            print('frame %d staged in %.0f s' % (t, time.time() - t0), flush=True)
    torch.save(out, CACHE)
    return out


def perturbed(stage, p):
    """p = 0: the tensors themselves; p > 0: + N(0, (SIGMA max|.|)^2), seeded by (frame-independent) p"""
    if p == 0:
        return stage['x'], stage['fcn_score']
    g = torch.Generator().manual_seed(1000 + p)
    x = [l + torch.randn(l.shape, generator=g) * (SIGMA * float(l.abs().max())) for l in stage['x']]
    s = stage['fcn_score']
    return x, s + torch.randn(s.shape, generator=g) * (SIGMA * float(s.abs().max()))


def head_inputs(sd, x):
    """the fc_cls-independent part of the box branch: proposals, rois, shared-FC features, bbox_pred"""
    proposals = OF.rpn_get_bboxes(OF.rpn_forward(sd, 'rpn_head.', x), (H, W))
    rois = torch.cat([proposals.new_zeros(proposals.size(0), 1), proposals[:, :4]], dim=-1)
    f = OF.roi_extract(x, rois, 7).view(rois.size(0), -1)
    for i in range(2):
        f = F.relu(F.linear(f, sd['bbox_head.shared_fcs.%d.weight' % i], sd['bbox_head.shared_fcs.%d.bias' % i]))
    bbox_pred = F.linear(f, sd['bbox_head.fc_reg.weight'], sd['bbox_head.fc_reg.bias'])
    return dict(proposals=proposals, rois=rois, feat=f, bbox_pred=bbox_pred)


def margins(rois, bbox_pred, cls_prob):
    """margins of the MaskROI decisions of one frame (mask_roi.py:88-121): -> dict(K, thr, gap, iou)"""
    im_info = np.array([[float(H), float(W), 1.0]])
    prob = cls_prob[:, 1:].reshape(-1).numpy()
    thr = float(np.abs(prob - 0.6).min())
    boxes = OF._clip_boxes_np(OF._bbox_transform_np(rois.numpy()[:, 1:], bbox_pred.numpy(), OF.CFG['bbox_reg_weights']), im_info[0, :2])
    boxes = boxes.reshape((rois.shape[0], -1, 4))[:, 1:, :].reshape((-1, 4))
    inds = np.where(prob > 0.6)[0]
    order = inds[np.argsort(-prob[inds], kind='stable')]
    b = torch.from_numpy(boxes[order]).float()
    iou = OF.bbox_overlaps(b, b).numpy()
    m = len(order)
    alive = np.ones(m, bool); keep = []
    iou_margin = 1.0
    for i in range(m):
        if not alive[i]:
            continue
        keep.append(i)
        for j in range(i + 1, m):
            if alive[j]:
                iou_margin = min(iou_margin, abs(iou[i, j] - 0.5))       # every comparison the greedy pass makes
                if iou[i, j] > 0.5:
                    alive[j] = False
    ks = prob[order][keep]
    gap = float(np.min(-np.diff(ks))) if len(ks) > 1 else 1.0
    # a suppressed candidate may also overtake its suppressor: the gap between a keeper and the best candidate it suppresses
    return dict(K=len(keep), thr=thr, gap=gap, iou=float(iou_margin))


def refined_boxes(h):
    """class-specific refined, clipped boxes [n, 8, 4] (mask_roi.py:50-55)"""
    im = np.array([float(H), float(W)])
    b = OF._clip_boxes_np(OF._bbox_transform_np(h['rois'].numpy()[:, 1:], h['bbox_pred'].numpy(), OF.CFG['bbox_reg_weights']), im)
    return torch.from_numpy(b.reshape((h['rois'].shape[0], -1, 4))[:, 1:, :].copy()).float()


def choose_objects(hin0, nobj, rng, top=400):
    """frame 0: `nobj` (roi, class) pairs whose refined boxes are mutually disjoint (IoU < 0.1) and >= 24 px; then the roi of every
    later frame whose refined box of that class overlaps the (shifted) object most (>= 0.6, else the object is absent there)"""
    boxes = [refined_boxes(h) for h in hin0]
    n = boxes[0].shape[0]
    picks = []
    for i in rng.permutation(min(n, top)):
        c = int(rng.integers(0, 8))
        b = boxes[0][i, c]
        if min(float(b[2] - b[0]), float(b[3] - b[1])) < 24:
            continue
        if all(float(OF.bbox_overlaps(b[None], boxes[0][j, cj][None])) < 0.1 for j, cj in picks):
            picks.append((int(i), c))
        if len(picks) == nobj:
            break
    objs = []
    for i, c in picks:
        track = [i]
        b = boxes[0][i, c]
        for t in range(1, NFRAMES):
            shifted = b - torch.tensor([2.0 * t, 1.0 * t, 2.0 * t, 1.0 * t])      # synth_clip: frame t = base translated by t*(2,1)
            iou = OF.bbox_overlaps(shifted[None], boxes[t][:, c])[0]
            j = int(iou.argmax())
            track.append(j if float(iou[j]) >= 0.6 else -1)
        objs.append((c, track))
    return objs, boxes


def fit_head(hin0, objs, boxes, probs, lam=3e-3, w_obj=1000.0, w_near=30.0):
    """weighted ridge regression of the 9 logits on the shared-FC features of all frames: objects -> logit(p), rows whose refined
    box of an object's class overlaps it (IoU >= 0.15) -> strongly negative, everything else negative; background logit 0"""
    Fm = torch.cat([h['feat'] for h in hin0], 0).double()
    mu, sd = Fm.mean(0), Fm.std(0) + 1e-6
    X = torch.cat([(Fm - mu) / sd, torch.ones(Fm.shape[0], 1, dtype=torch.float64)], 1)
    n = hin0[0]['feat'].shape[0]
    Y = torch.full((Fm.shape[0], 9), -7.0, dtype=torch.float64); Y[:, 0] = 0.0
    Wt = torch.ones(Fm.shape[0], 9, dtype=torch.float64)
    for (c, track), p in zip(objs, probs):
        for t, j in enumerate(track):
            if j < 0:
                continue
            iou = OF.bbox_overlaps(boxes[t][j, c][None], boxes[t][:, c])[0]
            near = (iou >= 0.15).nonzero()[:, 0]
            Wt[t * n + near, c + 1] = w_near
            Y[t * n + j, c + 1] = math.log(p / (1 - p)); Wt[t * n + j, :] = w_obj
    W = torch.zeros(9, X.shape[1], dtype=torch.float64)
    reg = lam * torch.eye(X.shape[1], dtype=torch.float64) * X.shape[0]; reg[-1, -1] = 0
    obj_rows = [(t * n + j, c + 1, math.log(p / (1 - p))) for (c, track), p in zip(objs, probs) for t, j in enumerate(track) if j >= 0]
    lhs = [None] * 9
    for c in range(1, 9):
        lhs[c] = torch.linalg.inv(X.t() @ (X * Wt[:, c:c + 1]) + reg)
    for it in range(4):
        # the ridge fit misses the object targets by a few hundredths: move the targets by the miss and fit again
        for c in range(1, 9):
            W[c] = lhs[c] @ ((X * Wt[:, c:c + 1]).t() @ Y[:, c])
        Z = X @ W.t()
        for r, c, z in obj_rows:
            # the probability also depends on the other logits of the row: aim at log(p/(1-p)) + log(sum of the others)
            others = torch.logsumexp(torch.cat([Z[r, :c], Z[r, c + 1:]]), 0)
            Y[r, c] += (z + float(others)) - float(Z[r, c])
    w = (W[:, :-1] / sd).float()
    b = (W[:, -1] - (W[:, :-1] * (mu / sd)).sum(1)).float()
    return w.contiguous(), b.contiguous()


def candidate_sd(sd, w, b):
    out = dict(sd)
    out['bbox_head.fc_cls.weight'], out['bbox_head.fc_cls.bias'] = w, b
    return out


def full_run(sd, stages, p):
    o = OF.FuseTrackOracle(sd, depth=DEPTH)
    res = []
    with torch.no_grad():
        for t in range(NFRAMES):
            x, fcn_score = perturbed(stages[t], p)
            fcn_output = F.interpolate(fcn_score, scale_factor=4, mode='bilinear', align_corners=False)
            det = o.detect(x, (H, W), t == 0)
            r = o.panoptic(x, fcn_output, det)
            res.append({k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in r.items() if k != 'mask_score'})
    return res


def detect_run(sd, stages, p):
    """[(classes, ids)] per frame of the detection + tracking part alone"""
    o = OF.FuseTrackOracle(sd, depth=DEPTH)
    out = []
    with torch.no_grad():
        for t in range(NFRAMES):
            det = o.detect(perturbed(stages[t], p)[0], (H, W), t == 0)
            out.append((det['cls_idx'].tolist(), np.asarray(det['det_obj_ids']).tolist()))
    return out


def same_listing(a, b):
    for ra, rb in zip(a, b):
        for k in ('panoptic_cls_inds', 'panoptic_det_labels', 'panoptic_det_obj_ids', 'keep_inds'):
            if not np.array_equal(ra[k], rb[k]):
                return False, k
        # the map itself: iid noise of this size flips the arg-max of the (random, near-tied) semantic logits at ~1 % of the pixels;
        # a changed instance would move far more. The < 0.1 % bound on the maps is asserted on the real HIP outputs by the GPU test.
        if (ra['panoptic_outputs'] != rb['panoptic_outputs']).mean() >= 3e-2:
            return False, 'pan %.4f%%' % (100 * (ra['panoptic_outputs'] != rb['panoptic_outputs']).mean())
    return True, ''


def main():
    """candidates: fitted heads that pass the margin filter on all frames, each with the oracle's complete outputs. The choice among
    them is made on the GPU (tools/select_separated.py: the HIP path in its three fp32-grade arithmetic modes against the oracle
    listing) — iid noise on the neck output turned out to be a poor proxy for the HIP path's error (it re-draws the ~1000 RPN
    proposals, which the real arithmetic differences do for 2..27 of them), see DESIGN.md 4."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ncand = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    sd = {k: v.float() for k, v in build_sd().items()}
    stages = stage_cache(sd)
    hcache = os.environ.get('VPS_SEP_HIN', '/tmp/vps_sep_hin%s.pt' % ('_c5' if CONFIG5 else ''))
    if os.path.exists(hcache):
        hin0 = torch.load(hcache)
    else:
        with torch.no_grad():
            hin0 = [head_inputs(sd, stages[t]['x']) for t in range(NFRAMES)]
        torch.save(hin0, hcache)
    print('head inputs ready', flush=True)
    outdir = os.path.join(HERE, '_cand')
    os.makedirs(outdir, exist_ok=True)
    found = 0
    for trial in range(400):
        rng = np.random.default_rng(trial)
        nobj = int(rng.integers(8, 15))
        top = (60, 150, 400)[trial % 3]
        lam = (1.0, 0.1, 0.3)[(trial // 3) % 3]
        objs, boxes = choose_objects(hin0, nobj, rng, top)
        probs = (0.63 + 0.35 * (rng.permutation(len(objs)) + 0.5) / len(objs)).tolist()
        w, b = fit_head(hin0, objs, boxes, probs, lam=lam, w_near=1.0)
        ms, ok = [], True
        for t in range(NFRAMES):
            h = hin0[t]
            prob = F.softmax(F.linear(h['feat'], w, b), dim=1)
            m = margins(h['rois'], h['bbox_pred'], prob)
            ms.append(m)
            ok = ok and 3 <= m['K'] <= 40 and m['thr'] >= 2 * MARGIN_P and m['gap'] >= 2 * MARGIN_P and m['iou'] >= 2 * MARGIN_IOU
        print('trial %d nobj %d top %d lam %.1f |w| rms %.3f: %s' % (trial, len(objs), top, lam, float(w.pow(2).mean().sqrt()),
                                                                      [(m['K'], '%.0e' % m['thr'], '%.0e' % m['gap'], '%.0e' % m['iou']) for m in ms]), flush=True)
        if not ok:
            continue
        base = full_run(candidate_sd(sd, w, b), stages, 0)
        rec = dict(weight=w.numpy(), bias=b.numpy(), trial=np.array(trial), margins=np.array([[m['K'], m['thr'], m['gap'], m['iou']] for m in ms]))
        for t, r in enumerate(base):
            for k in ('panoptic_cls_inds', 'panoptic_det_labels', 'panoptic_det_obj_ids', 'keep_inds', 'panoptic_cls_prob'):
                rec['f%d.%s' % (t, k)] = np.asarray(r[k])
            rec['f%d.pan_s4' % t] = r['panoptic_outputs'].astype(np.uint8)[0, ::4, ::4]
        np.savez_compressed(os.path.join(outdir, 'cand%s%03d.npz' % ('_c5_' if CONFIG5 else '', trial)), **rec)
        found += 1
        if CONFIG5:
            kept = [len(r['keep_inds']) for r in base]
            ids_max = max(int(np.asarray(r['panoptic_det_obj_ids']).max()) for r in base)
            if min(kept) < 3 or ids_max < max(kept):          # every frame lists instances, later frames open new ids
                print('  rejected: kept %s ids_max %d' % (kept, ids_max), flush=True)
                found -= 1
                continue
            ties = [int(len(h['proposals']) - len(np.unique(h['proposals'][:, 4].numpy()))) for h in hin0]
            print('  proposals with a score shared with another proposal, per frame: %s' % ties, flush=True)
            np.savez_compressed(os.path.join(HERE, 'config5_fc_cls.npz'), weight=rec['weight'], bias=rec['bias'], trial=rec['trial'], margins=rec['margins'],
                                required=np.array([2 * MARGIN_P, 2 * MARGIN_P, 2 * MARGIN_IOU]), kept=np.array(kept), ids_max=np.array(ids_max),
                                tied_proposals=np.array(ties), **{k: sd[k].numpy() for k in RPN_KEYS})
            print('  written: tests/golden/config5_fc_cls.npz', flush=True)
            return
        print('  candidate %d saved: kept %s ids %s' % (found, [len(r['keep_inds']) for r in base], [r['panoptic_det_obj_ids'].tolist() for r in base]), flush=True)
        if found >= ncand:
            break


if __name__ == '__main__':
    main()
