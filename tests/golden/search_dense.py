"""A DENSE strict fixture at 1024x2048 (VERDICT r3 "Next round" #3): 30..100 well-separated detections per frame over 6 frames, so that
the tracker memory grows past 60 entries and matched / lost / new objects occur in every frame - chosen by ORACLE margins only (the
HIP path is not run here, nor anywhere else in the selection: tools/select_separated.py is not used for this fixture).

Same construction as search_separated.py (which it imports): everything but `bbox_head.fc_cls` stays `synth.synth_state_dict(seed 0)`;
the classification layer is a weighted ridge regression on the shared-FC features that maps chosen (RoI, class) pairs to target
probabilities and everything else far below MaskROI's 0.6 threshold. Differences: objects are drawn in EVERY frame and followed
forwards and backwards through the clip (an object may be visible in a sub-range of the frames: lost and new objects for the
tracker), simultaneously visible objects overlap by IoU < 0.3 (the class-agnostic NMS compares them: far from its 0.5), each object
has ONE target probability for the whole clip, all targets evenly spaced in (0.62, 0.985); non-object rows that come out above
0.35 are re-weighted x10 and the fit repeated (`fit_dense`).

Accepted: the FIRST trial (fixed trial order, seeds 1000, 1001, ...) whose every frame has 30 <= K <= 100 detections, every
candidate probability >= MARGIN_THR from the 0.6 threshold, adjacent kept scores >= MARGIN_GAP apart, every IoU the greedy NMS
compares >= MARGIN_IOU from 0.5 (the HIP path's measured score error at this size is <= 9e-4, profiles/r03_fullsize_sep_strict_report.txt:
the gap margin is 2.8x that, the threshold margin 11x), and whose oracle run reaches a tracker memory >= 60. How dense it can be is
bounded by the construction: the 1024-d features of overlapping RoIs are nearly collinear, and beyond ~45 objects per frame a linear
read-out can no longer hold the objects on their targets AND every partial-overlap neighbour below the threshold (50+ per frame:
adjacent-score gaps collapse to 1e-4; measured while writing this script).

    python tests/golden/search_dense.py            # ~10 min of CPU for the 6 staged frames, then ~5 s per trial
    python tests/golden/make_golden.py fullsize_dense
"""
import math
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import search_separated as S          # noqa: E402
from oracle import fusetrack as OF    # noqa: E402
from vps_amd import synth             # noqa: E402

NF = 6
MARGIN_THR, MARGIN_GAP, MARGIN_IOU = 1.0e-2, 2.5e-3, 2.0e-2
CACHE = os.environ.get('VPS_DENSE_CACHE', '/tmp/vps_dense_cache.pt')
HIN = os.environ.get('VPS_DENSE_HIN', '/tmp/vps_dense_hin.pt')


def stages(sd):
    if os.path.exists(CACHE):
        return torch.load(CACHE)
    o = OF.FuseTrackOracle(sd)
    frames = synth.synth_clip(S.H, S.W, NF, S.SEED)
    out, prev = [], None
    with torch.no_grad():
        for t in range(NF):
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


def choose_objects(hin, rng, per_frame, max_iou=0.3, min_size=16, track_iou=0.55):
    """objects = (class, [roi index per frame or -1]). Frame by frame, (roi, class) pairs are drawn until `per_frame` objects are
    visible there; a new object is followed into the other frames through the clip's known translation (synth_clip: frame t = base
    shifted by t * (2, 1)); it is accepted when, in every frame where it is visible, it overlaps no visible object by >= max_iou"""
    boxes = [S.refined_boxes(h) for h in hin]
    objs = []
    vis = [[] for _ in range(NF)]                 # per frame: boxes of the visible objects

    def clear(t, b):
        return all(float(OF.bbox_overlaps(b[None], o[None])) < max_iou for o in vis[t])
    for t0 in range(NF):
        n = boxes[t0].shape[0]
        for i in rng.permutation(n):
            if len(vis[t0]) >= per_frame:
                break
            c = int(rng.integers(0, 8))
            b = boxes[t0][i, c]
            if min(float(b[2] - b[0]), float(b[3] - b[1])) < min_size or not clear(t0, b):
                continue
            track = [-1] * NF
            track[t0] = int(i)
            ok = True
            for t in list(range(t0 + 1, NF)) + list(range(t0 - 1, -1, -1)):
                d = t - t0
                shifted = b - torch.tensor([2.0 * d, 1.0 * d, 2.0 * d, 1.0 * d])
                iou = OF.bbox_overlaps(shifted[None], boxes[t][:, c])[0]
                j = int(iou.argmax())
                if float(iou[j]) >= track_iou:
                    if not clear(t, boxes[t][j, c]):
                        ok = False
                        break
                    track[t] = j
            if not ok:
                continue
            for t in range(NF):
                if track[t] >= 0:
                    vis[t].append(boxes[t][track[t], c])
            objs.append((c, track))
    return objs, boxes


def fit_dense(hin, objs, boxes, probs, lam=0.03, w_obj=1000.0, rounds=10, cold_w=1.0, hot_z=-0.6, mult=10.0):
    """search_separated.fit_head as an ACTIVE-SET fit: the objects are interpolated (weight w_obj, re-targeted every round by their
    miss: the probability also depends on the row's other logits); a non-object (row, class) only has to stay far below MaskROI's
    0.6 threshold, so it pulls with full weight (x4 per round, target logit -7) while its fitted logit is above -2 and with weight
    0.02 once it is below -4 - the least-squares capacity goes where the margin is at stake instead of into fitting -7 exactly"""
    Fm = torch.cat([h['feat'] for h in hin], 0).double()
    mu, sdv = Fm.mean(0), Fm.std(0) + 1e-6
    X = torch.cat([(Fm - mu) / sdv, torch.ones(Fm.shape[0], 1, dtype=torch.float64)], 1)
    n = hin[0]['feat'].shape[0]
    Y = torch.full((Fm.shape[0], 9), -7.0, dtype=torch.float64); Y[:, 0] = 0.0
    is_obj = torch.zeros(Fm.shape[0], 9, dtype=torch.bool)
    obj_row = torch.zeros(Fm.shape[0], dtype=torch.bool)
    obj_rows = []
    for (c, track), p in zip(objs, probs):
        for t, j in enumerate(track):
            if j >= 0:
                z = math.log(p / (1 - p))
                Y[t * n + j, c + 1] = z
                is_obj[t * n + j, c + 1] = True; obj_row[t * n + j] = True
                obj_rows.append((t * n + j, c + 1, z))
    boost = torch.ones(Fm.shape[0], 9, dtype=torch.float64)
    Wt = torch.ones(Fm.shape[0], 9, dtype=torch.float64)
    Wt[obj_row] = w_obj                       # the whole row of an object: its other classes must stay down exactly as much
    reg = lam * torch.eye(X.shape[1], dtype=torch.float64) * X.shape[0]; reg[-1, -1] = 0
    W = torch.zeros(9, X.shape[1], dtype=torch.float64)
    nbad = 0
    for rnd in range(rounds):
        for c in range(1, 9):
            Xw = X * Wt[:, c:c + 1]
            W[c] = torch.linalg.solve(X.t() @ Xw + reg, Xw.t() @ Y[:, c])
        Z = X @ W.t()
        for r, c, z in obj_rows:
            others = torch.logsumexp(torch.cat([Z[r, :c], Z[r, c + 1:]]), 0)
            Y[r, c] += (z + float(others)) - float(Z[r, c])
        neg = ~is_obj & ~obj_row[:, None]
        neg[:, 0] = False
        hot = neg & (Z > hot_z)
        cold = neg & (Z < -4.0)
        boost[hot] *= mult
        Wt[neg] = boost[neg]
        Wt[cold] = cold_w * boost[cold]
        nbad = int((neg & (torch.softmax(Z, 1) > 0.35)).sum())
    w = (W[:, :-1] / sdv).float()
    b = (W[:, -1] - (W[:, :-1] * (mu / sdv)).sum(1)).float()
    return w.contiguous(), b.contiguous(), nbad


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    S.NFRAMES = NF
    sd = {k: v.float() for k, v in S.build_sd().items()}
    st = stages(sd)
    if os.path.exists(HIN):
        hin = torch.load(HIN)
    else:
        with torch.no_grad():
            hin = [S.head_inputs(sd, st[t]['x']) for t in range(NF)]
        torch.save(hin, HIN)
    print('head inputs ready: %d frames' % len(hin), flush=True)
    for trial in range(200):
        rng = np.random.default_rng(1000 + trial)
        per_frame = (28, 32, 36, 40)[trial % 4]
        lam = (0.03, 0.1, 0.01)[(trial // 4) % 3]
        objs, boxes = choose_objects(hin, rng, per_frame)
        probs = (0.62 + 0.365 * (rng.permutation(len(objs)) + 0.5) / len(objs)).tolist()
        t0 = time.time()
        w, b, nbad = fit_dense(hin, objs, boxes, probs, lam=lam)
        ms, ok = [], True
        for t in range(NF):
            prob = F.softmax(F.linear(hin[t]['feat'], w, b), dim=1)
            m = S.margins(hin[t]['rois'], hin[t]['bbox_pred'], prob)
            ms.append(m)
            ok = ok and 30 <= m['K'] <= 100 and m['thr'] >= MARGIN_THR and m['gap'] >= MARGIN_GAP and m['iou'] >= MARGIN_IOU
        print('trial %d (%.0f s, %d rows still above 0.35) per_frame %d lam %.2f objects %d visible %s: %s' % (trial, time.time() - t0, nbad, per_frame, lam, len(objs), [sum(1 for _, tr in objs if tr[t] >= 0) for t in range(NF)],
                                                                             [(m['K'], '%.0e' % m['thr'], '%.0e' % m['gap'], '%.0e' % m['iou']) for m in ms]), flush=True)
        if not ok:
            continue
        cand = S.candidate_sd(sd, w, b)
        base = S.full_run(cand, st, 0)
        kept = [len(r['keep_inds']) for r in base]
        ids = [np.asarray(r['panoptic_det_obj_ids']) for r in base]
        mem = int(max(int(i.max()) for i in ids if len(i))) + 1
        print('  oracle run: kept %s, largest id %d' % (kept, mem - 1), flush=True)
        if mem < 60:
            continue
        np.savez_compressed(os.path.join(HERE, 'dense_fc_cls.npz'), weight=w.numpy(), bias=b.numpy(), trial=np.array(trial),
                            margins=np.array([[m['K'], m['thr'], m['gap'], m['iou']] for m in ms]),
                            required=np.array([MARGIN_THR, MARGIN_GAP, MARGIN_IOU]), kept=np.array(kept), ids_max=np.array(mem - 1))
        print('saved tests/golden/dense_fc_cls.npz (trial %d)' % trial)
        return
    raise SystemExit('no candidate passed the margins')


if __name__ == '__main__':
    main()
