"""A DENSE strict fixture at 1024x2048 (VERDICT r3 "Next round" #3): 40..100 well-separated detections per frame over 6 frames, so that
the tracker memory grows past 60 entries and matched / lost / new objects occur in every frame - chosen by ORACLE margins only (the
HIP path is not run here, nor anywhere else in the selection: tools/select_separated.py is not used for this fixture).

Same construction as search_separated.py (which it imports): everything but `bbox_head.fc_cls` stays `synth.synth_state_dict(seed 0)`;
the classification layer is a weighted ridge regression on the shared-FC features that maps chosen (RoI, class) pairs to target
probabilities and everything else far below MaskROI's 0.6 threshold. Differences: objects are drawn in EVERY frame and followed
forwards and backwards through the clip (an object may be visible in a sub-range of the frames: lost and new objects for the
tracker), simultaneously visible objects overlap by IoU < 0.3 (the class-agnostic NMS compares them: far from its 0.5), each object
has ONE target probability for the whole clip, all targets evenly spaced in (0.62, 0.985).

Accepted: the first fit whose every frame has 40 <= K <= 100 detections, every candidate probability >= MARGIN_THR from the 0.6
threshold, adjacent kept scores >= MARGIN_GAP apart, every IoU the greedy NMS compares >= MARGIN_IOU from 0.5 (the HIP path's
measured score error at this size is <= 9e-4, profiles/r03_fullsize_sep_strict_report.txt: the gap margin is 2.8x that, the
threshold margin 11x), and whose oracle run reaches a tracker memory >= 60.

    python tests/golden/search_dense.py            # ~20 min of CPU for the 6 staged frames, then seconds per trial
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
        per_frame = (60, 50, 70, 45)[trial % 4]
        lam = (0.03, 0.1, 0.01)[(trial // 4) % 3]
        objs, boxes = choose_objects(hin, rng, per_frame)
        probs = (0.62 + 0.365 * (rng.permutation(len(objs)) + 0.5) / len(objs)).tolist()
        w, b = S.fit_head(hin, objs, boxes, probs, lam=lam, w_near=1.0)
        ms, ok = [], True
        for t in range(NF):
            prob = F.softmax(F.linear(hin[t]['feat'], w, b), dim=1)
            m = S.margins(hin[t]['rois'], hin[t]['bbox_pred'], prob)
            ms.append(m)
            ok = ok and 40 <= m['K'] <= 100 and m['thr'] >= MARGIN_THR and m['gap'] >= MARGIN_GAP and m['iou'] >= MARGIN_IOU
        print('trial %d per_frame %d lam %.2f objects %d visible %s: %s' % (trial, per_frame, lam, len(objs), [sum(1 for _, tr in objs if tr[t] >= 0) for t in range(NF)],
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
