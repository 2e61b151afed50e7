"""UPSNet-style panoptic head logic of the FuseTrack detector: MaskROI, MaskRemoval, SegTerm + logit combine.

Mirrors mmdet/models/utils/mask_roi.py:24-147, utils/mask_removal.py:23-92, utils/unary_logits.py:70-108 and
detectors/panoptic_fusetrack.py:585-597. The order-defining logic of MaskROI (decode, threshold, sort, NMS, cap) and the
kept-list / instance-table construction run on the device (csrc/head_ops.hip); the host sees the detection list once per
frame (it needs K to size the launches behind it) and builds MaskRemoval's walk from that copy. Every per-pixel / per-box-pair
computation runs in HIP kernels, and nothing of size [k, H, W] is ever materialised.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

import os

from . import hip

# MaskRemoval's box walk on the device: 'hist' (default, round 6) = no chain at all: per-pixel box patterns counted in one pass, one
# wavefront per class decides its boxes over the distinct patterns (vps_mask_removal_hist; a class with 65..127 boxes takes two passes, more: 'dep');
# 'dep' (round 5) = ONE launch, one workgroup per box waiting for the boxes it depends on (vps_mask_removal_dep); 'level' = the round-3
# schedule, one count + one commit launch per dependency level (~20 levels per frame: the recovery path of the two above);
# 'single' = one workgroup per class walking its boxes in order (slowest; kept as the simplest statement of the loop)
MASK_REMOVAL_MODE = os.environ.get('VPS_MASK_REMOVAL', 'hist')
MASK_REMOVAL_SINGLE_LAUNCH = MASK_REMOVAL_MODE == 'single'
HIST_MAX_PAIRS = int(os.environ.get('VPS_MR_HIST_PAIRS', '200'))     # intersecting same-class box pairs up to which 'hist' is used
# frames whose one-launch MaskRemoval reported an expired dependency wait and were finished through the per-level launches (detector.py)
MR_RECOVERIES = [0]


class MaskROI(nn.Module):
    """utils/mask_roi.py:24-147 for the configured mode (class_agnostic=True, clip_boxes=True,
    bbox_class_agnostic=False). `bbox_reg_weights` / `max_det` are the UPSNet yaml-config values
    (tools/config/config.py:47,169) that the reference reads from a global.

    Device-resident (csrc/head_ops.hip): decode + clip + threshold + descending sort (`vps_maskroi_select`), the class-agnostic
    NMS (`vps_nms_batched`, count read on the device), the `max_det` cap / dummy row (`vps_maskroi_finish`). The host reads the
    finished list ONCE (8 KB: the frame's one mid-frame stream drain) — it needs K to size the launches that follow."""

    KCAP = 256        # rows of the result buffer; the uint8 panoptic map names at most 244 instances anyway

    def __init__(self, clip_boxes, bbox_class_agnostic, top_n, num_classes, nms_thresh, score_thresh,
                 class_agnostic=False, bbox_reg_weights=(10., 10., 5., 5.), max_det=100):
        super().__init__()
        assert clip_boxes and class_agnostic and not bbox_class_agnostic
        self.top_n, self.num_classes = top_n, num_classes
        self.nms_thresh, self.score_thresh = nms_thresh, score_thresh
        self.bbox_reg_weights, self.max_det = tuple(bbox_reg_weights), max_det
        self.last = None

    def forward(self, bottom_rois, bbox_delta, cls_prob, im_info, ws, n_valid=None):
        """-> (cls_prob [K], rois [K,5], cls_idx [K] in 1..num_classes-1) device tensors; one dummy row when empty
        (mask_roi.py:136-142). `self.last` keeps the host copy of the list (rows [K,8]: 0,x1,y1,x2,y2,score,class,candidate) and
        the device row buffer for the stages behind it. n_valid: device int32 [1], rows of `bottom_rois` that exist."""
        lib = hip.load()
        dev = bottom_rois.device
        n, nc = bottom_rois.shape[0], self.num_classes
        cap = min(n * (nc - 1), 8192)
        rois = bottom_rois.contiguous(); delta = bbox_delta.contiguous(); prob = cls_prob.contiguous()
        assert delta.shape == (n, 4 * nc) and prob.shape == (n, nc), (delta.shape, prob.shape)
        dets = ws.get('mroi.dets', (cap, 5), zero=False)
        cand = ws.get('mroi.cand', (cap,), dtype=torch.int32, zero=False)
        mcnt = ws.get('mroi.m', (4,), dtype=torch.int32, zero=False)
        cb = (cap + 63) // 64
        mask = ws.get('mroi.mask', (cap * cb,), dtype=torch.int64, zero=False)
        keep = ws.get('mroi.keep', (cap,), dtype=torch.int32, zero=False)
        nkeep = ws.get('mroi.nkeep', (1,), dtype=torch.int32, zero=False)
        res = ws.get('mroi.res', (8 + 8 * self.KCAP,), zero=False)
        sp = hip.stream_ptr()
        wts = (ctypes.c_float * 4)(*self.bbox_reg_weights)
        hip.check(lib.vps_maskroi_select(hip.ptr(rois), hip.ptr(delta), hip.ptr(prob), n, hip.ptr(n_valid), nc, float(self.score_thresh), wts,
                                         float(im_info[0, 0]), float(im_info[0, 1]), hip.ptr(dets), hip.ptr(cand), hip.ptr(mcnt), sp),
                  'vps_maskroi_select')
        hip.check(lib.vps_nms_batched(hip.ptr(dets), 1, cap, hip.ptr(mcnt), float(self.nms_thresh), hip.ptr(mask), hip.ptr(keep),
                                      hip.ptr(nkeep), sp), 'vps_nms_batched')
        hip.check(lib.vps_maskroi_finish(hip.ptr(dets), hip.ptr(cand), hip.ptr(mcnt), hip.ptr(keep), hip.ptr(nkeep), nc, int(self.max_det),
                                         self.KCAP, hip.ptr(res), sp), 'vps_maskroi_finish')
        host = res.cpu().numpy()                  # the ONE mid-frame host sync of the detection branch
        K, status = int(host[0]), int(host[3])
        if status & 1:
            raise hip.VpsHipError('MaskROI: more than 8192 candidates above score_thresh')
        if status & 2:
            raise hip.VpsHipError('MaskROI: more than %d detections after the max_det cap (tied scores)' % self.KCAP)
        rows_h = host[8:8 + 8 * K].reshape(K, 8).copy()
        rows_d = res[8:8 + 8 * K].view(K, 8)
        self.last = dict(K=K, rows_h=rows_h, rows_d=rows_d, ncand=int(host[1]), npost=int(host[2]), nrois=int(host[4]))
        return rows_d[:, 5].contiguous(), rows_d[:, 0:5].contiguous(), rows_d[:, 6].long()


class MaskRemoval(nn.Module):
    """utils/mask_removal.py:23-92. The box loop order (numpy argsort of cls_prob, reversed) and the skip rule are the
    reference's; the cv2.resize + binarise + overlap count + occupancy update of each box run on the device with the
    keep decision taken on the device; boxes are batched into dependency levels (2 launches per level). The walk order and the
    levels are built on the host from the detection list MaskROI already fetched (no D2H here), the kept flags STAY on the
    device: `vps_pan_instances` turns them into the kept list + the instance table of the combine kernel."""

    def __init__(self, fraction_threshold=0.3):
        super().__init__()
        self.fraction_threshold = fraction_threshold

    def forward(self, rows_h, rows_d, mask_prob, im_shape, ws, class_mapping, force_level=False):
        """rows_h / rows_d: the detection list (host numpy / device, [n,8]: 0,x1,y1,x2,y2,score,class,q), mask_prob [n,S,S] device.
        -> dict(inst: device vps_pan_inst table, kinfo: device int32 [4] = (k, masks_valid, status, -), keep: device int32 [n])"""
        dev = mask_prob.device
        H, W = int(im_shape[0]), int(im_shape[1])
        lib = hip.load()
        sp = hip.stream_ptr()
        n = rows_h.shape[0]
        S = mask_prob.shape[-1]
        inst = ws.get('mr.inst', (MaskROI.KCAP * ctypes.sizeof(hip.PanInst),), dtype=torch.uint8, zero=False)
        keep_d = ws.get('mr.keep', (MaskROI.KCAP,), dtype=torch.int32, zero=False)
        kinfo = ws.get('mr.kinfo', (4,), dtype=torch.int32, zero=False)
        kinfo.zero_()
        nclass = max(class_mapping) + 1
        cm = (ctypes.c_int32 * nclass)(*[int(class_mapping.get(c, 0)) for c in range(nclass)])
        rois_np = rows_h[:, 1:5]
        prob_np = rows_h[:, 5]
        cls_np = rows_h[:, 6].astype(np.int64)
        cls0 = cls_np - 1
        out = dict(inst=inst, kinfo=kinfo, keep=keep_d)
        if n == 1 and cls0[0] == -1:
            # the dummy row (mask_removal.py:51-53): keep = [0], all-zero mask logits
            hip.check(lib.vps_pan_instances(None, None, hip.ptr(rows_d), None, 1, cm, nclass, hip.ptr(inst), hip.ptr(keep_d), hip.ptr(kinfo), sp),
                      'vps_pan_instances')
            return out
        sorted_inds = np.argsort(prob_np)[::-1]
        ref_boxes = rois_np.astype(np.int32)
        ncls = int(np.max(cls_np))
        occ = ws.get('mr.occ', (ncls, H, W), dtype=torch.uint8, zero=False)
        flags = ws.get('mr.flags', (max(n, 1),), dtype=torch.int32, zero=False)
        mp = mask_prob.contiguous()
        # dependency levels over the score-sorted list: a box depends on the earlier same-class boxes whose rectangles
        # intersect it (mask_removal.py:75-80 only looks at the class plane inside the box); boxes of one level are independent
        sb = ref_boxes[sorted_inds].astype(np.int64)
        sc = cls0[sorted_inds]
        x0 = np.maximum(sb[:, 0], 0); x1 = np.minimum(sb[:, 2] + 1, W); y0 = np.maximum(sb[:, 1], 0); y1 = np.minimum(sb[:, 3] + 1, H)
        area = np.maximum(x1 - x0, 0) * np.maximum(y1 - y0, 0)
        # (force_level: the detector's second pass after an expired dependency wait of the one-launch kernel)
        # rank of a box among the boxes of its class in walk order (the bit it owns in the pixel patterns of vps_mask_removal_hist)
        rank = np.zeros(n, dtype=np.int64)
        per_cls = np.zeros(1, dtype=np.int64)
        hist_mode = MASK_REMOVAL_MODE == 'hist' and n <= MaskROI.KCAP and ncls <= 32 and int(sc.min()) >= 0 and not force_level
        if hist_mode:
            by_cls = np.argsort(sc, kind='stable')
            per_cls = np.bincount(sc)
            rank[by_cls] = np.arange(n) - (np.cumsum(per_cls) - per_cls)[sc[by_cls]]
            hist_mode = int(rank.max()) <= 126
        if hist_mode and int((per_cls * (per_cls - 1) // 2).sum()) > HIST_MAX_PAIRS:
            # the pattern tables hold 2048 distinct overlap patterns per class: lists whose same-class rectangles intersect in very many
            # pairs (the random-weight bench frames: 97 boxes of one class, 420..500 intersecting pairs, tables overflow) stay on the
            # dependency kernel; street-scene lists have tens (the crowded synthetic test lists: 6..130). A class of more than 64 boxes
            # with that many possible pairs is not examined further; else one vectorised n x n rectangle test (~0.1 ms of host time for
            # 100 boxes - only lists with > HIST_MAX_PAIRS possible pairs pay it).
            if int(per_cls.max()) > 64:
                hist_mode = False
            else:
                inter = ((sc[:, None] == sc[None, :]) & (x0[:, None] < x1[None, :]) & (x0[None, :] < x1[:, None])
                         & (y0[:, None] < y1[None, :]) & (y0[None, :] < y1[:, None]))
                hist_mode = (int(np.count_nonzero(inter)) - n) // 2 <= HIST_MAX_PAIRS
        dep_mode = (MASK_REMOVAL_MODE in ('dep', 'hist') and not hist_mode and W % 4 == 0 and n <= MaskROI.KCAP and S <= 32 and not force_level)
        lvl = np.zeros(n, dtype=np.int64)
        single = MASK_REMOVAL_SINGLE_LAUNCH and not force_level
        if not dep_mode and not single and not hist_mode:
            # (the one-launch kernel finds a box's dependencies itself: this O(n^2) host loop sat on the frame's critical path with the
            # GPU idle - 0.26 ms in the traced frame, profiles/r05_frame_occupancy_traced.json before / after)
            for i in range(1, n):
                dep = (sc[:i] == sc[i]) & (x0[:i] < x1[i]) & (x0[i] < x1[:i]) & (y0[:i] < y1[i]) & (y0[i] < y1[:i])
                if dep.any():
                    lvl[i] = lvl[:i][dep].max() + 1
        order = np.argsort(lvl, kind='stable')
        nlv = int(lvl.max()) + 1
        starts = np.searchsorted(lvl[order], np.arange(nlv + 1))
        host = np.concatenate([sb.reshape(-1), sc, sorted_inds, order, rank]).astype(np.int32)
        meta = torch.from_numpy(host).to(dev, non_blocking=True)
        counts = ws.get('mr.counts', (max(n, 1), 2), dtype=torch.int32, zero=False)
        if not dep_mode and not hist_mode:
            occ.zero_(); counts.zero_()             # (the one-launch entries zero what they use themselves)
        base = meta.data_ptr()
        if hist_mode:
            scratch = ws.get('mr.hist', (MaskROI.KCAP + 2 * (32 + 2 * 32 * 4096),), dtype=torch.int32, zero=False)
            # status word = kinfo[2] (read with the frame's end-of-frame read; bit 2: a pattern table was full -> the detector repeats
            # the walk through the level launches, as for an expired wait of the dependency kernel)
            hip.check(lib.vps_mask_removal_hist(hip.ptr(mp), S, ctypes.c_void_p(base), ctypes.c_void_p(base + 16 * n), ctypes.c_void_p(base + 20 * n),
                                                ctypes.c_void_p(base + 28 * n), int(rank.max()), n, ncls, H, W, hip.ptr(scratch), scratch.numel() * 4,
                                                float(self.fraction_threshold), hip.ptr(flags), ctypes.c_void_p(kinfo.data_ptr() + 8), sp),
                      'vps_mask_removal_hist')
            nlv = 0
        elif dep_mode:
            done = ws.get('mr.done', (MaskROI.KCAP,), dtype=torch.int32, zero=False)
            # status word = kinfo[2] (read with the frame's end-of-frame read; bit 2: a dependency wait expired)
            hip.check(lib.vps_mask_removal_dep(hip.ptr(mp), S, ctypes.c_void_p(base), ctypes.c_void_p(base + 16 * n), ctypes.c_void_p(base + 20 * n),
                                               n, ncls, H, W, hip.ptr(occ), float(self.fraction_threshold), hip.ptr(flags), hip.ptr(done),
                                               ctypes.c_void_p(kinfo.data_ptr() + 8), sp), 'vps_mask_removal_dep')
            nlv = 0
        elif single:
            # A/B switch: the whole walk in ONE launch, one workgroup per class walking its boxes in order (csrc/pan_ops.hip)
            hip.check(lib.vps_mask_removal(hip.ptr(mp), S, ctypes.c_void_p(base), ctypes.c_void_p(base + 16 * n), ctypes.c_void_p(base + 20 * n),
                                           n, ncls, H, W, hip.ptr(occ), float(self.fraction_threshold), hip.ptr(flags), sp), 'vps_mask_removal')
            nlv = 0
        for l in range(nlv):
            a, b = int(starts[l]), int(starts[l + 1])
            hip.check(lib.vps_mask_level(hip.ptr(mp), S, ctypes.c_void_p(base), ctypes.c_void_p(base + 16 * n),
                                         ctypes.c_void_p(base + 20 * n), ctypes.c_void_p(base + 24 * n + 4 * a), b - a,
                                         int(area[order[a:b]].max()), H, W, hip.ptr(occ), hip.ptr(counts),
                                         float(self.fraction_threshold), hip.ptr(flags), sp), 'vps_mask_level')
        hip.check(lib.vps_pan_instances(ctypes.c_void_p(base + 20 * n), hip.ptr(flags), hip.ptr(rows_d), ctypes.c_void_p(base), n, cm, nclass,
                                        hip.ptr(inst), hip.ptr(keep_d), hip.ptr(kinfo), sp), 'vps_pan_instances')
        out['_meta'] = meta          # keeps the uploaded walk alive until the kernels have run
        return out


def panoptic_combine(fcn_score, removal, mask_prob, num_stuff, num_classes, out_hw, ws):
    """SegTerm (utils/unary_logits.py:81-108 with boxes = mask_rois*4.0*0.25) + logit concat + argmax
    (panoptic_fusetrack.py:588-597), fused. `removal`: MaskRemoval's output — the instance table and its length live on the
    device (instance j of the panoptic map is kept detection j)."""
    H, W = out_hw
    pan = ws.get('pan.out', (1, H, W), dtype=torch.uint8, zero=False)
    sem = ws.get('sem.out', (1, H, W), dtype=torch.uint8, zero=False)
    S = mask_prob.shape[-1]
    rc = hip.load().vps_panoptic_combine_dev(fcn_score.ptr(), fcn_score.ld, fcn_score.H, fcn_score.W, num_classes, num_stuff,
                                             hip.ptr(removal['inst']), hip.ptr(removal['kinfo']), hip.ptr(mask_prob), S, hip.ptr(pan),
                                             hip.ptr(sem), H, W, hip.stream_ptr())
    hip.check(rc, 'vps_panoptic_combine_dev(S=%d, mask_prob %s, score %dx%d ld %d, out %dx%d, classes %d/%d)' % (
        S, tuple(mask_prob.shape), fcn_score.H, fcn_score.W, fcn_score.ld, H, W, num_stuff, num_classes))
    return pan, sem
