"""UPSNet-style panoptic head logic of the FuseTrack detector: MaskROI, MaskRemoval, SegTerm + logit combine.

Mirrors mmdet/models/utils/mask_roi.py:24-147, utils/mask_removal.py:23-92, utils/unary_logits.py:70-108 and
detectors/panoptic_fusetrack.py:585-597. The order-defining host logic (numpy argsort, score cap, per-box
sequential overlap rule) stays on the host exactly as in the reference because it defines instance ids; every
per-pixel / per-box-pair computation runs in HIP kernels, and nothing of size [k, H, W] is ever materialised.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import hip


def _bbox_transform(boxes, deltas, weights):
    """utils/upsnet/bbox/bbox_transform.py:290-330 (numpy, dtype of deltas)"""
    if boxes.shape[0] == 0:
        return np.zeros((0, deltas.shape[1]), dtype=deltas.dtype)
    boxes = boxes.astype(deltas.dtype, copy=False)
    widths = boxes[:, 2] - boxes[:, 0] + 1.0
    heights = boxes[:, 3] - boxes[:, 1] + 1.0
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx; dy = deltas[:, 1::4] / wy
    dw = np.minimum(deltas[:, 2::4] / ww, np.log(1000. / 16.)); dh = np.minimum(deltas[:, 3::4] / wh, np.log(1000. / 16.))
    pcx = dx * widths[:, np.newaxis] + ctr_x[:, np.newaxis]
    pcy = dy * heights[:, np.newaxis] + ctr_y[:, np.newaxis]
    pw = np.exp(dw) * widths[:, np.newaxis]; ph = np.exp(dh) * heights[:, np.newaxis]
    out = np.zeros(deltas.shape, dtype=deltas.dtype)
    out[:, 0::4] = pcx - 0.5 * pw; out[:, 1::4] = pcy - 0.5 * ph
    out[:, 2::4] = pcx + 0.5 * pw - 1; out[:, 3::4] = pcy + 0.5 * ph - 1
    return out


def _clip_boxes(boxes, im_shape):
    """bbox_transform.py:45-60"""
    boxes[:, 0::4] = np.maximum(np.minimum(boxes[:, 0::4], im_shape[1] - 1), 0)
    boxes[:, 1::4] = np.maximum(np.minimum(boxes[:, 1::4], im_shape[0] - 1), 0)
    boxes[:, 2::4] = np.maximum(np.minimum(boxes[:, 2::4], im_shape[1] - 1), 0)
    boxes[:, 3::4] = np.maximum(np.minimum(boxes[:, 3::4], im_shape[0] - 1), 0)
    return boxes


class MaskROI(nn.Module):
    """utils/mask_roi.py:24-147 for the configured mode (class_agnostic=True, clip_boxes=True,
    bbox_class_agnostic=False). `bbox_reg_weights` / `max_det` are the UPSNet yaml-config values
    (tools/config/config.py:47,169) that the reference reads from a global."""

    def __init__(self, clip_boxes, bbox_class_agnostic, top_n, num_classes, nms_thresh, score_thresh,
                 class_agnostic=False, bbox_reg_weights=(10., 10., 5., 5.), max_det=100):
        super().__init__()
        assert clip_boxes and class_agnostic and not bbox_class_agnostic
        self.top_n, self.num_classes = top_n, num_classes
        self.nms_thresh, self.score_thresh = nms_thresh, score_thresh
        self.bbox_reg_weights, self.max_det = tuple(bbox_reg_weights), max_det

    def forward(self, bottom_rois, bbox_delta, cls_prob, im_info, ws=None):
        """-> (cls_prob [K], rois [K,5], cls_idx [K] in 1..num_classes-1) device tensors; one dummy row when empty
        (mask_roi.py:136-142)."""
        dev = bottom_rois.device
        rois_np = bottom_rois.detach().cpu().numpy()
        delta_np = bbox_delta.detach().cpu().numpy()
        prob_cpu = cls_prob.detach().cpu()
        prob_np = prob_cpu.numpy()
        n = rois_np.shape[0]
        nc = self.num_classes
        proposal = _bbox_transform(rois_np[:, 1:], delta_np, self.bbox_reg_weights)
        proposal = _clip_boxes(proposal, im_info[0, :2])
        # class-agnostic flattening: candidate q = roi*(nc-1) + (cls-1)  (mask_roi.py:60-74)
        cand_prob = prob_np[:, 1:].reshape(-1)
        cand_box = proposal.reshape((n, -1, 4))[:, 1:, :].reshape((-1, 4))
        cand_cls = np.tile(np.arange(1, nc), n)
        inds = np.where(cand_prob > self.score_thresh)[0]
        scores_j = cand_prob[inds]
        dets_j = np.hstack((cand_box[inds], scores_j[:, np.newaxis])).astype(np.float32)
        if len(dets_j) == 0:
            return (torch.ones(1, device=dev), torch.zeros(1, 5, device=dev), torch.zeros(1, dtype=torch.long, device=dev))
        # gpu_nms (utils/upsnet/nms/gpu_nms.pyx:23-38): order = scores.argsort()[::-1]; device bitmask + device greedy
        order = dets_j[:, 4].argsort()[::-1]
        sorted_dets = np.ascontiguousarray(dets_j[order, :])
        m = sorted_dets.shape[0]
        lib = hip.load()
        bd = torch.from_numpy(sorted_dets).to(dev)
        cb = (m + 63) // 64
        mask = torch.empty(m * cb, dtype=torch.int64, device=dev)
        keep = torch.empty(m, dtype=torch.int32, device=dev)
        nkeep = torch.zeros(1, dtype=torch.int32, device=dev)
        cnt = torch.tensor([m], dtype=torch.int32, device=dev)
        hip.check(lib.vps_nms_batched(hip.ptr(bd), 1, m, hip.ptr(cnt), float(self.nms_thresh), hip.ptr(mask), hip.ptr(keep),
                                      hip.ptr(nkeep), hip.stream_ptr()), 'vps_nms_batched')
        nk = int(nkeep.item())
        keep_np = order[keep[:nk].cpu().numpy().astype(np.int64)]     # list(order[keep])
        nms_dets = dets_j[keep_np, :]
        sel = inds[keep_np]
        if self.max_det > 0 and len(nms_dets) > self.max_det:     # mask_roi.py:106-121
            image_thresh = np.sort(nms_dets[:, -1])[-self.max_det]
            k2 = np.where(nms_dets[:, -1] >= image_thresh)[0]
            nms_dets = nms_dets[k2, :]; sel = sel[k2]
        boxes = np.zeros((nms_dets.shape[0], 5))
        boxes[:, 1:] = nms_dets[:, :-1]
        scores_th = prob_cpu[:, 1:].contiguous().view(-1)[torch.from_numpy(sel).long()]   # from the torch tensor (mask_roi.py:94)
        return (scores_th.to(dev), torch.from_numpy(boxes).float().to(dev), torch.from_numpy(cand_cls[sel]).long().to(dev))


class MaskRemoval(nn.Module):
    """utils/mask_removal.py:23-92. The box loop order (numpy argsort of cls_prob, reversed) and the skip rule are the
    reference's; the cv2.resize + binarise + overlap count + occupancy update of each box run on the device with the
    keep decision taken on the device; boxes are batched into dependency levels (2 launches per level) and there is ONE host sync."""

    def __init__(self, fraction_threshold=0.3):
        super().__init__()
        self.fraction_threshold = fraction_threshold

    def forward(self, mask_rois, cls_prob, mask_prob, cls_idx, im_shape, ws):
        """mask_rois [n,4], cls_prob [n], mask_prob [n,S,S] (device), cls_idx [n] -> (keep_inds np.int64 in score order,
        ref_boxes np.int32 [n,4] truncated boxes, masks_valid). When nothing is kept the reference returns keep_inds=[0]
        with an all-zero mask_energy (mask_removal.py:51-53,89-91): masks_valid=False."""
        dev = mask_prob.device
        H, W = int(im_shape[0]), int(im_shape[1])
        rois_np = mask_rois.detach().cpu().numpy()
        prob_np = cls_prob.detach().cpu().numpy()
        cls_np = cls_idx.detach().cpu().numpy()
        n = rois_np.shape[0]
        S = mask_prob.shape[-1]
        sorted_inds = np.argsort(prob_np)[::-1]
        ref_boxes = rois_np.astype(np.int32)
        cls0 = cls_np - 1
        if n == 1 and cls0[0] == -1:
            return np.array([0], dtype=np.int64), ref_boxes, False
        lib = hip.load()
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
        lvl = np.zeros(n, dtype=np.int64)
        for i in range(1, n):
            dep = (sc[:i] == sc[i]) & (x0[:i] < x1[i]) & (x0[i] < x1[:i]) & (y0[:i] < y1[i]) & (y0[i] < y1[:i])
            if dep.any():
                lvl[i] = lvl[:i][dep].max() + 1
        order = np.argsort(lvl, kind='stable')
        nlv = int(lvl.max()) + 1
        starts = np.searchsorted(lvl[order], np.arange(nlv + 1))
        host = np.concatenate([sb.reshape(-1), sc, sorted_inds, order]).astype(np.int32)
        meta = torch.from_numpy(host).to(dev)
        counts = ws.get('mr.counts', (max(n, 1), 2), dtype=torch.int32, zero=False)
        occ.zero_(); counts.zero_()
        base = meta.data_ptr()
        sp = hip.stream_ptr()
        for l in range(nlv):
            a, b = int(starts[l]), int(starts[l + 1])
            hip.check(lib.vps_mask_level(hip.ptr(mp), S, ctypes.c_void_p(base), ctypes.c_void_p(base + 16 * n),
                                         ctypes.c_void_p(base + 20 * n), ctypes.c_void_p(base + 24 * n + 4 * a), b - a,
                                         int(area[order[a:b]].max()), H, W, hip.ptr(occ), hip.ptr(counts),
                                         float(self.fraction_threshold), hip.ptr(flags), sp), 'vps_mask_level')
        fl = flags[:n].cpu().numpy()
        keep_inds = [int(sorted_inds[pos]) for pos in range(n) if fl[pos]]
        if len(keep_inds) == 0:
            return np.array([0], dtype=np.int64), ref_boxes, False
        return np.array(keep_inds, dtype=np.int64), ref_boxes, True


def panoptic_combine(fcn_score, mask_rois_np, cls_idx_np, ref_boxes, keep_inds, mask_prob, class_mapping, num_stuff,
                     num_classes, out_hw, ws, masks_valid=True):
    """SegTerm (utils/unary_logits.py:81-108 with boxes = mask_rois*4.0*0.25) + logit concat + argmax
    (panoptic_fusetrack.py:588-597), fused. keep_inds index the ORIGINAL detections; instance j of the panoptic map
    is keep_inds[j]. masks_valid=False: the pasted mask logits are all zero (MaskRemoval kept nothing)."""
    H, W = out_hw
    dev = fcn_score.t.device
    k = len(keep_inds)
    inst = (hip.PanInst * max(k, 1))()
    for j, i in enumerate(keep_inds):
        b = (mask_rois_np[i].astype(np.float32) * np.float32(4.0)) * (1 / 4.0)     # float32 * 4.0, then numpy * 0.25
        it = inst[j]
        c = int(cls_idx_np[i])
        if c == 0:
            it.sx0 = it.sy0 = it.sx1 = it.sy1 = 0; it.seg_ch = 0
        else:
            it.sy0 = int(b[1]); it.sy1 = int(b[3].round() + 1)
            it.sx0 = int(b[0]); it.sx1 = int(b[2].round() + 1)
            it.seg_ch = int(class_mapping[c])
        if not masks_valid:
            it.bx1, it.by1, it.bx2, it.by2 = 0, 0, -1, -1     # empty paste region
        else:
            rb = ref_boxes[i]
            it.bx1, it.by1, it.bx2, it.by2 = int(rb[0]), int(rb[1]), int(rb[2]), int(rb[3])
        it.mask_idx = int(i)
    inst_d = torch.frombuffer(bytearray(bytes(inst)), dtype=torch.uint8).to(dev)
    pan = ws.get('pan.out', (1, H, W), dtype=torch.uint8, zero=False)
    sem = ws.get('sem.out', (1, H, W), dtype=torch.uint8, zero=False)
    S = mask_prob.shape[-1]
    rc = hip.load().vps_panoptic_combine(fcn_score.ptr(), fcn_score.ld, fcn_score.H, fcn_score.W, num_classes, num_stuff,
                                         hip.ptr(inst_d), k, hip.ptr(mask_prob), S, hip.ptr(pan), hip.ptr(sem), H, W,
                                         hip.stream_ptr())
    hip.check(rc, 'vps_panoptic_combine(k=%d, S=%d, mask_prob %s, score %dx%d ld %d, out %dx%d, classes %d/%d)' % (
        k, S, tuple(mask_prob.shape), fcn_score.H, fcn_score.W, fcn_score.ld, H, W, num_stuff, num_classes))
    return pan, sem
