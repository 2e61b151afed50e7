"""Oracle of PanopticFuseTrack.simple_test (TEST INFRASTRUCTURE ONLY): pure PyTorch-CPU fp32 + numpy.
Functional over a state_dict with the reference's key names. Paths relative to /root/reference/mmdet.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .flownet2 import flownet2

# configs/cityscapes/fusetrack.py + tools/config/config.py:47,169 + panoptic_fusetrack.py:83-87
CFG = dict(
    mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375],
    anchor_scales=[8], anchor_ratios=[0.5, 1.0, 2.0], anchor_strides=[4, 8, 16, 32, 64],
    rpn=dict(nms_pre=1000, nms_post=1000, max_num=1000, nms_thr=0.7, min_bbox_size=0),
    featmap_strides=[4, 8, 16, 32], finest_scale=56,
    bbox_reg_weights=(10., 10., 5., 5.), max_det=100,
    mask_roi=dict(score_thresh=0.6, nms_thresh=0.5),
    mask_removal_fraction=0.3, match_coeff=[1.0, 2.0, 10.0],
    class_mapping={1: 11, 2: 12, 3: 13, 4: 14, 5: 15, 6: 16, 7: 17, 8: 18},
    num_stuff=11, num_classes=19,
)


# ------------------------------------------------------------------ backbone (models/backbones/resnet.py)
def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        False, 0.0, 1e-5)


def _bottleneck(sd, p, x, stride, has_ds):
    # resnet.py:220-266, style='pytorch' (stride on the 3x3)
    out = F.relu(_bn(sd, p + '.bn1', F.conv2d(x, sd[p + '.conv1.weight'])))
    out = F.relu(_bn(sd, p + '.bn2', F.conv2d(out, sd[p + '.conv2.weight'], stride=stride, padding=1)))
    out = _bn(sd, p + '.bn3', F.conv2d(out, sd[p + '.conv3.weight']))
    identity = x
    if has_ds:
        identity = _bn(sd, p + '.downsample.1', F.conv2d(x, sd[p + '.downsample.0.weight'], stride=stride))
    return F.relu(out + identity)


def resnet(sd, p, x, depth=50):
    # resnet.py:506-517, stem :453-465
    blocks = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}[depth]
    x = F.relu(_bn(sd, p + 'bn1', F.conv2d(x, sd[p + 'conv1.weight'], stride=2, padding=3)))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = []
    for i, nb in enumerate(blocks):
        for j in range(nb):
            stride = (1 if i == 0 else 2) if j == 0 else 1
            x = _bottleneck(sd, '%slayer%d.%d' % (p, i + 1, j), x, stride, j == 0)
        outs.append(x)
    return outs


def fpn(sd, p, feats):
    # necks/fpn.py:100-139 (no norm, no activation; P6 = max_pool2d(k=1, s=2))
    lats = [F.conv2d(f, sd['%slateral_convs.%d.conv.weight' % (p, i)], sd['%slateral_convs.%d.conv.bias' % (p, i)])
            for i, f in enumerate(feats)]
    for i in range(len(lats) - 1, 0, -1):
        lats[i - 1] = lats[i - 1] + F.interpolate(lats[i], scale_factor=2, mode='nearest')
    outs = [F.conv2d(l, sd['%sfpn_convs.%d.conv.weight' % (p, i)], sd['%sfpn_convs.%d.conv.bias' % (p, i)], padding=1)
            for i, l in enumerate(lats)]
    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
    return outs


# ------------------------------------------------------------------ extra neck (models/extra_necks/bfp_tcea.py)
def warping_layer(x, flow):
    # flow_modules/flow_modules.py:126-148 (grid_sample defaults: bilinear, zeros, align_corners=False)
    b, _, h, w = x.shape
    gh = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, w).expand(b, 1, h, w)
    gv = torch.linspace(-1.0, 1.0, h).view(1, 1, h, 1).expand(b, 1, h, w)
    grid = torch.cat([gh, gv], 1)
    fg = torch.zeros_like(flow)
    fg[:, 0] = flow[:, 0] / ((flow.size(3) - 1.0) / 2.0)
    fg[:, 1] = flow[:, 1] / ((flow.size(2) - 1.0) / 2.0)
    grid = (grid + fg).permute(0, 2, 3, 1)
    return F.grid_sample(x, grid, mode='bilinear', padding_mode='zeros', align_corners=False)


def liteflownet_corr(sd, p, x1, x2, flow_init):
    # flow_modules.py:50-73: corr(81) -> cat[x1, corr, flow] -> 3x(conv3x3+LeakyReLU) + conv3x3
    corr = ops.correlation(x1.contiguous(), x2.contiguous(), pad_size=4, kernel_size=1, max_displacement=4,
                           stride1=1, stride2=1)
    x = torch.cat([x1, corr, flow_init], dim=1)
    for i in range(3):
        q = '%sflow_estimator.convs.%d.0' % (p, i)
        x = F.leaky_relu(F.conv2d(x, sd[q + '.weight'], sd[q + '.bias'], padding=1), 0.1)
    q = p + 'flow_estimator.convs.3'
    return F.conv2d(x, sd[q + '.weight'], sd[q + '.bias'], padding=1)


def tcea_fusion(sd, p, aligned, center=0):
    # utils/tcea_modules.py:50-78
    cv = lambda n, t, pad=0: F.conv2d(t, sd[p + n + '.weight'], sd[p + n + '.bias'], padding=pad)
    lrelu = lambda t: F.leaky_relu(t, 0.1)
    B, N, C, H, W = aligned.shape
    emb_ref = cv('tAtt_2', aligned[:, center].clone(), 1)
    emb = cv('tAtt_1', aligned.view(-1, C, H, W), 1).view(B, N, -1, H, W)
    cor_l = [torch.sum(emb[:, i] * emb_ref, 1).unsqueeze(1) for i in range(N)]
    cor_prob = torch.sigmoid(torch.cat(cor_l, dim=1))
    cor_prob = cor_prob.unsqueeze(2).repeat(1, 1, C, 1, 1).view(B, -1, H, W)
    aligned = aligned.view(B, -1, H, W) * cor_prob
    fea = lrelu(cv('fea_fusion', aligned))
    att = lrelu(cv('sAtt_1', aligned))
    att_max = F.max_pool2d(att, 3, stride=2, padding=1)
    att_avg = F.avg_pool2d(att, 3, stride=2, padding=1)
    att = lrelu(cv('sAtt_2', torch.cat([att_max, att_avg], dim=1)))
    att = lrelu(cv('sAtt_3', att, 1))
    att = F.interpolate(att, scale_factor=2, mode='bilinear', align_corners=False)
    att = cv('sAtt_4', att, 1)
    att_add = cv('sAtt_add_2', lrelu(cv('sAtt_add_1', att)))
    att = torch.sigmoid(att)
    return fea * att * 2 + att_add


def bfp_gather(inputs):
    # bfp_tcea.py:96-109, refine_level=0: all levels nearest-resized to level 0
    size = inputs[0].shape[2:]
    feats = [F.interpolate(t, size=size, mode='nearest') for t in inputs]
    return sum(feats) / len(feats)


def bfp_tcea(sd, p, inputs, ref_inputs, flow_init, return_aux=False):
    # bfp_tcea.py:111-149 (nframes=2, center=0, refine_type='conv')
    bsf = bfp_gather(inputs)
    ref_bsf = bfp_gather(ref_inputs)
    warp = warping_layer(ref_bsf, flow_init)
    flow_fine = liteflownet_corr(sd, p + 'liteflownet.', bsf, warp, flow_init)
    warp = warping_layer(warp, flow_fine)
    stack = torch.stack([bsf, warp], dim=1)
    fused = tcea_fusion(sd, p + 'tcea_fusion.', stack, center=0)
    ref = F.relu(F.conv2d(fused, sd[p + 'refine.conv.weight'], sd[p + 'refine.conv.bias'], padding=1))
    outs = []
    for i in range(len(inputs)):
        residual = F.adaptive_max_pool2d(ref, output_size=inputs[i].shape[2:])
        outs.append(residual + inputs[i])
    if return_aux:
        return outs, dict(bsf=bsf, ref_bsf=ref_bsf, flow_fine=flow_fine, warp=warp, fused=fused, refined=ref)
    return outs


# ------------------------------------------------------------------ semantic head (models/panoptic/upsnetFPN.py)
def upsnet_fpn(sd, p, inputs):
    # upsnetFPN.py:67-81; DeformConvWithOffset utils/deform_conv_with_offset.py:8-37; GroupNorm(32)+ReLU
    px = []
    for x in inputs:
        for conv_i, gn_i in ((0, 1), (3, 4), (6, 7)):
            q = '%sdeform_convs.0.%d' % (p, conv_i)
            off = F.conv2d(x, sd[q + '.conv_offset.weight'], sd[q + '.conv_offset.bias'], padding=1)
            x = ops.deform_conv(x, off, sd[q + '.conv.weight'], stride=1, padding=1)
            g = '%sdeform_convs.0.%d' % (p, gn_i)
            x = F.relu(F.group_norm(x, 32, sd[g + '.weight'], sd[g + '.bias'], 1e-5))
        px.append(x)
    feats = [px[0]] + [F.interpolate(px[i], None, 2 ** i, mode='bilinear', align_corners=False) for i in (1, 2, 3)]
    feat = torch.cat(feats, dim=1)
    fcn_score = F.conv2d(feat, sd[p + 'conv_pred.conv.weight'], sd[p + 'conv_pred.conv.bias'])
    fcn_output = F.interpolate(fcn_score, scale_factor=4, mode='bilinear', align_corners=False)
    return fcn_output, fcn_score


# ------------------------------------------------------------------ RPN (anchor_heads/rpn_head.py, core/anchor)
def gen_base_anchors(base_size, scales, ratios):
    # core/anchor/anchor_generator.py:18-46 (scale_major=True, ctr=None)
    scales = torch.Tensor(scales); ratios = torch.Tensor(ratios)
    w = h = base_size
    x_ctr = 0.5 * (w - 1); y_ctr = 0.5 * (h - 1)
    h_ratios = torch.sqrt(ratios); w_ratios = 1 / h_ratios
    ws = (w * w_ratios[:, None] * scales[None, :]).view(-1)
    hs = (h * h_ratios[:, None] * scales[None, :]).view(-1)
    return torch.stack([x_ctr - 0.5 * (ws - 1), y_ctr - 0.5 * (hs - 1), x_ctr + 0.5 * (ws - 1), y_ctr + 0.5 * (hs - 1)],
                       dim=-1).round()


def grid_anchors(base_anchors, featmap_size, stride):
    # anchor_generator.py:55-72
    feat_h, feat_w = featmap_size
    shift_x = torch.arange(0, feat_w) * stride
    shift_y = torch.arange(0, feat_h) * stride
    xx = shift_x.repeat(len(shift_y))
    yy = shift_y.view(-1, 1).repeat(1, len(shift_x)).view(-1)
    shifts = torch.stack([xx, yy, xx, yy], dim=-1).type_as(base_anchors)
    return (base_anchors[None, :, :] + shifts[:, None, :]).view(-1, 4)


def delta2bbox(rois, deltas, means=(0, 0, 0, 0), stds=(1, 1, 1, 1), max_shape=None, wh_ratio_clip=16 / 1000):
    # core/bbox/transforms.py:34-68
    means = deltas.new_tensor(means).repeat(1, deltas.size(1) // 4)
    stds = deltas.new_tensor(stds).repeat(1, deltas.size(1) // 4)
    d = deltas * stds + means
    dx, dy, dw, dh = d[:, 0::4], d[:, 1::4], d[:, 2::4], d[:, 3::4]
    max_ratio = np.abs(np.log(wh_ratio_clip))
    dw = dw.clamp(min=-max_ratio, max=max_ratio); dh = dh.clamp(min=-max_ratio, max=max_ratio)
    px = ((rois[:, 0] + rois[:, 2]) * 0.5).unsqueeze(1).expand_as(dx)
    py = ((rois[:, 1] + rois[:, 3]) * 0.5).unsqueeze(1).expand_as(dy)
    pw = (rois[:, 2] - rois[:, 0] + 1.0).unsqueeze(1).expand_as(dw)
    ph = (rois[:, 3] - rois[:, 1] + 1.0).unsqueeze(1).expand_as(dh)
    gw = pw * dw.exp(); gh = ph * dh.exp()
    gx = torch.addcmul(px, pw, dx); gy = torch.addcmul(py, ph, dy)
    x1 = gx - gw * 0.5 + 0.5; y1 = gy - gh * 0.5 + 0.5
    x2 = gx + gw * 0.5 - 0.5; y2 = gy + gh * 0.5 - 0.5
    if max_shape is not None:
        x1 = x1.clamp(min=0, max=max_shape[1] - 1); y1 = y1.clamp(min=0, max=max_shape[0] - 1)
        x2 = x2.clamp(min=0, max=max_shape[1] - 1); y2 = y2.clamp(min=0, max=max_shape[0] - 1)
    return torch.stack([x1, y1, x2, y2], dim=-1).view_as(deltas)


def rpn_forward(sd, p, feats):
    # rpn_head.py:30-35
    outs = []
    for x in feats:
        x = F.relu(F.conv2d(x, sd[p + 'rpn_conv.weight'], sd[p + 'rpn_conv.bias'], padding=1))
        outs.append((F.conv2d(x, sd[p + 'rpn_cls.weight'], sd[p + 'rpn_cls.bias']),
                     F.conv2d(x, sd[p + 'rpn_reg.weight'], sd[p + 'rpn_reg.bias'])))
    return outs


def rpn_get_bboxes(rpn_outs, img_shape, cfg=CFG):
    # anchor_head.py:198-223 + rpn_head.py:55-104 (use_sigmoid_cls, nms_across_levels=False)
    rc = cfg['rpn']
    mlvl = []
    for idx, (cls, reg) in enumerate(rpn_outs):
        cls = cls[0]; reg = reg[0]
        base = gen_base_anchors(cfg['anchor_strides'][idx], cfg['anchor_scales'], cfg['anchor_ratios'])
        anchors = grid_anchors(base, cls.shape[-2:], cfg['anchor_strides'][idx])
        scores = cls.permute(1, 2, 0).reshape(-1).sigmoid()
        reg = reg.permute(1, 2, 0).reshape(-1, 4)
        if rc['nms_pre'] > 0 and scores.shape[0] > rc['nms_pre']:
            _, topk = scores.topk(rc['nms_pre'])
            reg = reg[topk, :]; anchors = anchors[topk, :]; scores = scores[topk]
        props = delta2bbox(anchors, reg, (0., 0., 0., 0.), (1., 1., 1., 1.), img_shape)
        props = torch.cat([props, scores.unsqueeze(-1)], dim=-1)
        props, _ = ops.nms_mmdet(props, rc['nms_thr'])
        mlvl.append(props[:rc['nms_post'], :])
    props = torch.cat(mlvl, 0)
    num = min(rc['max_num'], props.shape[0])
    _, topk = props[:, 4].topk(num)
    return props[topk, :]


# ------------------------------------------------------------------ RoI extractor (roi_extractors/single_level.py)
def map_roi_levels(rois, num_levels, finest_scale=56):
    scale = torch.sqrt((rois[:, 3] - rois[:, 1] + 1) * (rois[:, 4] - rois[:, 2] + 1))
    lv = torch.floor(torch.log2(scale / finest_scale + 1e-6))
    return lv.clamp(min=0, max=num_levels - 1).long()


def roi_extract(feats, rois, out_size, cfg=CFG):
    strides = cfg['featmap_strides']
    feats = feats[:len(strides)]
    lv = map_roi_levels(rois, len(feats), cfg['finest_scale'])
    out = feats[0].new_zeros(rois.size(0), feats[0].shape[1], out_size, out_size)
    for i in range(len(feats)):
        inds = lv == i
        if inds.any():
            out[inds] = ops.roi_align(feats[i], rois[inds, :], out_size, 1.0 / strides[i], 2)
    return out


def bbox_head(sd, p, x):
    # bbox_heads/convfc_bbox_head.py:132-168 (SharedFCBBoxHead, 2 fcs)
    x = x.view(x.size(0), -1)
    for i in range(2):
        x = F.relu(F.linear(x, sd['%sshared_fcs.%d.weight' % (p, i)], sd['%sshared_fcs.%d.bias' % (p, i)]))
    return (F.linear(x, sd[p + 'fc_cls.weight'], sd[p + 'fc_cls.bias']),
            F.linear(x, sd[p + 'fc_reg.weight'], sd[p + 'fc_reg.bias']))


# ------------------------------------------------------------------ MaskROI (utils/mask_roi.py:37-147)
def _bbox_transform_np(boxes, deltas, weights):
    # utils/upsnet/bbox/bbox_transform.py:290-330
    if boxes.shape[0] == 0:
        return np.zeros((0, deltas.shape[1]), dtype=deltas.dtype)
    boxes = boxes.astype(deltas.dtype, copy=False)
    widths = boxes[:, 2] - boxes[:, 0] + 1.0
    heights = boxes[:, 3] - boxes[:, 1] + 1.0
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx; dy = deltas[:, 1::4] / wy
    dw = deltas[:, 2::4] / ww; dh = deltas[:, 3::4] / wh
    dw = np.minimum(dw, np.log(1000. / 16.)); dh = np.minimum(dh, np.log(1000. / 16.))
    pcx = dx * widths[:, np.newaxis] + ctr_x[:, np.newaxis]
    pcy = dy * heights[:, np.newaxis] + ctr_y[:, np.newaxis]
    pw = np.exp(dw) * widths[:, np.newaxis]; ph = np.exp(dh) * heights[:, np.newaxis]
    out = np.zeros(deltas.shape, dtype=deltas.dtype)
    out[:, 0::4] = pcx - 0.5 * pw; out[:, 1::4] = pcy - 0.5 * ph
    out[:, 2::4] = pcx + 0.5 * pw - 1; out[:, 3::4] = pcy + 0.5 * ph - 1
    return out


def _clip_boxes_np(boxes, im_shape):
    # bbox_transform.py:45-60
    boxes[:, 0::4] = np.maximum(np.minimum(boxes[:, 0::4], im_shape[1] - 1), 0)
    boxes[:, 1::4] = np.maximum(np.minimum(boxes[:, 1::4], im_shape[0] - 1), 0)
    boxes[:, 2::4] = np.maximum(np.minimum(boxes[:, 2::4], im_shape[1] - 1), 0)
    boxes[:, 3::4] = np.maximum(np.minimum(boxes[:, 3::4], im_shape[0] - 1), 0)
    return boxes


def mask_roi(rois, bbox_delta, cls_prob, im_info, cfg=CFG, num_classes=9):
    """class_agnostic=True, clip_boxes=True, top_n=100 -> (scores[K], rois[K,5], cls_idx[K] in 1..8);
    empty -> the dummy row (score 1, box 0, cls 0)."""
    rois_np = rois.detach().numpy()
    delta_np = bbox_delta.detach().numpy()
    prob_np = cls_prob.detach().numpy()
    proposal = _bbox_transform_np(rois_np[:, 1:], delta_np, cfg['bbox_reg_weights'])
    proposal = _clip_boxes_np(proposal, im_info[0, :2])
    n = proposal.shape[0]
    cls_idx = [[c for _ in range(n)] for c in range(num_classes)]
    prob_np = prob_np[:, 1:].reshape((-1, 1))
    prob_np = np.hstack((np.zeros_like(prob_np), prob_np))
    prob_t = cls_prob[:, 1:].contiguous().view(-1, 1)
    prob_t = torch.cat([torch.zeros_like(prob_t), prob_t], dim=1)
    proposal = proposal.reshape((n, -1, 4))[:, 1:, :].reshape((-1, 4))
    proposal = np.hstack((np.zeros_like(proposal), proposal))
    cls_idx = np.array(cls_idx).T[:, 1:].reshape((1, -1))
    cls_idx = np.vstack((np.zeros_like(cls_idx), cls_idx)).tolist()
    j = 1
    inds = np.where(prob_np[:, j] > cfg['mask_roi']['score_thresh'])[0]
    scores_j = prob_np[inds, j]
    boxes_j = proposal[inds, j * 4:(j + 1) * 4]
    dets_j = np.hstack((boxes_j, scores_j[:, np.newaxis])).astype(np.float32)
    keep = [] if len(dets_j) == 0 else ops.nms_upsnet(dets_j, cfg['mask_roi']['nms_thresh'])
    nms_dets = dets_j[keep, :]
    scores_th = prob_t[torch.from_numpy(inds).long(), j][torch.from_numpy(np.array(keep, dtype=np.int64)).long()]
    cls_j = np.array(cls_idx[j])[inds][keep]
    if cfg['max_det'] > 0 and len(nms_dets) > cfg['max_det']:
        image_thresh = np.sort(nms_dets[:, -1])[-cfg['max_det']]
        k2 = np.where(nms_dets[:, -1] >= image_thresh)[0]
        nms_dets = nms_dets[k2, :]; scores_th = scores_th[torch.from_numpy(k2)]; cls_j = cls_j[k2]
    boxes = np.zeros((nms_dets.shape[0], 5))
    boxes[:, 1:] = nms_dets[:, :-1]
    if nms_dets.shape[0] == 0:
        return torch.ones(1), torch.zeros(1, 5), torch.zeros(1, dtype=torch.long)
    return scores_th, torch.from_numpy(boxes).float(), torch.from_numpy(np.asarray(cls_j)).long()


# ------------------------------------------------------------------ track head (track_heads/track_head.py)
def track_embed(sd, p, x):
    # track_head.py:104-111 (fc, relu, fc — no relu after the last)
    x = x.view(x.size(0), -1)
    x = F.relu(F.linear(x, sd[p + 'fcs.0.weight'], sd[p + 'fcs.0.bias']))
    return F.linear(x, sd[p + 'fcs.1.weight'], sd[p + 'fcs.1.bias'])


def bbox_overlaps(b1, b2):
    # core/bbox/geometry.py:47-62
    lt = torch.max(b1[:, None, :2], b2[:, :2]); rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    overlap = wh[:, :, 0] * wh[:, :, 1]
    a1 = (b1[:, 2] - b1[:, 0] + 1) * (b1[:, 3] - b1[:, 1] + 1)
    a2 = (b2[:, 2] - b2[:, 0] + 1) * (b2[:, 3] - b2[:, 1] + 1)
    return overlap / (a1[:, None] + a2 - overlap)


def track_scores(sd, p, det_feats, prev_feats, cls_prob, det_bboxes, prev_bboxes, det_labels, prev_labels, cfg=CFG):
    # panoptic_fusetrack.py:412-422 + track_head.py:73-91,112-130
    x = track_embed(sd, p, det_feats); r = track_embed(sd, p, prev_feats)
    prod = torch.mm(x, r.t())
    match_score = torch.cat([torch.zeros(prod.size(0), 1), prod], dim=1)
    ll = F.log_softmax(match_score, dim=1)
    label_delta = (prev_labels == det_labels.view(-1, 1)).float()
    ious = bbox_overlaps(det_bboxes[:, :4], prev_bboxes[:, :4])
    ious = torch.cat((torch.zeros(ious.size(0), 1), ious), dim=1)
    label_delta = torch.cat((torch.ones(ious.size(0), 1), label_delta), dim=1)
    mc = cfg['match_coeff']
    return ll + mc[0] * torch.log(cls_prob.view(-1, 1)) + mc[1] * ious + mc[2] * label_delta


def greedy_assign(comp_scores, n_prev):
    """panoptic_fusetrack.py:424-469 host logic. Returns det_obj_ids and the list of memory updates
    [('add', det_idx) | ('set', obj_id, det_idx)] in the order the reference applies them."""
    ml, mi = torch.max(comp_scores, dim=1)
    ml = ml.numpy(); mi = mi.numpy().astype(np.int32)
    det_obj_ids = np.ones((mi.shape[0]), dtype=np.int32) * (-1)
    best_scores = np.ones((n_prev)) * (-100)
    best_ids = np.ones((n_prev), dtype=np.int32) * (-1)
    mem = n_prev
    updates = []
    for idx, match_id in enumerate(mi):
        if match_id == 0:
            det_obj_ids[idx] = mem; mem += 1; updates.append(('add', idx))
        else:
            obj_id = match_id - 1
            if ml[idx] > best_scores[obj_id]:
                det_obj_ids[idx] = obj_id
                if best_ids[obj_id] >= 0:
                    det_obj_ids[best_ids[obj_id]] = -1
                best_scores[obj_id] = ml[idx]; best_ids[obj_id] = idx
                updates.append(('set', obj_id, idx))
    for idx, oid in enumerate(det_obj_ids):
        if oid >= 0:
            continue
        det_obj_ids[idx] = mem; mem += 1; updates.append(('add', idx))
    return det_obj_ids, updates


# ------------------------------------------------------------------ mask head (mask_heads/fcn_mask_head.py:95-103)
def mask_head(sd, p, x):
    for i in range(4):
        x = F.relu(F.conv2d(x, sd['%sconvs.%d.conv.weight' % (p, i)], sd['%sconvs.%d.conv.bias' % (p, i)], padding=1))
    x = F.relu(F.conv_transpose2d(x, sd[p + 'upsample.weight'], sd[p + 'upsample.bias'], stride=2))
    return F.conv2d(x, sd[p + 'conv_logits.weight'], sd[p + 'conv_logits.bias'])


# ------------------------------------------------------------------ MaskRemoval (utils/mask_removal.py:29-92)
def mask_removal(mask_rois, cls_prob, mask_prob, cls_idx, im_shape, fraction_threshold=0.3):
    """Returns keep_inds (np.int64, original indices in score order) and mask_energy [1,k,H,W]."""
    mask_energy = mask_rois.new_zeros(1, mask_rois.size(0), im_shape[0], im_shape[1])
    frame_id = 0
    rois = mask_rois.detach().numpy()
    prob = cls_prob.detach().numpy()
    logit_all = mask_prob.detach().numpy()
    cls = cls_idx.detach().numpy()
    mask_image = np.zeros((np.max(cls),) + tuple(im_shape), dtype=np.uint8)
    sorted_inds = np.argsort(prob)[::-1]
    rois = rois[sorted_inds]; logit_all = logit_all[sorted_inds]; cls = cls[sorted_inds] - 1
    if len(cls) == 1 and cls[0] == -1:
        return np.array([0], dtype=np.int64), mask_prob.new_zeros(1, 1, im_shape[0], im_shape[1])
    keep_inds = []
    ref_boxes = rois.astype(np.int32)
    for i in range(sorted_inds.shape[0]):
        rb = ref_boxes[i, :].astype(np.int32)
        w = max(rb[2] - rb[0] + 1, 1); h = max(rb[3] - rb[1] + 1, 1)
        logit = ops.cv2_resize_linear(logit_all[i].squeeze(), (int(w), int(h)))
        mask = np.array(logit > 0, dtype=np.uint8)
        x_0 = max(rb[0], 0); x_1 = min(rb[2] + 1, im_shape[1])
        y_0 = max(rb[1], 0); y_1 = min(rb[3] + 1, im_shape[0])
        crop = mask[(y_0 - rb[1]):(y_1 - rb[1]), (x_0 - rb[0]):(x_1 - rb[0])]
        msum = crop.sum()
        mcrop = mask_image[cls[i]][y_0:y_1, x_0:x_1]
        if msum == 0 or (np.logical_and(mcrop >= 1, crop == 1).sum() / msum > fraction_threshold):
            continue
        keep_inds.append(sorted_inds[i])
        mask_image[cls[i]][y_0:y_1, x_0:x_1] += crop
        mask_energy[0, frame_id, y_0:y_1, x_0:x_1] = torch.from_numpy(
            logit[(y_0 - rb[1]):(y_1 - rb[1]), (x_0 - rb[0]):(x_1 - rb[0])])
        frame_id += 1
    mask_energy = mask_energy[:, :len(keep_inds)]
    if len(keep_inds) == 0:
        return np.array([0], dtype=np.int64), mask_prob.new_zeros(1, 1, im_shape[0], im_shape[1])
    return np.array(keep_inds), mask_energy


def seg_term(cls_indices, seg_score, boxes, cfg=CFG):
    # utils/unary_logits.py:81-108 (boxes arrive as mask_rois*4.0, box_scale 1/4)
    cls_np = cls_indices.numpy()
    stuff = seg_score[[0], :cfg['num_stuff'], :, :]
    b = boxes.numpy()[:, 1:] * (1 / 4.0)
    inst = torch.zeros((1, cls_np.shape[0], seg_score.shape[2], seg_score.shape[3]))
    for i in range(cls_np.shape[0]):
        if cls_np[i] == 0:
            continue
        y0 = int(b[i][1]); y1 = int(b[i][3].round() + 1)
        x0 = int(b[i][0]); x1 = int(b[i][2].round() + 1)
        inst[0, i, y0:y1, x0:x1] = seg_score[0, cfg['class_mapping'][int(cls_np[i])], y0:y1, x0:x1]
    return stuff, inst


# ------------------------------------------------------------------ the detector
class FuseTrackOracle:
    """PanopticFuseTrack.simple_test restated (panoptic_fusetrack.py:502-606). `sd` uses the reference keys:
    backbone.*, neck.*, extra_neck.*, panopticFPN.*, rpn_head.*, bbox_head.*, track_head.*, mask_head.*, flownet2.*"""

    def __init__(self, sd, cfg=CFG, depth=50, with_fusion=True, with_track=True):
        """with_track=False: PanopticFuse (panoptic_fuse.py:399-472, no ids); with_fusion=False: PanopticTrack
        (panoptic_track.py:443-536, the FPN outputs feed the heads directly)."""
        self.sd = {k: v.float() for k, v in sd.items()}
        self.cfg = cfg
        self.depth = depth
        self.with_fusion, self.with_track = with_fusion, with_track
        self.prev_bboxes = self.prev_roi_feats = self.prev_det_labels = None

    def compute_flow(self, img, ref_img, scale_factor=0.25):
        # panoptic_fusetrack.py:117-143 + utils/flow_utils.py:5-10
        def denorm(t):
            t = t.clone()
            for c in range(3):
                t[:, c] = t[:, c] * self.cfg['std'][c] + self.cfg['mean'][c]
            return t
        rgbs = torch.stack([denorm(img), denorm(ref_img)], dim=2)
        H, W = rgbs.size(-2), rgbs.size(-1)
        if H == 800 and W == 1600:                       # :125-128 "Pad zeros"
            rgbs = F.pad(rgbs, (0, 64, 0, 32))
        elif H == 200 and W == 400:
            rgbs = F.pad(rgbs, (0, 48, 0, 56))
        assert rgbs.size(-2) % 64 == 0 and rgbs.size(-1) % 64 == 0
        flow = flownet2(self.sd, 'flownet2.', rgbs)
        flow = flow[:, :, :H, :W]                        # :135-138 "Trim zeros" (index_select of arange(H), arange(W))
        self.last_flow_full = flow
        return F.interpolate(flow, scale_factor=scale_factor, mode='bilinear', align_corners=False) * scale_factor

    def extract_feat(self, img):
        return fpn(self.sd, 'neck.', resnet(self.sd, 'backbone.', img, self.depth))

    def detect(self, x, im_shape, is_first, inject=None):
        """steps (5)-(6) of SURVEY §3.2; `inject` may override cls_score/bbox_pred/proposals for synthetic heads."""
        sd, cfg = self.sd, self.cfg
        im_info = np.array([[float(im_shape[0]), float(im_shape[1]), 1.0]])
        if inject is not None and 'proposals' in inject:
            proposals = inject['proposals']
        else:
            proposals = rpn_get_bboxes(rpn_forward(sd, 'rpn_head.', x), im_shape, cfg)
        rois = torch.cat([proposals.new_zeros(proposals.size(0), 1), proposals[:, :4]], dim=-1)
        roi_feats = roi_extract(x, rois, 7, cfg)
        cls_score, bbox_pred = bbox_head(sd, 'bbox_head.', roi_feats)
        if inject is not None and 'cls_score' in inject:
            cls_score, bbox_pred = inject['cls_score'], inject['bbox_pred']
        cls_prob = F.softmax(cls_score, dim=1)
        cls_prob, det_rois, cls_idx = mask_roi(rois, bbox_pred, cls_prob, im_info, cfg)
        det_labels = cls_idx - 1
        det_roi_feats = roi_extract(x, det_rois, 7, cfg)
        det_bboxes = det_rois[:, 1:]
        if not self.with_track:
            return dict(proposals=proposals, cls_score=cls_score, bbox_pred=bbox_pred, cls_prob=cls_prob, det_rois=det_rois,
                        cls_idx=cls_idx, det_labels=det_labels, det_obj_ids=np.full((det_bboxes.size(0),), -1), comp_scores=None,
                        det_roi_feats=det_roi_feats)
        if is_first or self.prev_bboxes is None:
            det_obj_ids = np.arange(det_bboxes.size(0))
            self.prev_bboxes = det_bboxes.clone(); self.prev_roi_feats = det_roi_feats.clone()
            self.prev_det_labels = det_labels.clone()
            comp = None
        else:
            comp = track_scores(sd, 'track_head.', det_roi_feats, self.prev_roi_feats, cls_prob, det_bboxes,
                                self.prev_bboxes, det_labels, self.prev_det_labels, cfg)
            det_obj_ids, updates = greedy_assign(comp, self.prev_bboxes.size(0))
            for u in updates:
                if u[0] == 'add':
                    i = u[1]
                    self.prev_roi_feats = torch.cat((self.prev_roi_feats, det_roi_feats[i][None]), dim=0)
                    self.prev_bboxes = torch.cat((self.prev_bboxes, det_bboxes[i][None]), dim=0)
                    self.prev_det_labels = torch.cat((self.prev_det_labels, det_labels[i][None]), dim=0)
                else:
                    _, oid, i = u
                    self.prev_roi_feats[oid] = det_roi_feats[i]; self.prev_bboxes[oid] = det_bboxes[i]
        return dict(proposals=proposals, cls_score=cls_score, bbox_pred=bbox_pred, cls_prob=cls_prob, det_rois=det_rois,
                    cls_idx=cls_idx, det_labels=det_labels, det_obj_ids=np.asarray(det_obj_ids), comp_scores=comp,
                    det_roi_feats=det_roi_feats)

    def panoptic(self, x, fcn_output, det, inject=None):
        """steps (8)-(11)."""
        sd, cfg = self.sd, self.cfg
        mask_rois, cls_idx, cls_prob = det['det_rois'], det['cls_idx'], det['cls_prob']
        mask_feats = roi_extract(x, mask_rois, 14, cfg)
        mask_score = mask_head(sd, 'mask_head.', mask_feats)
        if inject is not None and 'mask_score' in inject:
            mask_score = inject['mask_score'][:mask_rois.size(0)]          # a bank of >= K rows
        nobj, _, H, W = mask_score.shape
        mask_score = mask_score.gather(1, cls_idx.view(-1, 1, 1, 1).expand(-1, -1, H, W))
        keep_inds, mask_logits = mask_removal(mask_rois[:, 1:], cls_prob, mask_score, cls_idx, tuple(fcn_output.shape[2:]),
                                              cfg['mask_removal_fraction'])
        keep_t = torch.from_numpy(np.asarray(keep_inds, dtype=np.int64))
        det_obj_ids = torch.from_numpy(np.asarray(det['det_obj_ids']).astype(np.int64))
        mask_rois = mask_rois[keep_t]; cls_idx = cls_idx[keep_t]
        det_labels = det['det_labels'][keep_t]; det_obj_ids = det_obj_ids[keep_t]; cls_prob = cls_prob[keep_t]
        stuff, inst = seg_term(cls_idx, fcn_output, mask_rois * 4.0, cfg)
        pan_logits = torch.cat([stuff, inst + mask_logits], dim=1)
        pan_out = torch.max(F.softmax(pan_logits, dim=1), dim=1)[1]
        sem_out = torch.max(F.softmax(fcn_output, dim=1), dim=1)[1]
        return dict(fcn_outputs=sem_out, panoptic_outputs=pan_out, panoptic_cls_inds=cls_idx, panoptic_cls_prob=cls_prob,
                    panoptic_det_labels=det_labels, panoptic_det_obj_ids=det_obj_ids, keep_inds=np.asarray(keep_inds),
                    mask_score=mask_score)

    def simple_test(self, img, ref_img, is_first, ref_x=None, inject=None, return_aux=False):
        x = self.extract_feat(img)
        pre_neck = x
        flow = None
        if self.with_fusion:
            flow = self.compute_flow(img, ref_img, 0.25)
            if ref_x is None:
                ref_x = self.extract_feat(ref_img)
            x = bfp_tcea(self.sd, 'extra_neck.', x, ref_x, flow)
        fcn_output, fcn_score = upsnet_fpn(self.sd, 'panopticFPN.', x[0:4])
        if inject is not None and 'fcn_score' in inject:
            fcn_score = inject['fcn_score']
            fcn_output = F.interpolate(fcn_score, scale_factor=4, mode='bilinear', align_corners=False)
        det = self.detect(x, tuple(img.shape[2:]), is_first, inject)
        pano = self.panoptic(x, fcn_output, det, inject)
        if not self.with_track:
            pano.pop('panoptic_det_labels'); pano.pop('panoptic_det_obj_ids')
        if return_aux:
            pano.update(flow=flow, flow_full=self.last_flow_full if self.with_fusion else None, feats=x, pre_neck=pre_neck,
                        fcn_score=fcn_score, det=det)
        return pano
