"""Golden vectors of the reference's head-logic EDGE paths (VERDICT r2 "Missing" #3): the REAL `MaskROI.forward`
(mask_roi.py:37-147: score threshold, class-agnostic NMS, `max_det` cap with ties, dummy row), the REAL tracking block of
`PanopticFuseTrack.simple_test_bboxes` (panoptic_fusetrack.py:400-469: `TrackHead.forward` / `compute_comp_scores`, greedy
assignment with the undo branch, memory growth), the REAL `MaskRemoval.forward` (mask_removal.py:29-92: overlap rule,
keep-nothing path), `SegTerm.forward` (unary_logits.py:81-108: `cls == 0` skip) and the logit concat / arg-max of
`simple_test` (:585-597), run at 1024x2048 on the nine injected cases of tests/inject_cases.py.

How: the real detector is built as in make_golden.py (import shims of ref_shims.py); the modules in FRONT of the head logic are
replaced by the injected tensors — `extra_neck` returns `inject_cases.neck_features()`, `panopticFPN` the injected
`fcn_score` (+ its x4 upsampling, upsnetFPN.py:77-80), `simple_test_rpn` the injected proposals, `bbox_head` the injected
`cls_score` / `bbox_pred`, `mask_head` the injected mask logits — and `ref.simple_test` itself runs, untouched, from
`simple_test_bboxes` to the returned dicts.

    python tests/golden/make_inject_golden.py        # needs /root/reference; writes tests/golden/inject_cases.npz (~2 MB)

Stored per case / frame: MaskROI outputs, the comprehensive track scores, ids, kept list, the returned per-instance vectors,
the two maps at stride 4 plus their label histograms over the full frame (the random semantic logits make the maps
incompressible: 2 MB each in full).
"""
import os
import sys
import tempfile
import warnings

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
warnings.simplefilter('ignore')


def build_reference():
    import ref_shims
    ref_shims.install()
    import vps_amd
    from vps_amd import synth
    from vps_amd.registry import Config, ConfigDict
    cfg = Config.fromfile('/root/reference/configs/cityscapes/fusetrack.py')
    ours = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in ours.state_dict().items()}, 0)
    tmp = tempfile.mkdtemp(prefix='vps_golden_')
    os.makedirs(os.path.join(tmp, 'work_dirs', 'flownet'))
    torch.save({'state_dict': {k[len('flownet2.'):]: v for k, v in sd.items() if k.startswith('flownet2.')}},
               os.path.join(tmp, 'work_dirs', 'flownet', 'FlowNet2_checkpoint.pth.tar'))
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        builder = sys.modules['mmdet.models.builder']
        model_cfg = ConfigDict.wrap(dict(cfg.model))
        model_cfg['pretrained'] = None
        ref = builder.build_detector(model_cfg, train_cfg=None, test_cfg=cfg.test_cfg)
    finally:
        os.chdir(cwd)
    ref.eval()
    ref.load_state_dict(sd)
    return ref


def main():
    import inject_cases as IC
    from vps_amd import synth
    ref = build_reference()
    H, W = IC.H, IC.W
    x = IC.neck_features()
    cur = {}
    cap = {}

    # --- the stages in FRONT of the head logic hand over the injected tensors --------------------------------------------
    ref.compute_flow = lambda *a, **k: (None, None)
    ref.extract_feat = lambda img: None
    ref.extra_neck.forward = lambda feats, ref_feats, flow: tuple(x)
    ref.panopticFPN.forward = lambda feats: (F.interpolate(cur['fcn_score'], scale_factor=4, mode='bilinear', align_corners=False),
                                             cur['fcn_score'])
    ref.simple_test_rpn = lambda feats, meta, cfg: [cur['proposals']]
    ref.bbox_head.forward = lambda roi_feats: (cur['cls_score'], cur['bbox_pred'])
    ref.mask_head.forward = lambda feats: cur['mask_score'][:feats.size(0)]

    # --- taps on the REAL functions (outputs only) --------------------------------------------------------------------------
    real_mask_roi = ref.mask_roi_panoptic.forward

    def tap_mask_roi(*a, **k):
        r = real_mask_roi(*a, **k)
        cap['mask_roi'] = [t.clone() for t in r]
        return r
    ref.mask_roi_panoptic.forward = tap_mask_roi
    real_comp = ref.track_head.compute_comp_scores

    def tap_comp(*a, **k):
        r = real_comp(*a, **k)
        cap['comp_scores'] = r.clone()
        return r
    ref.track_head.compute_comp_scores = tap_comp
    real_removal = ref.mask_removal.forward

    def tap_removal(*a, **k):
        r = real_removal(*a, **k)
        cap['keep_inds'] = r[0].clone()
        cap['mask_energy_nonzero'] = bool((r[1] != 0).any())
        return r
    ref.mask_removal.forward = tap_removal

    out = {}
    img = torch.zeros(1, 3, H, W)
    for case in IC.CASES:
        ref.prev_bboxes = ref.prev_roi_feats = ref.prev_det_labels = None
        for t, inj in enumerate(IC.frames_of(case)):
            cur.clear(); cur.update(IC.public(inj)); cap.clear()
            meta = synth.img_meta(H, W, 10000 + t + 1)
            M_before = 0 if ref.prev_bboxes is None else int(ref.prev_bboxes.size(0))
            with torch.no_grad():
                bbox_res, mask_res, pano = ref(return_loss=False, rescale=True, img=[img], img_meta=[[meta]], ref_img=[img])
            p = '%s.f%d.' % (case, t)
            out[p + 'mask_roi_scores'] = cap['mask_roi'][0].numpy()
            out[p + 'mask_roi_rois'] = cap['mask_roi'][1].numpy()
            out[p + 'mask_roi_cls_idx'] = cap['mask_roi'][2].numpy()
            out[p + 'M_before'] = np.array(M_before)
            if 'comp_scores' in cap:
                out[p + 'comp_scores'] = cap['comp_scores'].numpy()
            out[p + 'keep_inds'] = cap['keep_inds'].numpy()
            out[p + 'mask_energy_nonzero'] = np.array(cap['mask_energy_nonzero'])
            out[p + 'bbox_ids'] = np.array(sorted(int(k) for k in bbox_res.keys()), dtype=np.int64)
            for k in ('panoptic_cls_inds', 'panoptic_cls_prob', 'panoptic_det_labels', 'panoptic_det_obj_ids'):
                out[p + k] = np.asarray(pano[k].numpy())
            for k in ('panoptic_outputs', 'fcn_outputs'):
                m = pano[k].numpy().astype(np.uint8)[0]
                assert pano[k].max() <= 255
                out[p + k + '_s4'] = m[::4, ::4].copy()
                out[p + k + '_hist'] = np.bincount(m.reshape(-1), minlength=256).astype(np.int64)
            print('%s frame %d: K=%d M=%d kept=%d ids=%s' % (case, t, cap['mask_roi'][2].numel(), M_before, cap['keep_inds'].numel(),
                                                            out[p + 'panoptic_det_obj_ids'][:10]), flush=True)
    path = os.path.join(HERE, 'inject_cases.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) / 1e6, 'MB')


if __name__ == '__main__':
    main()
