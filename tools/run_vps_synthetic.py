"""End-to-end run of the whole drop-in on a synthetic video set — the flow of the reference's tools/test_vpq.py:93-198 followed by
tools/eval_vpq.py:261-330, with every device-side stage of this package in place of the reference's:

    decoded uint8 frames -> DeviceImagePrep / PairFeeder (Normalize, Pad, ImageToTensor; ref_img = previous frame, prepared once)
    -> build_detector(cfg.model) + synthetic checkpoint -> model(return_loss=False, rescale=True, img=[..], img_meta=[..], ref_img=[..])
    -> the pano_results bookkeeping of single_gpu_test (test_vpq.py:28-69), maps kept on the device
    -> PanopticUnifier.get_unified_pan_result (cityscapes_vps.py:162-226)
    -> inference_panoptic_video: labelled-frame sampling, 2-channel -> colour PNG + segments_info, pan_2ch/ pan_pred/ pred.json
    -> vpq_compute_single_core for the window lengths of eval_vpq.py (k = 0, 5, 10, 15 <-> nframes 1..4) against a ground truth

There is no dataset offline, so the "ground truth" is the prediction itself (VPQ must come out as exactly 100 for every class
present), or the prediction of a second run in another arithmetic mode (`--gt-prec`): the VPQ between two fp32-grade modes is
the end-to-end parity number in the reference's own metric.

    python tools/run_vps_synthetic.py --videos 2 --frames 30 --height 256 --width 512 --out gpurun_out/vps_synth [--prec f16x3 --gt-prec f32]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CATEGORIES = [{'id': c, 'name': 'class%d' % c, 'isthing': 1 if c >= 11 else 0, 'color': [(37 * c) % 256, (91 * c) % 256, (53 * c + 80) % 256]}
              for c in range(19)]


class ColorGenerator:
    """stand-in for panopticapi.utils.IdGenerator (absent offline): distinct colours per call, deterministic"""

    def __init__(self, categories):
        self.categories = categories
        self.taken = set([0])

    def get_color(self, cat_id):
        base = self.categories[cat_id]['color']
        k = 0
        while True:
            c = [(base[0] + 7 * k) % 256, (base[1] + 13 * k) % 256, (base[2] + 29 * k) % 256]
            key = c[0] + 256 * c[1] + 65536 * c[2]
            if key not in self.taken:
                self.taken.add(key)
                return c
            k += 1


def uint8_frame(H, W, seed, shift):
    from vps_amd import synth
    return synth.synth_frame(H, W, seed=seed, shift=shift, noise=2.0 if shift != (0, 0) else 0.0).astype(np.uint8)


def run_model(prec, videos, nframes, H, W, dev, separated=False):
    """test_vpq.py:129-149 + single_gpu_test (:28-69) on `videos` synthetic clips -> pano_results with DEVICE maps"""
    import vps_amd
    from vps_amd import nhwc, synth
    from vps_amd.pipeline import DeviceImagePrep, PairFeeder
    old = nhwc.DEFAULT_PREC
    nhwc.DEFAULT_PREC = nhwc.PREC_NAMES[prec]
    try:
        cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'cityscapes', 'fusetrack.py'))
        model = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
        over = synth.separated_overrides(os.path.join(ROOT, 'tests', 'golden', 'separated_fc_cls.npz')) if separated else None
        synth.load_synth(model, 0, overrides=over)
        model.ensure_packed(dev)
    finally:
        nhwc.DEFAULT_PREC = old
    prep = DeviceImagePrep(**cfg.img_norm_cfg, size_divisor=32, img_scale=(max(H, W), min(H, W)), device=dev)
    feed = PairFeeder(prep)
    res = dict(all_names=[], all_ssegs=[], all_panos=[], all_pano_cls_inds=[], all_pano_obj_ids=[])
    decoded = [[uint8_frame(H, W, seed=v, shift=(2 * f, f)) for f in range(nframes)] for v in range(videos)]   # "cv2.imread" results (BGR uint8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for v in range(videos):
        feed.reset()
        for f in range(nframes):
            img, ref = feed(decoded[v][f])                                         # 6 MB upload + Normalize / Pad / ToTensor on the device
            name = '%04d_%04d_city_%06d_%06d_newImg8bit.png' % (v, f, v, f)
            meta = dict(filename=name, iid=v * 10000 + f + 1, img_shape=(H, W, 3), ori_shape=(H, W, 3), pad_shape=tuple(img.shape[2:]) + (3,),
                        scale_factor=1.0, flip=False)
            with torch.no_grad():
                result = model(return_loss=False, rescale=True, img=[img], img_meta=[[meta]], ref_img=[ref])
            res['all_ssegs'].append(result[2]['fcn_outputs'][0])                   # device uint8 maps (test_vpq.py:51-56 moves them to the host)
            res['all_panos'].append(result[2]['panoptic_outputs'][0])
            res['all_pano_cls_inds'].append(result[2]['panoptic_cls_inds'].cpu().numpy())
            res['all_pano_obj_ids'].append(result[2]['panoptic_det_obj_ids'].cpu().numpy())
            res['all_names'].append(name)
    torch.cuda.synchronize()
    return res, time.perf_counter() - t0


def postprocess(res, out_dir, videos, dev, labeled_fid, lambda_, nper):
    """test_vpq.py:178-198 with the device-side unifier / converter / asynchronous writer"""
    from vps_amd.postprocess import PanopticUnifier, inference_panoptic_video
    unifier = PanopticUnifier(dev, 19, 9)
    two = unifier.get_unified_pan_result(res['all_ssegs'], res['all_panos'], res['all_pano_cls_inds'], obj_ids=res['all_pano_obj_ids'],
                                         stuff_area_limit=2048, names=res['all_names'])
    keys = sorted(two.keys())
    pred_pans_2ch = [two[k] for k in keys]
    names = keys[(labeled_fid // lambda_)::lambda_]                               # names of the labelled frames (im_jsons['images'])
    cats = {c['id']: c for c in CATEGORIES}
    pans, pj = inference_panoptic_video(pred_pans_2ch, out_dir, CATEGORIES, names, n_video=videos, color_generator=ColorGenerator(cats), device=dev,
                                        labeled_fid=labeled_fid, lambda_=lambda_, nframes_per_video=nper)
    return names, pans, pj


def vpq(gt, pred, videos, nper, dev):
    """eval_vpq.py:261-330: per video, window lengths 1..4, PQ averaged over the categories with support; VPQ = mean over k"""
    from vps_amd.evaluate import vpq_compute_single_core
    cats = {c['id']: c for c in CATEGORIES}
    out = {}
    for nf in (1, 2, 3, 4):
        from vps_amd.evaluate import PQStat
        stat = PQStat()
        for v in range(videos):
            sl = slice(v * nper, (v + 1) * nper)
            clip = [(g, p, gp, pp, {}) for g, p, gp, pp in zip(gt[1]['annotations'][sl], pred[1]['annotations'][sl], gt[0][sl], pred[0][sl])]
            stat += vpq_compute_single_core(clip, cats, nframes=nf, device=dev)
        res, _ = stat.pq_average(cats, isthing=None)
        out[nf] = res
    out['vpq'] = float(np.mean([100 * out[nf]['pq'] for nf in (1, 2, 3, 4)]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--videos', type=int, default=2)
    ap.add_argument('--frames', type=int, default=30)
    ap.add_argument('--height', type=int, default=256)
    ap.add_argument('--width', type=int, default=512)
    ap.add_argument('--prec', default='f16x3')
    ap.add_argument('--gt-prec', default=None, help='take the "ground truth" from a second run in this arithmetic mode (default: the prediction itself)')
    ap.add_argument('--separated', action='store_true', help='box classification layer of tests/golden/separated_fc_cls.npz (few, well-separated detections)')
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'vps_synth'))
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    labeled_fid, lambda_ = 20, 5
    nper = len(range(labeled_fid // lambda_, args.frames, lambda_))
    res, dt = run_model(args.prec, args.videos, args.frames, args.height, args.width, dev, args.separated)
    t0 = time.perf_counter()
    names, pans, pj = postprocess(res, os.path.join(args.out, 'pred'), args.videos, dev, labeled_fid, lambda_, nper)
    dpost = time.perf_counter() - t0
    pred = (pans, pj)
    gt = pred
    if args.gt_prec:
        res2, _ = run_model(args.gt_prec, args.videos, args.frames, args.height, args.width, dev, args.separated)
        _, pans2, pj2 = postprocess(res2, os.path.join(args.out, 'gt'), args.videos, dev, labeled_fid, lambda_, nper)
        gt = (pans2, pj2)
    t0 = time.perf_counter()
    score = vpq(gt, pred, args.videos, nper, dev)
    deval = time.perf_counter() - t0
    files = sorted(os.listdir(os.path.join(args.out, 'pred', 'pan_pred')))
    report = dict(videos=args.videos, frames_per_video=args.frames, size=[args.height, args.width], prec=args.prec, gt=args.gt_prec or 'self', weights='synthetic seed 0' + (' + separated fc_cls' if args.separated else ''),
                  labelled_frames=len(names), png_files=len(files), vpq=round(score['vpq'], 4),
                  pq_per_window={str(k): round(100 * score[k]['pq'], 4) for k in (1, 2, 3, 4)},
                  seconds=dict(model=round(dt, 3), postprocess_and_png=round(dpost, 3), eval=round(deval, 3)),
                  frames_per_s_upload_prep_model_sequential=round(args.videos * args.frames / dt, 2))
    with open(os.path.join(args.out, 'report.json'), 'w') as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))
    return report


if __name__ == '__main__':
    main()
