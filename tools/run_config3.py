"""BASELINE config 3 in one command: the released checkpoint + the Cityscapes-VPS val videos through the whole drop-in, the flow of
tools/test_vpq.py:93-198 (then tools/eval_vpq.py when the ground truth is given) with this package in the reference's place:

    python tools/run_config3.py --config configs/cityscapes/fusetrack.py \\
        --checkpoint work_dirs/cityscapes_vps/fusetrack_vpct/latest.pth --flownet-checkpoint work_dirs/flownet/FlowNet2_checkpoint.pth.tar \\
        --data-root data/cityscapes_vps --out work_dirs/cityscapes_vps/fusetrack_vpct/val.pkl [--truth-dir data/cityscapes_vps/val/panoptic_video]

    file / directory                                   reference                                  read by
    <data-root>/im_all_info_val_city_vps.json           configs/cityscapes/fusetrack.py:216-222    frame list, ids (iid = video*10000 + frame + 1)
    <data-root>/val/img_all/<file_name>                 img_prefix == ref_prefix                   vps_amd.pipeline.ClipFeeder (every file decoded once)
    <data-root>/panoptic_im_val_city_vps.json           test_vpq.py:86,165-169                     labelled-frame names + categories
    <checkpoint>  ('state_dict' / 'meta', 'module.')    test_vpq.py:135-143                        vps_amd.load_checkpoint (mmcv semantics)
    <flownet-checkpoint> ('state_dict')                 panoptic_fusetrack.py:100-106              detector constructor
    <truth-dir>, <data-root>/panoptic_gt_val_city_vps.json   eval_vpq.py:251-330                   tools/eval_vpq_device.py (vpq-*.txt)

Outputs (test_vpq.py:150-198): <out>_pans_unified/{pan_2ch,pan_pred}/*.png + pred.json, and the VPQ tables when --truth-dir is given.
The artefacts do not exist offline (no weights, no dataset). `--dry-run` writes a synthetic checkpoint pair IN THE
REFERENCE'S FILE LAYOUT ('module.'-prefixed state_dict + meta; FlowNet2 'state_dict') and a synthetic dataset tree (PNG frames, the
two json files, ground truth = the run's own prediction) into a scratch directory and runs the same code on it;
`--check-only` stops after the model is built, the checkpoints are loaded (missing / unexpected keys reported) and the dataset is
listed - no GPU needed. One JSON report line on stdout.
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
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description='VPSNet FuseTrack test on Cityscapes-VPS (vps_amd)')
    ap.add_argument('--config', default=os.path.join(ROOT, 'configs', 'cityscapes', 'fusetrack.py'))
    ap.add_argument('--checkpoint', default='work_dirs/cityscapes_vps/fusetrack_vpct/latest.pth')
    ap.add_argument('--flownet-checkpoint', default='work_dirs/flownet/FlowNet2_checkpoint.pth.tar')
    ap.add_argument('--data-root', default='data/cityscapes_vps')
    ap.add_argument('--out', default='work_dirs/cityscapes_vps/fusetrack_vpct/val.pkl', help='like test_vpq.py --out: <out minus .pkl>_pans_unified/ receives the results')
    ap.add_argument('--mode', default='val', choices=['val', 'test'])
    ap.add_argument('--n-video', type=int, default=50)
    ap.add_argument('--nframes-span-test', type=int, default=30)
    ap.add_argument('--stuff-area-limit', type=int, default=2048, help='configs/cityscapes/test_cityscapes_1gpu.yaml:29')
    ap.add_argument('--truth-dir', default=None, help='ground-truth PNG directory: evaluate with tools/eval_vpq_device.py afterwards')
    ap.add_argument('--pan-gt-json', default=None)
    ap.add_argument('--prec', default='f16x3', choices=['f32', 'bf16x6', 'f16x3'])
    ap.add_argument('--strict', action='store_true', help='load_checkpoint(strict=True): missing / unexpected keys are fatal')
    ap.add_argument('--png-workers', type=int, default=6)
    ap.add_argument('--dry-run', action='store_true')
    ap.add_argument('--dry-size', default='128x256'); ap.add_argument('--dry-videos', type=int, default=1); ap.add_argument('--dry-frames', type=int, default=16)
    ap.add_argument('--check-only', action='store_true')
    return ap.parse_args(argv)


def make_dry_run_artefacts(args, tmp):
    """synthetic artefacts in the layout of download_weights.sh / the dataset README: returns the patched args"""
    import vps_amd
    from PIL import Image
    from vps_amd import synth
    H, W = [int(v) for v in args.dry_size.split('x')]
    cfg = vps_amd.Config.fromfile(args.config)
    model = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    sd = synth.synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, 0)
    os.makedirs(os.path.join(tmp, 'work_dirs', 'flownet'), exist_ok=True)
    os.makedirs(os.path.join(tmp, 'work_dirs', 'cityscapes_vps', 'fusetrack_vpct'), exist_ok=True)
    fn = {k[len('flownet2.'):]: v for k, v in sd.items() if k.startswith('flownet2.')}
    args.flownet_checkpoint = os.path.join(tmp, 'work_dirs', 'flownet', 'FlowNet2_checkpoint.pth.tar')
    torch.save({'epoch': 0, 'state_dict': fn}, args.flownet_checkpoint)
    # latest.pth as mmcv's save_checkpoint writes it from a MMDistributedDataParallel model: 'module.' prefix, 'meta', an optimizer blob
    from collections import OrderedDict
    args.checkpoint = os.path.join(tmp, 'work_dirs', 'cityscapes_vps', 'fusetrack_vpct', 'latest.pth')
    torch.save({'meta': {'epoch': 12, 'CLASSES': ('person', 'rider', 'car', 'truck', 'bus', 'train', 'motorcycle', 'bicycle')},
                'state_dict': OrderedDict(('module.' + k, v) for k, v in sd.items()), 'optimizer': {}}, args.checkpoint)
    root = os.path.join(tmp, 'data', 'cityscapes_vps')
    img_dir = os.path.join(root, args.mode, 'img_all')
    os.makedirs(img_dir, exist_ok=True)
    images, labelled = [], []
    for v in range(args.dry_videos):
        for f in range(args.dry_frames):
            name = '%04d_%04d_frankfurt_%06d_%06d_newImg8bit.png' % (v, f, v, f)
            fr = synth.synth_frame(H, W, seed=v, shift=(2 * f, f), noise=2.0 if f else 0.0).astype(np.uint8)       # BGR
            Image.fromarray(np.ascontiguousarray(fr[:, :, ::-1])).save(os.path.join(img_dir, name))
            images.append(dict(id=v * 10000 + f + 1, file_name=name, height=H, width=W))
            if f % 5 == args.dry_label_first % 5 and f >= args.dry_label_first:
                # panoptic_im_*.json: id = the frame's base name (eval_vpq.py:280 opens pan_pred/<id>.png), file_name = the image name
                labelled.append(dict(id=name.replace('_newImg8bit.png', ''), file_name=name, height=H, width=W))
    cats = [{'id': c, 'name': 'class%d' % c, 'isthing': 1 if c >= 11 else 0, 'color': [(37 * c) % 256, (91 * c) % 256, (53 * c + 80) % 256]} for c in range(19)]
    json.dump(dict(images=images, categories=cats), open(os.path.join(root, 'im_all_info_%s_city_vps.json' % args.mode), 'w'))
    json.dump(dict(images=labelled, categories=cats), open(os.path.join(root, 'panoptic_im_%s_city_vps.json' % args.mode), 'w'))
    args.data_root = root
    args.out = os.path.join(tmp, 'work_dirs', 'cityscapes_vps', 'fusetrack_vpct', args.mode + '.pkl')
    args.n_video, args.nframes_span_test = args.dry_videos, args.dry_frames
    return args


def main(argv=None):
    args = parse_args(argv)
    args.dry_label_first = 0
    tmp = None
    if args.dry_run:
        import tempfile
        tmp = tempfile.mkdtemp(prefix='vps_config3_')
    try:
        return run(args, tmp)
    finally:
        if tmp and not os.environ.get('VPS_KEEP_DRY_RUN'):
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)


def run(args, tmp):
    report = dict(config=args.config, dry_run=bool(args.dry_run), prec=args.prec)
    if args.dry_run:
        args = make_dry_run_artefacts(args, tmp)
    import vps_amd
    from vps_amd import nhwc
    for f in (args.checkpoint, args.flownet_checkpoint, os.path.join(args.data_root, 'im_all_info_%s_city_vps.json' % args.mode),
              os.path.join(args.data_root, 'panoptic_im_%s_city_vps.json' % args.mode)):
        if not os.path.exists(f):
            print(json.dumps(dict(report, error='missing artefact: %s (the reference fetches it with download_weights.sh / the dataset README; '
                                                'use --dry-run for synthetic stand-ins)' % f)))
            return 2
    # ---- test_vpq.py:129-143: build, load, classes ---------------------------------------------------------------------------
    nhwc.DEFAULT_PREC = nhwc.PREC_NAMES[args.prec]
    cfg = vps_amd.Config.fromfile(args.config)
    cfg.model['pretrained'] = None
    model = vps_amd.build_detector(dict(cfg.model, flownet_checkpoint=args.flownet_checkpoint), train_cfg=None, test_cfg=cfg.test_cfg)
    checkpoint = vps_amd.load_checkpoint(model, args.checkpoint, map_location='cpu', strict=args.strict)
    rep = checkpoint.get('_vps_load_report', {})
    model.CLASSES = checkpoint.get('meta', {}).get('CLASSES', ('person', 'rider', 'car', 'truck', 'bus', 'train', 'motorcycle', 'bicycle'))
    report['checkpoint'] = dict(file=args.checkpoint, loaded=rep.get('loaded'), missing=rep.get('missing'), unexpected=rep.get('unexpected'),
                                classes=list(model.CLASSES))
    # ---- the dataset list (cityscapes_vps.py:137-148: the previous frame is the reference, a video's first frame its own) -----------
    info = json.load(open(os.path.join(args.data_root, 'im_all_info_%s_city_vps.json' % args.mode)))['images']
    img_prefix = os.path.join(args.data_root, args.mode, 'img_all')
    im_jsons = json.load(open(os.path.join(args.data_root, 'panoptic_im_%s_city_vps.json' % args.mode)))
    names = sorted(x['file_name'] for x in im_jsons['images'])
    categories = im_jsons['categories']
    span = args.nframes_span_test
    report['dataset'] = dict(frames=len(info), videos=len(info) // max(span, 1), labelled_frames=len(names), img_prefix=img_prefix)
    if args.check_only:
        print(json.dumps(report))
        return 0
    assert torch.cuda.is_available(), 'the run needs the MI355X (there is no CPU path); --check-only stops before it'
    dev = torch.device('cuda:0')
    from vps_amd.pipeline import ClipFeeder, DeviceImagePrep
    from vps_amd.postprocess import PanopticUnifier, inference_panoptic_video
    model.ensure_packed(dev)
    prep = DeviceImagePrep(**cfg.img_norm_cfg, size_divisor=32, img_scale=(2048, 1024), device=dev)
    files = [os.path.join(img_prefix, x['file_name']) for x in info]
    feeder = ClipFeeder(files, prep, workers=args.png_workers).start()
    res = dict(all_names=[], all_ssegs=[], all_panos=[], all_pano_cls_inds=[], all_pano_obj_ids=[])
    t0 = time.perf_counter()
    with torch.no_grad():
        for idx, im in enumerate(info):
            img = feeder(idx)
            ref = feeder(idx - 1) if idx % span > 0 else img
            nxt = idx + 1 if idx + 1 < len(info) else None
            meta = dict(feeder.meta(idx), iid=im['id'])       # filename, ori_shape, img_shape, pad_shape, scale_factor, flip + the video index
            # the next pair is known: its image-only stages are announced to the detector (the drop-in's own clip pipelining)
            pf = (feeder(nxt), img) if (nxt is not None and nxt % span > 0) else None
            result = model.simple_test(img, [meta], rescale=True, ref_img=[ref], prefetch=pf)
            res['all_ssegs'].append(result[2]['fcn_outputs'][0]); res['all_panos'].append(result[2]['panoptic_outputs'][0])
            res['all_pano_cls_inds'].append(result[2]['panoptic_cls_inds'].cpu().numpy())
            res['all_pano_obj_ids'].append(result[2]['panoptic_det_obj_ids'].cpu().numpy())
            res['all_names'].append(os.path.basename(files[idx]))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    feeder.close()
    # ---- test_vpq.py:150-198 ------------------------------------------------------------------------------------------------------
    output_dir = args.out.replace('.pkl', '_pans_unified/')
    unifier = PanopticUnifier(dev, 19, 9)
    two = unifier.get_unified_pan_result(res['all_ssegs'], res['all_panos'], res['all_pano_cls_inds'], obj_ids=res['all_pano_obj_ids'],
                                         stuff_area_limit=args.stuff_area_limit, names=res['all_names'])
    pred_pans_2ch = [two[k] for k in sorted(two.keys())]
    try:
        from panopticapi.utils import IdGenerator
        gen = IdGenerator({c['id']: c for c in categories})
    except ImportError:
        from run_vps_synthetic import ColorGenerator          # stand-in with the same get_color(cat_id) contract (panopticapi is absent offline)
        gen = ColorGenerator({c['id']: c for c in categories})
    kw = {}
    if args.dry_run:
        kw = dict(labeled_fid=args.dry_label_first, lambda_=5, nframes_per_video=len(names) // max(args.n_video, 1))
    pans, pj = inference_panoptic_video(pred_pans_2ch, output_dir, categories, names, n_video=args.n_video, color_generator=gen, device=dev, **kw)
    report['run'] = dict(frames=len(info), seconds=round(dt, 3), frames_per_s=round(len(info) / dt, 2), decodes=feeder.decodes, output_dir=output_dir,
                         png_files=len(os.listdir(os.path.join(output_dir, 'pan_pred'))))
    truth_dir, gt_json = args.truth_dir, args.pan_gt_json
    if args.dry_run:
        # ground truth = this run's own prediction, written in the ground-truth layout: the reference's metric must come out as 100
        truth_dir = os.path.join(args.data_root, args.mode, 'panoptic_video')
        os.makedirs(truth_dir, exist_ok=True)
        import shutil
        ims = sorted(im_jsons['images'], key=lambda x: x['file_name'])
        for im in ims:          # eval_vpq.py:272-276: ground-truth PNG = <image name> with _newImg8bit.png -> _final_mask.png
            shutil.copy(os.path.join(output_dir, 'pan_pred', im['id'] + '.png'), os.path.join(truth_dir, im['file_name'].replace('_newImg8bit.png', '_final_mask.png')))
        gt_json = os.path.join(args.data_root, 'panoptic_gt_%s_city_vps.json' % args.mode)
        json.dump(dict(annotations=pj['annotations'], categories=categories, images=ims), open(gt_json, 'w'))
    if truth_dir:
        import eval_vpq_device
        ev = ['--submit_dir', output_dir, '--truth_dir', truth_dir, '--pan_gt_json_file', gt_json or os.path.join(args.data_root, 'panoptic_gt_%s_city_vps.json' % args.mode)]
        if args.dry_run:
            ev += ['--nframes_per_video', str(len(names) // max(args.n_video, 1))]
        eval_vpq_device.main(ev)
        fin = os.path.join(output_dir, 'vpq-final.txt')
        report['vpq_final'] = open(fin).read().strip().splitlines()[-3:] if os.path.exists(fin) else None
    print(json.dumps(report))
    return 0


if __name__ == '__main__':
    sys.exit(main())
