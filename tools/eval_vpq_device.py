"""tools/eval_vpq.py of the reference (main, :251-330, and vpq_compute, :212-248) on this package: same command line, same input
files (pred.json + pan_pred/*.png of tools/test_vpq.py, the ground-truth json + *_final_mask.png / *_gtFine_color.png), same
output files (vpq-0.txt, vpq-5.txt, vpq-10.txt, vpq-15.txt, vpq-final.txt, same layout) — with every frame's confusion counts
taken once on the device (vps_amd.evaluate.FrameCounts -> vps_pair_count) and shared by the four window lengths, instead of
np.unique over nframes x H x W keys per window and per window length.

    python tools/eval_vpq_device.py --submit_dir work_dirs/.../val_pans_unified --truth_dir data/cityscapes_vps/val/panoptic_video \\
        --pan_gt_json_file data/cityscapes_vps/panoptic_gt_val_city_vps.json

`--nframes_per_video` (6 in the reference, eval_vpq.py:300) is an argument only so that small test sets can be evaluated."""
import argparse
import json
import os
import sys
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description='VPSNet eval (device-side counting)')
    ap.add_argument('--submit_dir', type=str, default='work_dirs/cityscapes_vps/fusetrack_vpct/val_pans_unified/', help='test output directory')
    ap.add_argument('--truth_dir', type=str, default='data/cityscapes_vps/val/panoptic_video', help='ground truth directory')
    ap.add_argument('--pan_gt_json_file', type=str, default='data/cityscapes_vps/panpotic_gt_val_city_vps.json', help='ground truth json')
    ap.add_argument('--nframes_per_video', type=int, default=6)
    ap.add_argument('--device', type=str, default='cuda')
    return ap.parse_args(argv)


def write_table(path, results, metrics):
    """the layout of vpq-<k>.txt (eval_vpq.py:233-244)"""
    with open(path, 'w') as f:
        f.write('=' * 48 + '\n')
        f.write('{:10s}| {:>5s}  {:>5s}  {:>5s} {:>5s}'.format('', 'PQ', 'SQ', 'RQ', 'N\n'))
        f.write('-' * (10 + 7 * 4) + '\n')
        for name, _ in metrics:
            r = results[name]
            f.write('{:10s}| {:5.1f}  {:5.1f}  {:5.1f} {:5d}\n'.format(name, 100 * r['pq'], 100 * r['sq'], 100 * r['rq'], r['n']))
        f.write('{:4s}| {:>5s} {:>5s} {:>5s} {:>6s} {:>7s} {:>7s} {:>7s}\n'.format('IDX', 'PQ', 'SQ', 'RQ', 'IoU', 'TP', 'FP', 'FN'))
        for idx, r in results['per_class'].items():
            f.write('{:4d} | {:5.1f} {:5.1f} {:5.1f} {:6.1f} {:7d} {:7d} {:7d}\n'.format(idx, 100 * r['pq'], 100 * r['sq'], 100 * r['rq'], r['iou'],
                                                                                   r['tp'], r['fp'], r['fn']))


def main(argv=None):
    from vps_amd.evaluate import PQStat, vpq_compute_single_core
    args = parse_args(argv)
    submit_dir, truth_dir, output_dir = args.submit_dir, args.truth_dir, args.submit_dir
    if not os.path.isdir(submit_dir):
        raise SystemExit("%s doesn't exist" % submit_dir)
    t_all = time.time()
    with open(os.path.join(submit_dir, 'pred.json')) as f:
        pred_jsons = json.load(f)
    with open(args.pan_gt_json_file) as f:
        gt_jsons = json.load(f)
    categories = {el['id']: el for el in gt_jsons['categories']}
    gt_files = sorted(item['file_name'].replace('_newImg8bit.png', '_final_mask.png').replace('_leftImg8bit.png', '_gtFine_color.png')
                      for item in gt_jsons['images'])
    gt_pans = [np.array(Image.open(os.path.join(truth_dir, fn))) for fn in gt_files]
    pred_pans = [np.array(Image.open(os.path.join(submit_dir, 'pan_pred', item['id'] + '.png'))) for item in gt_jsons['images']]
    assert len(gt_pans) == len(pred_pans), 'number of prediction does not match with the groud truth.'
    print('==> gt_pans / pred_pans: %d // %.2f sec' % (len(gt_pans), time.time() - t_all))
    gt_ann, pred_ann, images = gt_jsons['annotations'], pred_jsons['annotations'], gt_jsons['images']
    assert len(gt_ann) == len(pred_ann) == len(gt_pans)
    nper = args.nframes_per_video
    nvid = len(gt_ann) // nper
    videos = [[(gt_ann[i], pred_ann[i], gt_pans[i], pred_pans[i], images[i]) for i in range(v * len(gt_ann) // nvid, (v + 1) * len(gt_ann) // nvid)]
              for v in range(nvid)]
    metrics = [('All', None), ('Things', True), ('Stuff', False)]
    caches = [dict() for _ in videos]            # one per video: the frame counts are shared by the four window lengths
    vpq = {'All': [], 'Things': [], 'Stuff': []}
    for nframes in (1, 2, 3, 4):                 # k = 0, 5, 10, 15
        t0 = time.time()
        stat = PQStat()
        for v, clip in enumerate(videos):
            stat += vpq_compute_single_core(clip, categories, nframes=nframes, device=args.device, _cache=caches[v])
        k = (nframes - 1) * 5
        results = {}
        for name, isthing in metrics:
            results[name], per_class = stat.pq_average(categories, isthing=isthing)
            if name == 'All':
                results['per_class'] = per_class
        write_table(os.path.join(output_dir, 'vpq-%d.txt' % k), results, metrics)
        for name, _ in metrics:
            vpq[name].append(100 * results[name]['pq'])
        print('==> %d-frame vpq_stat: %.2f sec  %s' % (k, time.time() - t0, [round(vpq[n][-1], 4) for n in ('All', 'Things', 'Stuff')]))
    with open(os.path.join(output_dir, 'vpq-final.txt'), 'w') as f:
        f.write('vpq_all:%.4f\n' % (sum(vpq['All']) / 4))
        f.write('vpq_thing:%.4f\n' % (sum(vpq['Things']) / 4))
        f.write('vpq_stuff:%.4f\n' % (sum(vpq['Stuff']) / 4))
    print('==> All: %.2f sec' % (time.time() - t_all))
    return {n: sum(v) / 4 for n, v in vpq.items()}


if __name__ == '__main__':
    main()
