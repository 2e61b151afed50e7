"""Attribution of the VPQ between the arithmetic modes: is the deficit of the benchmarked f16x3 mode against the exact
fp32 kernels a precision effect, or the noise floor ANY change of summation order has on near-tied synthetic scores?

On the same 2 x 30 synthetic frames at 1024x2048 (whole drop-in chain of tools/run_vps_synthetic.py: upload / prep -> detector ->
unifier -> panoptic video -> VPQ over window lengths 1..4), against the exact-fp32 kernels in their default schedule:

    f32 / other split-K      exact fp32 MFMA, another VPS_SPLITK_TARGET (different partial-sum grouping of the low-resolution layers)
    f32 / one stream         exact fp32 MFMA, no stream overlap (must be bitwise: a control)
    bf16x6                   three bf16 planes per operand, six products
    f16x3                    the benchmarked arithmetic

plus, per run, how many frames list different instances and the fraction of differing pixels of both maps. Both weight sets
(near-tied default heads / separated fc_cls fixture).

    python tools/vpq_attribution.py --height 1024 --width 2048 --videos 2 --frames 30 > profiles/r04_vpq_attribution.json
"""
import argparse
import json
import os
import shutil
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import run_vps_synthetic as RV   # noqa: E402


def run(tag, prec, args, dev, separated, splitk=None, streams=True):
    import vps_amd.detector as D
    from vps_amd import nhwc
    old_t = nhwc.SPLITK_TARGET_BLOCKS
    try:
        if splitk is not None:
            nhwc.SPLITK_TARGET_BLOCKS = splitk
        if not streams:
            orig_init = D.PanopticFuseTrack.__init__

            def init(self, *a, **k):
                orig_init(self, *a, **k)
                self.overlap_streams = False
            D.PanopticFuseTrack.__init__ = init
        try:
            res, dt = RV.run_model(prec, args.videos, args.frames, args.height, args.width, dev, separated)
        finally:
            if not streams:
                D.PanopticFuseTrack.__init__ = orig_init
    finally:
        nhwc.SPLITK_TARGET_BLOCKS = old_t
    out = os.path.join(args.out, tag)
    nper = len(range(20 // 5, args.frames, 5))
    names, pans, pj = RV.postprocess(res, out, args.videos, dev, 20, 5, nper)
    host = dict(cls=[np.asarray(c) for c in res['all_pano_cls_inds']], ids=[np.asarray(c) for c in res['all_pano_obj_ids']],
                sem=[m.cpu().numpy() for m in res['all_ssegs']], pan=[m.cpu().numpy() for m in res['all_panos']])
    shutil.rmtree(out, ignore_errors=True)
    return dict(pans=pans, pj=pj, host=host, nper=nper, fps=args.videos * args.frames / dt)


def compare(ref, got, args, dev):
    score = RV.vpq((ref['pans'], ref['pj']), (got['pans'], got['pj']), args.videos, ref['nper'], dev)
    h0, h1 = ref['host'], got['host']
    n = len(h0['cls'])
    lists = sum(1 for a, b in zip(h0['cls'], h1['cls']) if not (a.shape == b.shape and np.array_equal(a, b)))
    ids = sum(1 for a, b in zip(h0['ids'], h1['ids']) if not (a.shape == b.shape and np.array_equal(a, b)))
    sem = float(np.mean([np.mean(a != b) for a, b in zip(h0['sem'], h1['sem'])]))
    pan = float(np.mean([np.mean(a != b) for a, b in zip(h0['pan'], h1['pan'])]))
    return dict(vpq=round(score['vpq'], 4), pq_per_window={str(k): round(100 * score[k]['pq'], 4) for k in (1, 2, 3, 4)},
                frames=n, frames_with_a_different_instance_list=lists, frames_with_different_ids=ids,
                semantic_map_pixels_differing=round(sem, 7), panoptic_map_pixels_differing=round(pan, 7), frames_per_s=round(got['fps'], 2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--videos', type=int, default=2)
    ap.add_argument('--frames', type=int, default=30)
    ap.add_argument('--height', type=int, default=1024)
    ap.add_argument('--width', type=int, default=2048)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'vpq_attr'))
    ap.add_argument('--sets', default='near_tied,separated')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    report = dict(size=[args.height, args.width], videos=args.videos, frames_per_video=args.frames, reference='exact fp32 MFMA kernels, default schedule',
                  note='VPQ of each run against the reference run (the reference run is the "ground truth"): 100 = identical panoptic videos')
    for wset in args.sets.split(','):
        sep = wset == 'separated'
        ref = run('ref', 'f32', args, dev, sep)
        rows = {}
        rows['f32_self_control'] = compare(ref, run('a', 'f32', args, dev, sep), args, dev)
        rows['f32_other_splitk_target_256'] = compare(ref, run('b', 'f32', args, dev, sep, splitk=256), args, dev)
        rows['f32_other_splitk_target_1024'] = compare(ref, run('b2', 'f32', args, dev, sep, splitk=1024), args, dev)
        rows['f32_one_stream'] = compare(ref, run('c', 'f32', args, dev, sep, streams=False), args, dev)
        rows['bf16x6'] = compare(ref, run('d', 'bf16x6', args, dev, sep), args, dev)
        rows['f16x3'] = compare(ref, run('e', 'f16x3', args, dev, sep), args, dev)
        report[wset] = rows
        print(wset, json.dumps(rows), file=sys.stderr, flush=True)
    print(json.dumps(report, indent=1))


if __name__ == '__main__':
    main()
