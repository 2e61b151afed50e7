"""Functional check of the 2-rank clip pipeline with the PRODUCT backend (vps_amd.clip_shard.DetectorBackend) on one box:

    VPS_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
        tools/check_two_rank.py [--height 128 --width 256 --frames 6] [--separated]

Two processes (sharing the GPU when the box has one: gloo, NOT RCCL) run ClipShardRunner over a synthetic clip; rank 0 then runs
the same clip sequentially in one process and compares every frame: track ids, classes, scores, both maps (array_equal) and the feature
rank 1 received at the shard boundary against gathered_feature(last frame of rank 0) (bitwise). Exit code 1 on a difference.
  --withhold R   rank R computes nothing and sends nothing (a stalled peer): the ranks that wait for it must give up after
                 VPS_CLIP_TIMEOUT_S seconds with ONE JSON line {"error": ...} on stdout and exit code 3 - not hang (tests/test_two_rank_gpu.py)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--height', type=int, default=128)
    ap.add_argument('--width', type=int, default=256)
    ap.add_argument('--frames', type=int, default=6)
    ap.add_argument('--prec', default='f16x3')
    ap.add_argument('--separated', action='store_true')
    ap.add_argument('--withhold', type=int, default=-1)
    args = ap.parse_args()
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    backend = os.environ.get('VPS_BENCH_BACKEND', 'nccl')
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=dev)
    else:
        dist.init_process_group(backend)
    import vps_amd
    from vps_amd import nhwc, synth
    from vps_amd.clip_shard import ClipShardRunner, DetectorBackend
    nhwc.DEFAULT_PREC = nhwc.PREC_NAMES[args.prec]
    H, W, n = args.height, args.width, args.frames
    cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'cityscapes', 'fusetrack.py'))
    model = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    over = synth.separated_overrides(os.path.join(ROOT, 'tests', 'golden', 'separated_fc_cls.npz')) if args.separated else None
    synth.load_synth(model, 0, overrides=over)
    frames = [f.to(dev) for f in synth.synth_clip(H, W, n, 0)]
    runner = ClipShardRunner(DetectorBackend(model, H, W), rank, world, dist, dev)
    if rank == args.withhold:
        time.sleep(3 * runner.wait_timeout_s + 30)           # a stalled peer: alive, silent (the launcher ends it when another rank fails)
        sys.exit(4)
    from vps_amd.clip_shard import ClipShardError, partition
    try:
        outs = runner.run(lambda t: frames[t], n, video_id=1)
    except ClipShardError as ex:
        print(json.dumps({'error': str(ex), 'rank': rank, 'world': world}), flush=True)
        os._exit(3)                                          # no collective teardown with a peer that does not answer
    torch.cuda.synchronize()
    # the feature every later rank received at its shard boundary -> rank 0 (gathered on the host side of the check, not the data path)
    got = [None] * world
    dist.all_gather_object(got, runner.last_ref_feature.cpu() if rank > 0 and getattr(runner, 'last_ref_feature', None) is not None else None)
    dist.barrier()
    bad = 0
    if rank == 0:
        parts = partition(n, world)
        for r in range(1, world):
            if got[r] is None:
                continue
            want = model.gathered_feature(frames[parts[r][0] - 1]).cpu()
            same = bool(torch.equal(want.reshape(-1), got[r].reshape(-1)))
            print('hand-off feature rank %d -> %d (frame %d, %.1f MB): %s' % (r - 1, r, parts[r][0] - 1, want.numel() * 4 / 1e6, 'bitwise equal' if same else 'DIFFERENT'), flush=True)
            bad += not same
        model._cache = None; model._pf = None; model._handoff = None; model.reset_tracker()
        for t in range(n):
            out = model(return_loss=False, rescale=True, img=[frames[t]], img_meta=[[synth.img_meta(H, W, 10000 + t + 1)]],
                        ref_img=[frames[t - 1 if t else 0]])
            torch.cuda.synchronize()
            r = {k: v.cpu().numpy() for k, v in out[2].items()}
            o = outs[t]
            same = {
                'ids': bool(np.array_equal(np.asarray(o['panoptic_det_obj_ids']), r['panoptic_det_obj_ids'])),
                'cls': bool(np.array_equal(torch.as_tensor(o['panoptic_cls_inds']).cpu().numpy(), r['panoptic_cls_inds'])),
                'prob': bool(np.array_equal(torch.as_tensor(o['panoptic_cls_prob']).cpu().numpy(), r['panoptic_cls_prob'])),
                'pan': bool(np.array_equal(torch.as_tensor(o['panoptic_outputs']).cpu().numpy(), r['panoptic_outputs'])),
                'sem': bool(np.array_equal(torch.as_tensor(o['fcn_outputs']).cpu().numpy(), r['fcn_outputs'])),
            }
            extra = '' if same['ids'] else '  ids %s vs %s' % (np.asarray(o['panoptic_det_obj_ids']).tolist(), r['panoptic_det_obj_ids'].tolist())
            if not same['prob']:
                pa, pb = torch.as_tensor(o['panoptic_cls_prob']).cpu().numpy(), r['panoptic_cls_prob']
                extra += ('  max |dprob| %.3e over %d of %d scores; pipeline %s sequential %s' % (float(np.abs(pa - pb).max()), int((pa != pb).sum()), pa.size, pa[:4].tolist(), pb[:4].tolist())) if pa.shape == pb.shape else '  prob shapes %s vs %s' % (pa.shape, pb.shape)
            print('frame %d (rank %d): %s%s' % (t, 0 if t < (n + 1) // 2 else 1, same, extra), flush=True)
            bad += not all(same.values())
        print('2-rank pipeline %s the sequential run (%dx%d, %d frames, %s, backend %s)' % ('EQUALS' if not bad else 'DIFFERS FROM', H, W, n, args.prec, backend))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
