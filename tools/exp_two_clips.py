"""Experiment: is there throughput left in running MORE of the frame graph concurrently? Two independent detectors (own weights,
workspace and streams) process two synthetic clips from two host threads of ONE process on ONE GPU; the aggregate frames/s is
compared with one clip alone. If two clips in flight are not faster than one, the chip is already full with the two-stream
pipeline of `detector.py` and a deeper cross-frame pipeline cannot pay.

    python tools/exp_two_clips.py [--frames 24] [--prec f16x3]
"""
import argparse
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=24)
    ap.add_argument('--prec', default='f16x3')
    ap.add_argument('--height', type=int, default=1024)
    ap.add_argument('--width', type=int, default=2048)
    args = ap.parse_args()
    import vps_amd
    from vps_amd import nhwc, synth
    from vps_amd.clip_shard import ClipShardRunner, DetectorBackend
    nhwc.DEFAULT_PREC = nhwc.PREC_NAMES[args.prec]
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    H, W, n = args.height, args.width, args.frames
    cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'cityscapes', 'fusetrack.py'))
    pool = [f.to(dev) for f in synth.synth_clip(H, W, 8, 0)]
    runners = []
    for i in range(2):
        m = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
        synth.load_synth(m, 0)
        runners.append(ClipShardRunner(DetectorBackend(m, H, W), 0, 1, None, dev))

    def run(i, nframes, vid):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            runners[i].run(lambda t: pool[(t + 3 * i) % 8], nframes, video_id=vid)
            torch.cuda.current_stream().synchronize()

    for i in range(2):
        run(i, 4, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(0, n, 2)
    torch.cuda.synchronize()
    one = n / (time.perf_counter() - t0)
    th = [threading.Thread(target=run, args=(i, n, 3)) for i in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    two = 2 * n / (time.perf_counter() - t0)
    print(json.dumps(dict(frames_per_clip=n, prec=args.prec, one_clip_frames_per_s=round(one, 2), two_clips_aggregate_frames_per_s=round(two, 2),
                          gain=round(two / one, 3))))


if __name__ == '__main__':
    main()
