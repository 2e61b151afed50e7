"""SURVEY 8(f) row 1, measured end to end (VERDICT r2 "Missing" #6 / next-round #10): can the input side feed the detector?

    python tools/bench_input_pipeline.py [--frames 48] [--workers 1 2 4 8 16]

Writes `--frames` synthetic 1024x2048 8-bit PNG frames (Cityscapes-VPS size) to a temporary directory, then times
  decode   `vps_amd.pipeline.imread` (mmcv.imread semantics; PIL or cv2) with N host worker threads, frames/s
  upload   the decoded uint8 HWC frame -> device (6 MB, pinned and pageable)
  prep     `DeviceImagePrep.prep` (Normalize + Pad + ImageToTensor as one kernel)
and the three chained with a prefetching thread pool (decode of frame t+k overlaps upload + prep of frame t), which is what a
data loader in front of `tools/test_vpq.py` would do. The detector runs at ~42 frames/s on one MI355X: the question is how many
host threads the decode needs to keep up. Prints one JSON line. The GPU part is skipped without a GPU.
"""
import argparse
import json
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=48)
    ap.add_argument('--workers', type=int, nargs='+', default=[1, 2, 4, 8, 16])
    ap.add_argument('--height', type=int, default=1024)
    ap.add_argument('--width', type=int, default=2048)
    args = ap.parse_args()
    import torch
    from PIL import Image
    from vps_amd import synth
    from vps_amd.pipeline import imread
    H, W = args.height, args.width
    tmp = tempfile.mkdtemp(prefix='vps_input_')
    files = []
    for t in range(args.frames):
        # low-pass noise + per-frame translation and noise: compresses like a camera frame rather than like a flat image
        fr = synth.synth_frame(H, W, seed=t % 4, shift=(2 * (t % 8), t % 8), noise=2.0).astype(np.uint8)
        fn = os.path.join(tmp, 'f%04d_leftImg8bit.png' % t)
        Image.fromarray(np.ascontiguousarray(fr[:, :, ::-1])).save(fn, compress_level=6)
        files.append(fn)
    mb = sum(os.path.getsize(f) for f in files) / 1e6 / len(files)
    out = dict(frames=args.frames, size=[H, W], png_MB_per_frame=round(mb, 2), host_cpus=os.cpu_count(), decode_frames_per_s={})
    for nw in args.workers:
        with ThreadPoolExecutor(nw) as ex:
            list(ex.map(imread, files[:nw]))                       # warm-up
            t0 = time.perf_counter()
            list(ex.map(imread, files))
            out['decode_frames_per_s'][str(nw)] = round(len(files) / (time.perf_counter() - t0), 1)
    if torch.cuda.is_available():
        from vps_amd.pipeline import DeviceImagePrep
        dev = torch.device('cuda:0')
        prep = DeviceImagePrep(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True, size_divisor=32, device=dev)
        img = imread(files[0])
        pinned = torch.from_numpy(img).pin_memory()
        for name, src in (('pageable', torch.from_numpy(img)), ('pinned', pinned)):
            for _ in range(3):
                src.to(dev, non_blocking=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                src.to(dev, non_blocking=True)
            torch.cuda.synchronize()
            out['upload_ms_' + name] = round(1e3 * (time.perf_counter() - t0) / 20, 3)
        d = pinned.to(dev)
        for _ in range(3):
            prep.prep(d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            prep.prep(d)
        torch.cuda.synchronize()
        out['prep_ms'] = round(1e3 * (time.perf_counter() - t0) / 50, 4)
        out['chained_frames_per_s'] = {}
        for nw in args.workers:
            with ThreadPoolExecutor(nw) as ex:
                t0 = time.perf_counter()
                futs = [ex.submit(imread, f) for f in files]       # the pool runs ahead of the consumer
                for f in futs:
                    prep.prep(torch.from_numpy(f.result()).to(dev, non_blocking=True))
                torch.cuda.synchronize()
                out['chained_frames_per_s'][str(nw)] = round(len(files) / (time.perf_counter() - t0), 1)
    for f in files:
        os.remove(f)
    os.rmdir(tmp)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
