"""SURVEY 8(f) row 1, measured end to end (VERDICT r2 "Missing" #6 / next-round #10): can the input side feed the detector?

    python tools/bench_input_pipeline.py [--frames 48] [--workers 1 2 4 8 16]

Writes `--frames` synthetic 1024x2048 8-bit PNG frames (Cityscapes-VPS size) to a temporary directory, then times
  decode   `vps_amd.pipeline.imread` (mmcv.imread semantics; PIL or cv2) with N host worker threads, frames/s
  upload   the decoded uint8 HWC frame -> device (6 MB, pinned and pageable)
  prep     `DeviceImagePrep.prep` (Normalize + Pad + ImageToTensor as one kernel)
and the three chained with a prefetching thread pool (decode of frame t+k overlaps upload + prep of frame t), which is what a
data loader in front of `tools/test_vpq.py` would do. The detector runs at ~42 frames/s on one MI355X: the question is how many
host threads the decode needs to keep up. Prints one JSON line. The GPU part is skipped without a GPU.

    python tools/bench_input_pipeline.py --node 8 --threads 6 [--seconds 8]

NODE mode (round 6): what the host of an 8-GPU node has to deliver - `--node` decoder PROCESSES (one per rank, as
`bench.py --gpus 8` starts them), each with the thread budget of a rank's `ClipFeeder` (`--threads`, default 6) and pinned to its own
block of cores (os.sched_setaffinity: cores [r * C/8, (r+1) * C/8) - a rank's decode threads stay on the cores next to its GPU and
never compete with another rank's), all decoding 1024x2048 PNG frames with the library's native decoder (`vps_png_decode_bgr8`,
what ClipFeeder uses) for `--seconds`. Reports per-rank and aggregate frames/s; the node needs 8 x (frames/s of one GPU).
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


def _rank_decode(rank, nranks, threads, files, seconds, q):
    # one rank's input side: `threads` decoder threads on this rank's own block of cores
    ncpu = os.cpu_count() or 1
    per = max(ncpu // nranks, 1)
    cores = list(range(rank * per, min((rank + 1) * per, ncpu)))
    pinned = False
    try:
        os.sched_setaffinity(0, cores)
        pinned = True
    except (AttributeError, OSError):
        pass
    import ctypes
    from vps_amd import hip
    lib = hip.load_host()
    # the files sit in the page cache; decode is what is measured - into per-thread staging buffers allocated once, like ClipFeeder's
    # pinned ring slots (a fresh 6 MB numpy array per frame is an mmap + page faults per decode: the process' threads then queue on the mm lock)
    datas = [bytearray(open(f, 'rb').read()) for f in files]
    cbufs = [(ctypes.c_char * len(d)).from_buffer(d) for d in datas]
    H, W, C = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    assert lib.vps_png_info(cbufs[0], len(datas[0]), ctypes.byref(H), ctypes.byref(W), ctypes.byref(C)) == 0
    stages = [np.empty(H.value * W.value * 3, dtype=np.uint8) for _ in range(threads)]
    n = [0] * threads
    stop = time.perf_counter() + seconds

    def work(i):
        k = i
        out = stages[i].ctypes.data_as(ctypes.c_void_p)
        while time.perf_counter() < stop:
            j = k % len(datas)
            hip.check(lib.vps_png_decode_bgr8(cbufs[j], len(datas[j]), out, stages[i].nbytes), 'vps_png_decode_bgr8')
            n[i] += 1
            k += threads
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(threads)))
    q.put((rank, sum(n) / (time.perf_counter() - t0), len(cores), pinned))


def node_mode(args):
    import multiprocessing as mp
    from PIL import Image
    from vps_amd import synth
    H, W = args.height, args.width
    tmp = tempfile.mkdtemp(prefix='vps_input_')
    files = []
    for t in range(8):
        fr = synth.synth_frame(H, W, seed=t % 4, shift=(2 * (t % 8), t % 8), noise=2.0).astype(np.uint8)
        fn = os.path.join(tmp, 'f%04d_leftImg8bit.png' % t)
        Image.fromarray(np.ascontiguousarray(fr[:, :, ::-1])).save(fn, compress_level=6)
        files.append(fn)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_decode, args=(r, args.node, args.threads, files, args.seconds, q)) for r in range(args.node)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=args.seconds * 4 + 120) for _ in procs)
    for p in procs:
        p.join()
    for f in files:
        os.remove(f)
    os.rmdir(tmp)
    per_rank = [round(r[1], 1) for r in res]
    out = dict(mode='node', ranks=args.node, threads_per_rank=args.threads, host_cpus=os.cpu_count(), cores_per_rank=res[0][2], pinned=all(r[3] for r in res),
               size=[H, W], png_MB_per_frame=round(sum(os.path.getsize(f) for f in files if os.path.exists(f)) / 1e6 / 8, 2) if False else None,
               decode_frames_per_s_per_rank=per_rank, aggregate_frames_per_s=round(sum(per_rank), 1), slowest_rank_frames_per_s=min(per_rank),
               needed_per_rank=args.need, headroom_of_the_slowest_rank=round(min(per_rank) / args.need, 2), seconds=args.seconds,
               decoder='vps_png_decode_bgr8 (libvpship host code, zlib inflate + unfilter + BGR), the decoder of vps_amd.pipeline.ClipFeeder')
    out.pop('png_MB_per_frame')
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=48)
    ap.add_argument('--workers', type=int, nargs='+', default=[1, 2, 4, 8, 16])
    ap.add_argument('--height', type=int, default=1024)
    ap.add_argument('--width', type=int, default=2048)
    ap.add_argument('--node', type=int, default=0, help='node mode: this many rank processes decode concurrently')
    ap.add_argument('--threads', type=int, default=6, help='node mode: decode threads per rank (ClipFeeder workers)')
    ap.add_argument('--seconds', type=float, default=8.0)
    ap.add_argument('--need', type=float, default=56.0, help='node mode: frames/s one GPU consumes')
    args = ap.parse_args()
    if args.node:
        return node_mode(args)
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
