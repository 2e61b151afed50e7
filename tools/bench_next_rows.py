"""Throughput of the two widened rows of SURVEY §8(f) at 1024x2048 (GPU box only), one JSON line each:
row 2, device-side panoptic post-processing (k = 100 instances), and row 1, device-side Normalize -> Pad -> ImageToTensor,
each next to the NumPy restatement of the reference code on the host."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import pipeline as opl
from oracle import postprocess as opp
from vps_amd import pipeline as pl
from vps_amd import postprocess as pp


def bench_prep():
    H, W = 1024, 2048
    norm = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
    img = np.random.default_rng(0).integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    dev = torch.device('cuda:0')
    prep = pl.DeviceImagePrep(**norm, device=dev)
    imgd = torch.from_numpy(img).to(dev)
    out = prep.prep(imgd)[0]
    assert np.array_equal(out.cpu().numpy(), opl.prepare(img, **norm))
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    reps = 50
    e0.record()
    for _ in range(reps):
        prep.prep(imgd)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    t0 = time.perf_counter()
    for _ in range(3):
        opl.prepare(img, **norm)
    cpu_ms = (time.perf_counter() - t0) / 3 * 1e3
    abytes = H * W * 3 * (1 + 4)
    print(json.dumps({
        'metric': 'images/sec Normalize+Pad+ImageToTensor 1024x2048', 'value': round(1e3 / ms, 1), 'unit': 'images/s',
        'ms_per_image': round(ms, 4), 'dtype': 'u8 -> f32', 'data': 'synthetic', 'config': {'workload': 'decoded uint8 BGR 1024x2048 image resident in HBM'},
        'roofline': {'bound': 'hbm', 'achieved': round(abytes / ms / 1e6, 2), 'peak': 8000.0, 'unit': 'GB/s',
                     'frac': round(abytes / ms / 1e6 / 8000.0, 4), 'traffic': None, 'algorithmic_bytes': abytes},
        'cpu_baseline': {'value': round(1e3 / cpu_ms, 2), 'unit': 'images/s', 'cores': 1, 'kind': 'port', 'sample': '3 images, NumPy restatement'}}))


def main():
    H, W, k = 1024, 2048, 100
    rng = np.random.default_rng(0)
    seg = (np.arange(W) * 11 // W).astype(np.uint8)[None, :].repeat(H, 0)
    seg[H // 2:] = rng.integers(0, 19, size=(H - H // 2, W)).astype(np.uint8)
    pan = np.minimum(seg, 10).astype(np.uint8)
    cls_ind = rng.integers(0, 8, size=k).astype(np.int64)
    for i in range(k):
        h, w = int(rng.integers(20, 200)), int(rng.integers(20, 300))
        y, x = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
        pan[y:y + h, x:x + w] = 11 + i
        if i % 3 == 0:
            seg[y:y + h, x:x + w] = 11 + cls_ind[i]
    obj = rng.permutation(200)[:k].astype(np.int64)
    dev = torch.device('cuda:0')
    u = pp.PanopticUnifier(dev)
    sd, pd = torch.from_numpy(seg).to(dev), torch.from_numpy(pan).to(dev)
    out = u.unify_frame(sd, pd, cls_ind, obj)
    ref = opp.unify_frame(seg, pan, cls_ind, obj)
    assert np.array_equal(out.cpu().numpy(), ref)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    reps = 50
    e0.record()
    for _ in range(reps):
        u.unify_frame(sd, pd, cls_ind, obj)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    t0 = time.perf_counter()
    for _ in range(3):
        opp.unify_frame(seg, pan, cls_ind, obj)
    cpu_ms = (time.perf_counter() - t0) / 3 * 1e3
    npix = H * W
    abytes = npix * (2 + 1 + 3)                      # hist pass reads pan+seg, write pass reads pan and writes 3 channels
    print(json.dumps({
        'metric': 'frames/sec panoptic unify 1024x2048 (get_unified_pan_result body)', 'value': round(1e3 / ms, 1), 'unit': 'frames/s',
        'ms_per_frame': round(ms, 4), 'dtype': 'u8', 'data': 'synthetic', 'config': {'workload': '1024x2048 maps, 100 instances, object ids'},
        'roofline': {'bound': 'hbm', 'achieved': round(abytes / ms / 1e6, 2), 'peak': 8000.0, 'unit': 'GB/s',
                     'frac': round(abytes / ms / 1e6 / 8000.0, 4), 'traffic': None, 'algorithmic_bytes': abytes},
        'cpu_baseline': {'value': round(1e3 / cpu_ms, 2), 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
                         'sample': '3 frames, NumPy restatement (table-based; the reference builds one boolean mask per instance)'}}))


class _Colors:
    def __init__(self):
        self.n = 0

    def get_color(self, cat_id):
        self.n += 1
        return [int(cat_id) * 7 % 256, self.n % 256, (self.n * 37) % 256]


def bench_converter():
    """converter_2ch_track_core on the (pan_seg, pan_ins, pan_obj) maps the unify step produces"""
    H, W, k = 1024, 2048, 60
    rng = np.random.default_rng(1)
    seg = np.ascontiguousarray(rng.integers(0, 11, size=(H // 64, W // 64)).astype(np.uint8).repeat(64, 0).repeat(64, 1))
    ins = np.zeros((H, W), np.uint8); obj = seg.copy()
    for i in range(k):
        h, w = int(rng.integers(20, 200)), int(rng.integers(20, 300))
        y, x = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
        seg[y:y + h, x:x + w] = 11 + i % 8; ins[y:y + h, x:x + w] = i + 1; obj[y:y + h, x:x + w] = i + 1
    pan2 = np.stack([seg, ins, obj], -1)
    dev = torch.device('cuda:0')
    conv = pp.TrackConverter(dev)
    pd = torch.from_numpy(pan2).to(dev)
    ann_d, pans_d = conv.convert([pd], _Colors())
    t0 = time.perf_counter()
    ann_r, pans_r = opp.converter_2ch_track_core([pan2], _Colors())
    cpu_ms = (time.perf_counter() - t0) * 1e3
    assert np.array_equal(pans_d[0], pans_r[0]) and ann_d == ann_r
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        conv.convert([pd], _Colors())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3            # wall clock: includes the host colour logic and the 6 MB D2H of the result
    abytes = H * W * (3 + 3 + 3)
    print(json.dumps({
        'metric': 'frames/sec converter_2ch_track_core 1024x2048', 'value': round(1e3 / ms, 1), 'unit': 'frames/s', 'ms_per_frame': round(ms, 3),
        'dtype': 'u8', 'data': 'synthetic', 'config': {'workload': '1024x2048, 60 instances + 11 stuff classes, wall clock incl. host colour logic and D2H'},
        'roofline': {'bound': 'hbm', 'achieved': round(abytes / ms / 1e6, 2), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(abytes / ms / 1e6 / 8000.0, 4),
                     'traffic': None, 'algorithmic_bytes': abytes},
        'cpu_baseline': {'value': round(1e3 / cpu_ms, 2), 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
                         'sample': '1 frame, NumPy restatement (one boolean mask per segment, as the reference)'}}))


if __name__ == '__main__':
    main()
    bench_prep()
    bench_converter()
