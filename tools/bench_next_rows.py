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


def bench_vpq():
    """row 3: tube statistics of an 8-frame 1024x2048 clip for the four window lengths of eval_vpq.py:main"""
    from oracle import evaluate as oev
    from vps_amd import evaluate as ev
    H, W, nfr, ninst = 1024, 2048, 8, 40
    rng = np.random.default_rng(2)
    cats = {c: {'id': c, 'isthing': 1 if c >= 11 else 0} for c in range(19)}
    base = rng.integers(0, 11, size=(H // 64, W // 64))
    boxes = [(int(rng.integers(0, H - 200)), int(rng.integers(0, W - 300)), int(rng.integers(40, 200)), int(rng.integers(40, 300))) for _ in range(ninst)]
    rgb = lambda m: np.stack([m % 256, (m // 256) % 256, m // 65536], -1).astype(np.uint8)
    frames = []
    for f in range(nfr):
        st = base.repeat(64, 0).repeat(64, 1)
        gt = (1000 + st).astype(np.int64); pr = gt.copy()
        gseg = {int(1000 + c): int(c) for c in np.unique(st)}; pseg = dict(gseg)
        for i, (y, x, h, w) in enumerate(boxes):
            y = min(y + 2 * f, H - h); x = min(x + 3 * f, W - w)
            gt[y:y + h, x:x + w] = 5000 + i; gseg[5000 + i] = 11 + i % 8
            if i % 6 != 5:
                pr[y + 3:y + h, x + 2:x + w] = 9000 + i; pseg[9000 + i] = 11 + i % 8
        ginfo = [{'id': k, 'category_id': v, 'iscrowd': 0, 'area': int((gt == k).sum())} for k, v in gseg.items() if (gt == k).any()]
        pinfo = [{'id': k, 'category_id': v, 'iscrowd': 0, 'area': int((pr == k).sum())} for k, v in pseg.items() if (pr == k).any()]
        frames.append(({'segments_info': ginfo}, {'segments_info': pinfo}, rgb(gt), rgb(pr), {}))
    dev = torch.device('cuda:0')
    dframes = [(a, b, torch.from_numpy(c).to(dev), torch.from_numpy(d).to(dev), e) for a, b, c, d, e in frames]
    ev.vpq_compute_single_core(dframes[:2], cats, nframes=1, device=dev)              # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cache, res = {}, {}
    for nf in (1, 2, 3, 4):
        res[nf] = ev.vpq_compute_single_core(dframes, cats, nframes=nf, device=dev, _cache=cache)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    ref = oev.vpq_compute_single_core(frames, cats, nframes=2)
    cpu_ms_nf2 = (time.perf_counter() - t0) * 1e3
    for c in cats:
        assert (res[2][c].tp, res[2][c].fp, res[2][c].fn, res[2][c].iou) == (ref[c].tp, ref[c].fp, ref[c].fn, ref[c].iou)
    abytes = nfr * H * W * 6
    print(json.dumps({
        'metric': 'clips/sec VPQ tube statistics (8 frames 1024x2048, window lengths 1-4)', 'value': round(1e3 / ms, 2), 'unit': 'clips/s',
        'ms_per_clip': round(ms, 2), 'dtype': 'u8 / int', 'data': 'synthetic',
        'config': {'workload': '8-frame clip, 40 instances + stuff, PNG arrays resident in HBM, wall clock incl. host matching'},
        'roofline': {'bound': 'hbm', 'achieved': round(abytes / ms / 1e6, 2), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(abytes / ms / 1e6 / 8000.0, 5),
                     'traffic': None, 'algorithmic_bytes': abytes},
        'cpu_baseline': {'value': round(1e3 / (cpu_ms_nf2 * 4), 3), 'unit': 'clips/s', 'cores': 1, 'kind': 'port',
                         'sample': 'window length 2 only (%.0f ms), x4 for the four lengths' % cpu_ms_nf2}}))


if __name__ == '__main__':
    main()
    bench_prep()
    bench_converter()
    bench_vpq()
