"""Panoptic post-processing (get_unified_pan_result).
CPU: the oracle restatement against golden outputs of the REAL reference function (tests/golden/make_unify_golden.py).
GPU: the device path (vps_unify_* through the C-ABI) against the oracle and the golden outputs, bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import postprocess as opp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'unify_cases.npz')
# the same cases through the REAL tools/dataset/viper.py:661-727 (23 segmentation classes, 11 of them things: id_last_stuff 12)
GOLD_VIPER = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'unify_cases_viper.npz')
DATASETS = [('cityscapes_vps', GOLD, 19, 9), ('viper', GOLD_VIPER, 23, 11)]


def _clips(gold=GOLD):
    z = np.load(gold)
    for ci in range(int(z['nclips'])):
        n = int(z['clip%d_n' % ci]); with_obj = bool(z['clip%d_with_obj' % ci])
        segs = [z['clip%d_f%d_seg' % (ci, f)] for f in range(n)]
        pans = [z['clip%d_f%d_pan' % (ci, f)] for f in range(n)]
        clss = [z['clip%d_f%d_cls' % (ci, f)] for f in range(n)]
        objs = [z['clip%d_f%d_obj' % (ci, f)] for f in range(n)] if with_obj else None
        outs = [z['clip%d_f%d_out' % (ci, f)] for f in range(n)]
        yield ci, segs, pans, clss, objs, int(z['clip%d_limit' % ci]), ['f%d' % f for f in range(n)], outs


@pytest.mark.parametrize('name,gold,nseg,ncls', DATASETS, ids=[d[0] for d in DATASETS])
def test_oracle_matches_reference_function_bit_for_bit(name, gold, nseg, ncls):
    assert int(np.load(gold)['id_last_stuff']) == nseg - ncls
    ncase = 0
    for ci, segs, pans, clss, objs, limit, names, outs in _clips(gold):
        res = opp.get_unified_pan_result(segs, pans, clss, objs, limit, names, id_last_stuff=nseg - ncls)
        for n, o in zip(names, outs):
            assert res[n].dtype == np.uint8 and res[n].shape == o.shape
            assert np.array_equal(res[n], o), 'clip %d frame %s differs in %d pixels' % (ci, n, int((res[n] != o).any(-1).sum()))
            ncase += 1
    assert ncase >= 10


def test_oracle_dedup_keeps_last_occurrence_and_counts_across_frames():
    ids, mx = opp.dedup_obj_ids(np.array([5, 7, 5, 9, 5, 7]), 100)
    assert ids.tolist() == [101, 102, 100, 9, 5, 7] and mx == 103


def test_host_dedup_equals_the_reference_statements():
    """vps_amd's own list-based de-duplication against the restated numpy statements of the reference, random id lists"""
    from vps_amd import postprocess as pp
    rng = np.random.default_rng(5)
    mx_a = mx_b = 100
    for _ in range(300):
        ids = rng.integers(0, 12, size=int(rng.integers(0, 15))).astype(np.int64)
        a, mx_a = pp.dedup_obj_ids(ids, mx_a)
        b, mx_b = opp.dedup_obj_ids(ids, mx_b)
        assert a.tolist() == list(b) and mx_a == mx_b


@pytest.mark.gpu
@pytest.mark.parametrize('name,gold,nseg,ncls', DATASETS, ids=[d[0] for d in DATASETS])
def test_device_unify_matches_reference_golden(dev, name, gold, nseg, ncls):
    from vps_amd import postprocess as pp
    assert pp.DATASETS[name]['num_seg_classes'] == nseg and pp.DATASETS[name]['num_classes'] == ncls
    for ci, segs, pans, clss, objs, limit, names, outs in _clips(gold):
        u = pp.PanopticUnifier(dev, **{k: pp.DATASETS[name][k] for k in ('num_seg_classes', 'num_classes')})
        res = u.get_unified_pan_result([torch.from_numpy(s).to(dev) for s in segs], [torch.from_numpy(p).to(dev) for p in pans],
                                       clss, objs, limit, names)
        for n, o in zip(names, outs):
            assert res[n].dtype == np.uint8 and np.array_equal(res[n], o), 'clip %d frame %s' % (ci, n)


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,k,seed', [(1024, 2048, 100, 0), (1024, 2048, 3, 1), (37, 53, 17, 2), (8, 8, 0, 3)])
def test_device_unify_matches_oracle_random(dev, H, W, k, seed):
    """full-size and ragged maps: random instance rectangles over random stuff, repeated object ids, void pixels"""
    from vps_amd import postprocess as pp
    rng = np.random.default_rng(seed)
    seg = rng.integers(0, 19, size=(H, W)).astype(np.uint8)
    seg[: H // 2] = (np.arange(W) * 11 // max(W, 1)).astype(np.uint8)[None, :]      # large coherent stuff areas
    pan = np.minimum(seg, 10).astype(np.uint8)
    cls_ind = rng.integers(0, 8, size=k).astype(np.int64)
    for i in range(k):
        h, w = int(rng.integers(2, max(3, H // 4))), int(rng.integers(2, max(3, W // 4)))
        y, x = int(rng.integers(0, max(1, H - h))), int(rng.integers(0, max(1, W - w)))
        pan[y:y + h, x:x + w] = 11 + i
        if i % 3 == 0:
            seg[y:y + h, x:x + w] = 11 + cls_ind[i]
        elif i % 3 == 1:
            seg[y:y + h, x:x + w] = rng.integers(0, 11)
    if H > 8:
        pan[-2:, -3:] = 255
    obj = rng.integers(0, 40, size=k).astype(np.int64) if k else np.zeros(0, np.int64)
    ref = opp.get_unified_pan_result([seg, seg], [pan, pan], [cls_ind, cls_ind], [obj, obj], 4 * 64 * 64, ['a', 'b'])
    u = pp.PanopticUnifier(dev)
    sd, pd = torch.from_numpy(seg).to(dev), torch.from_numpy(pan).to(dev)
    res = u.get_unified_pan_result([sd, sd], [pd, pd], [cls_ind, cls_ind], [obj, obj], 4 * 64 * 64, ['a', 'b'])
    for n in ('a', 'b'):
        assert np.array_equal(res[n], ref[n]), n
    # without object ids
    ref2 = opp.get_unified_pan_result([seg], [pan], [cls_ind], None, 500, ['a'])
    res2 = u.get_unified_pan_result([sd], [pd], [cls_ind], None, 500, ['a'])
    assert np.array_equal(res2['a'], ref2['a'])


class _Colors:
    """deterministic stand-in for panopticapi's IdGenerator (which draws random shades of the category colour): distinct
    colours in call order, with one deliberate repeat so that two segments share a colour id"""

    def __init__(self):
        self.n = 0

    def get_color(self, cat_id):
        self.n += 1
        k = self.n if self.n != 7 else 3           # the 7th call repeats the 3rd colour
        return [int(cat_id) * 7 % 256, k % 256, (k * 37) % 256]


def _pan2ch_clip(rng, H, W, nfr):
    clip = []
    for f in range(nfr):
        seg = np.ascontiguousarray(rng.integers(0, 11, size=((H + 7) // 8, (W + 7) // 8)).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:H, :W])
        ins = np.zeros((H, W), np.uint8); obj = seg.copy()
        for i in range(6):
            h, w = int(rng.integers(3, H // 2)), int(rng.integers(3, W // 2))
            y, x = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
            seg[y:y + h, x:x + w] = 11 + i % 3; ins[y:y + h, x:x + w] = i + 1; obj[y:y + h, x:x + w] = 1 + (i + f) % 5
        seg[:2, :3] = 255; obj[:2, :3] = 255
        clip.append(np.stack([seg, ins, obj], -1))
    return clip


def test_oracle_converter_runs_and_counts_by_colour():
    clip = _pan2ch_clip(np.random.default_rng(3), 32, 48, 2)
    ann, pans = opp.converter_2ch_track_core(clip, _Colors())
    assert len(ann) == 2 and pans[0].shape == (32, 48, 3) and pans[0].dtype == np.uint8
    tot = sum(s['area'] for s in ann[0]['segments_info'])
    assert tot == int((clip[0][..., 0] != 255).sum())            # every non-void pixel is counted exactly once


@pytest.mark.gpu
@pytest.mark.parametrize('H,W', [(1024, 2048), (37, 53)])
def test_device_converter_matches_oracle(dev, H, W):
    from vps_amd import postprocess as pp
    clip = _pan2ch_clip(np.random.default_rng(H), H, W, 3)
    ann_r, pans_r = opp.converter_2ch_track_core(clip, _Colors())
    ann_d, pans_d = pp.TrackConverter(dev).convert([torch.from_numpy(c).to(dev) for c in clip], _Colors())
    for f in range(3):
        assert np.array_equal(pans_d[f], pans_r[f]), f
        assert ann_d[f] == ann_r[f], f


def test_png_name_follows_the_reference():
    from vps_amd import postprocess as pp
    assert pp.png_name('out/pan_pred', 'frankfurt_000000_001736_leftImg8bit.png') == 'out/pan_pred/frankfurt_000000_001736.png'
    assert pp.png_name('o', '0005_0025_frankfurt_000000_001736_newImg8bit.jpg') == 'o/0005_0025_frankfurt_000000_001736.png'


@pytest.mark.gpu
def test_inference_panoptic_video_writes_the_reference_files(dev, tmp_path):
    """the output side of tools/test_vpq.py:194-198 (cityscapes_vps.py:27-94): labelled-frame sampling, per-video conversion,
    pan_2ch / pan_pred PNGs and pred.json — against the oracle's converter applied to the same sampled frames"""
    import json
    from PIL import Image
    from vps_amd import postprocess as pp
    H, W, nvid, nfr = 64, 96, 2, 30
    rng = np.random.default_rng(5)
    frames = []
    for v in range(nvid):
        frames += _pan2ch_clip(rng, H, W, nfr)
    names = ['%04d_%04d_city_%06d_newImg8bit.png' % (v, f, f) for v in range(nvid) for f in range(nfr)]
    sampled = frames[4::5]; snames = names[4::5]                   # (labeled_fid // lambda_) :: lambda_  ->  12 labelled frames
    assert len(sampled) == 12
    pans, pj = pp.inference_panoptic_video([torch.from_numpy(f).to(dev) for f in frames], str(tmp_path), None, snames, n_video=nvid,
                                           color_generator=_Colors(), device=dev)
    gen = _Colors()
    ann_r, pans_r = [], []
    for v0 in range(0, 12, 6):                                     # one converter state (instance -> colour) per video
        a, p = opp.converter_2ch_track_core(sampled[v0:v0 + 6], gen)
        ann_r += a; pans_r += p
    assert pj == {'annotations': ann_r}
    assert json.load(open(tmp_path / 'pred.json')) == json.loads(json.dumps({'annotations': ann_r}))
    for i in range(12):
        assert np.array_equal(pans[i], pans_r[i])
        assert np.array_equal(np.asarray(Image.open(pp.png_name(str(tmp_path / 'pan_pred'), snames[i]))), pans_r[i])
        assert np.array_equal(np.asarray(Image.open(pp.png_name(str(tmp_path / 'pan_2ch'), snames[i]))), sampled[i])
