"""BASELINE config 5: the ResNet-101 backbone variant (mmdet/models/backbones/resnet.py:361, arch_settings[101]) of the
FuseTrack model, VIPER-scale frames (1080x1920 -> Pad(32) -> 1088x1920), bf16 arithmetic.

* golden: tests/golden/make_golden.py r101 runs the REAL reference PanopticFuseTrack built with depth=101 (935 state_dict keys)
  on a 2-frame 128x256 clip -> tests/golden/fusetrack_r101_clip.npz;
* CPU: the oracle (depth=101) reproduces it, and vps_amd's state_dict equals the reference module tree's;
* GPU: the HIP path reproduces it in the fp32-grade modes (stage tensors within 2e-3, the detections / ids of the golden frame
  up to reorderings of detections whose scores differ by less than the fp32 noise — the exact-fp32 kernels reproduce the
  listing strictly). In the plain bf16 mode ("bf16 in, fp32 accumulate": the arithmetic config 5 names; operands rounded to 8
  significand bits) the image-only stages (FlowNet2 flow, ResNet-101 + FPN) stay within 3e-2 of the fp32 reference; behind
  them the RANDOM synthetic weights amplify the rounding (flow-guided warping, deformable offsets: semantic logits differ by
  ~0.3 of their range, 30 % of the argmax pixels) — reported, not asserted: trained weights are needed for a meaningful bf16
  end-to-end criterion (VPQ), which this container cannot provide;
* GPU, full scale: 1088x1920 frames run through the ResNet-101 model in bf16 and f16x3 (shapes, instance limits, image-only
  stages compared).
The detections / ids of the golden frames are compared up to reorderings of near-tied scores and one relabelling of the track ids
(the strict listing is printed): with these synthetic heads a borderline proposal flips on last-bit differences of the build.
"""
import json
import os

import numpy as np
import pytest
import torch

import vps_amd
from vps_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden', 'fusetrack_r101_clip.npz')
CFG = os.path.join(ROOT, 'configs', 'viper', 'fusetrack_r101.py')


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-12))


def _build(prec=None):
    from vps_amd import nhwc
    cfg = vps_amd.Config.fromfile(CFG)
    old = nhwc.DEFAULT_PREC
    if prec is not None:
        nhwc.DEFAULT_PREC = nhwc.PREC_NAMES[prec]
    try:
        m = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
        sd = synth.load_synth(m, 0)
        if prec is not None:
            m.ensure_packed(torch.device('cuda:0'))
    finally:
        nhwc.DEFAULT_PREC = old
    return m, sd


def test_r101_state_dict_matches_reference_module_tree():
    g = np.load(GOLD)
    manifest = json.loads(bytes(g['state_dict_manifest']).decode())
    cfg = vps_amd.Config.fromfile(CFG)
    m = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    ours = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert ours == manifest and len(ours) == 935
    assert sum(1 for k in ours if k.startswith('backbone.layer3.') and k.endswith('conv1.weight')) == 23


def test_r101_oracle_matches_reference():
    from oracle.fusetrack import FuseTrackOracle
    g = np.load(GOLD)
    H, W, n, seed = [int(v) for v in g['meta']]
    cfg = vps_amd.Config.fromfile(CFG)
    m = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    sd = synth.synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed)
    o = FuseTrackOracle(sd, depth=101)
    fr = synth.synth_clip(H, W, n, seed)
    with torch.no_grad():
        for t in range(n):
            r = o.simple_test(fr[t], fr[t - 1 if t else 0], t == 0, return_aux=True)
            p = 'f%d.' % t
            assert _rel(r['pre_neck'][0][0, :8].numpy(), g[p + 'fpn_p2']) < 1e-4
            assert _rel(r['fcn_score'][0].numpy(), g[p + 'fcn_score']) < 5e-4
            assert _rel(r['det']['cls_score'].numpy(), g[p + 'cls_score']) < 5e-4
            for k in ('panoptic_cls_inds', 'panoptic_det_labels', 'panoptic_det_obj_ids'):
                assert np.array_equal(r[k].numpy(), g[p + k]), k
            assert (r['panoptic_outputs'].numpy().astype(np.uint8) != g[p + 'panoptic_outputs']).mean() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize('prec', ['f32', 'f16x3', 'bf16'])
def test_r101_hip_matches_reference_golden(dev, prec):
    g = np.load(GOLD)
    H, W, n, seed = [int(v) for v in g['meta']]
    m, _ = _build(prec)
    fr = synth.synth_clip(H, W, n, seed)
    tol = 3e-2 if prec == 'bf16' else 2e-3
    asserted = ('fpn_p2', 'fpn_p5', 'flow') if prec == 'bf16' else ('fpn_p2', 'fpn_p5', 'neck_p2', 'fcn_score', 'flow')
    id_map, id_back = {}, {}
    for t in range(n):
        out = m(return_loss=False, rescale=True, img=[fr[t].to(dev)], img_meta=[[synth.img_meta(H, W, 10000 + t + 1)]],
                ref_img=[fr[t - 1 if t else 0].to(dev)])
        torch.cuda.synchronize()
        p = 'f%d.' % t
        a = m._aux
        errs = dict(fpn_p2=_rel(a['levels'][0].to_nchw().cpu().numpy()[0, :8], g[p + 'fpn_p2']),
                    fpn_p5=_rel(a['levels'][3].to_nchw().cpu().numpy()[0], g[p + 'fpn_p5']),
                    neck_p2=_rel(a['neck_out'][0].to_nchw().cpu().numpy()[0, :8], g[p + 'neck_out_p2']),
                    fcn_score=_rel(a['fcn_score'].to_nchw().cpu().numpy()[0], g[p + 'fcn_score']),
                    flow=_rel(a['flow'].to_nchw().cpu().numpy()[0][:, ::2, ::2], g[p + 'flow_full']))
        r = {k: v.cpu().numpy() for k, v in out[2].items()}
        dsem = float((r['fcn_outputs'] != g[p + 'fcn_outputs']).mean())
        print('[r101 %s] frame %d: %s sem mismatch %.4f%% kept %d (golden %d)' % (prec, t, {k: '%.2e' % v for k, v in errs.items()}, 100 * dsem,
                                                                               len(r['panoptic_cls_inds']), len(g[p + 'panoptic_cls_inds'])))
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'config5_report.txt'), 'a') as f:
            f.write('r101 128x256 %s frame %d %s sem_mismatch %.5f\n' % (prec, t, errs, dsem))
        for k in asserted:
            assert errs[k] < tol, (k, errs[k])
        assert all(np.isfinite(v) and v < 1.0 for v in errs.values()), errs
        if prec == 'bf16':
            continue
        assert dsem < 1e-3
        gc, gp = g[p + 'panoptic_cls_inds'], g[p + 'panoptic_cls_prob']
        strict = all(np.array_equal(r[k], g[p + k]) for k in ('panoptic_cls_inds', 'panoptic_det_labels', 'panoptic_det_obj_ids'))
        print('[r101 %s] frame %d listing identical to the golden: %s' % (prec, t, strict))
        assert float((r['panoptic_outputs'] != g[p + 'panoptic_outputs']).mean()) < (1e-3 if strict else 5e-2)
        # same kept detections (class, score within 2e-3); the listing order may differ between near-tied scores, and one borderline
        # box more or less in front of the NMS shifts every later track id by one: ids are compared up to ONE relabelling of the clip
        used = set()
        for i in range(len(r['panoptic_cls_inds'])):
            d = np.abs(gp - r['panoptic_cls_prob'][i]) + 1e6 * (gc != r['panoptic_cls_inds'][i])
            for j in used:
                d[j] = 1e9
            j = int(np.argmin(d))
            assert d[j] < 2e-3, (i, d[j])
            used.add(j)
            a, b = int(r['panoptic_det_obj_ids'][i]), int(g[p + 'panoptic_det_obj_ids'][j])
            assert id_map.setdefault(a, b) == b and id_back.setdefault(b, a) == a, (t, i, a, b)
        assert len(used) == len(gc)


@pytest.mark.gpu
def test_r101_viper_scale_bf16_against_f16x3(dev):
    H, W = 1088, 1920
    fr = [f.to(dev) for f in synth.synth_clip(H, W, 2, 0)]
    res = {}
    for prec in ('f16x3', 'bf16'):
        m, _ = _build(prec)
        rec = []
        for t in range(2):
            out = m(return_loss=False, rescale=True, img=[fr[t]], img_meta=[[synth.img_meta(H, W, 10000 + t + 1)]], ref_img=[fr[t - 1 if t else 0]])
            torch.cuda.synchronize()
            a = m._aux
            assert out[2]['panoptic_outputs'].shape == (1, H, W)
            rec.append(dict(sem=out[2]['fcn_outputs'].cpu().numpy(), k=len(out[2]['panoptic_cls_inds']),
                            p2=a['levels'][0].to_nchw().cpu().numpy(), fcn=a['fcn_score'].to_nchw().cpu().numpy(), flow=a['flow'].to_nchw().cpu().numpy()))
        res[prec] = rec
        del m
        torch.cuda.empty_cache()
    for t in range(2):
        a, b = res['bf16'][t], res['f16x3'][t]
        e = dict(p2=_rel(a['p2'], b['p2']), fcn=_rel(a['fcn'], b['fcn']), flow=_rel(a['flow'], b['flow']))
        dsem = float((a['sem'] != b['sem']).mean())
        print('[r101 1088x1920] frame %d bf16 vs f16x3: %s sem mismatch %.4f%%, instances %d / %d' % (t, {k: '%.2e' % v for k, v in e.items()}, 100 * dsem, a['k'], b['k']))
        with open(os.path.join(ROOT, 'gpurun_out', 'config5_report.txt'), 'a') as f:
            f.write('r101 1088x1920 frame %d bf16-vs-f16x3 %s sem_mismatch %.5f instances %d/%d\n' % (t, e, dsem, a['k'], b['k']))
        assert 0 < b['k'] <= 244 and 0 < a['k'] <= 244
        assert e['p2'] < 3e-2 and e['flow'] < 3e-2, e                 # image-only stages; the rest is reported (module docstring)
        assert all(np.isfinite(v) and v < 1.0 for v in e.values()), e
