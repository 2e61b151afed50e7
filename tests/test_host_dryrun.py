"""CPU coverage of the HOST side of the HIP path: the three detectors run a 3-frame clip with every kernel launch replaced by a
recording stub (outputs are meaningless; there is no CPU compute path in the product — the stubs live in this test only).
Checks: the Python control flow of simple_test (fusion / tracking branches, reference-feature cache, the host side of MaskROI /
MaskRemoval / the tracker, the two host reads of a frame, result assembly) executes, results have the reference's keys and shapes, every symbol the host calls is
declared in include/vps_hip.h, and the persistent workspace stops growing after the first frames (up to the small per-detection buffers)."""
import ctypes
import os

import numpy as np
import pytest
import torch

import vps_amd
from vps_amd import hip, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _RecordingLib:
    def __init__(self):
        self.called = set()

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        self.called.add(name)
        impl = type(self).__dict__.get('_' + name)
        return (lambda *a, **k: impl(self, *a, **k)) if impl is not None else (lambda *a, **k: 0)

    # the few launches whose results steer host control flow get plausible values
    def _vps_nms_batched(self, boxes, nb, nmax, counts, thr, mask, keep, nkeep, stream):
        cnt = (ctypes.c_int32 * nb).from_address(counts.value)
        kp = (ctypes.c_int32 * (nb * nmax)).from_address(keep.value)
        nk = (ctypes.c_int32 * nb).from_address(nkeep.value)
        for b in range(nb):
            k = min(cnt[b], 40)
            nk[b] = k
            for i in range(k):
                kp[b * nmax + i] = i
        return 0

    def _vps_row_softmax(self, inp, out, rows, cols, mode, stream):
        r = np.random.default_rng(rows).random((rows, cols)).astype(np.float32)
        r /= r.sum(1, keepdims=True)
        r[:, min(3, cols - 1)] += (np.arange(rows) % 7 == 0) * 0.9
        if mode == 1:
            r = np.log(r)
        ctypes.memmove(out.value, r.ctypes.data, r.nbytes)
        return 0

    def _vps_maskroi_finish(self, dets, cand, m_in, keep, nkeep, nc, max_det, kcap, res, stream):
        K = 12
        r = np.zeros(8 + 8 * kcap, dtype=np.float32)
        r[0], r[1], r[2], r[4] = K, 40, 40, 1000
        rows = r[8:8 + 8 * K].reshape(K, 8)
        rows[:, 1] = 10 + 18 * np.arange(K); rows[:, 2] = 20 + 5 * (np.arange(K) % 4)
        rows[:, 3] = rows[:, 1] + 30; rows[:, 4] = rows[:, 2] + 40
        rows[:, 5] = 0.95 - 0.02 * np.arange(K); rows[:, 6] = 1 + np.arange(K) % (nc - 1); rows[:, 7] = np.arange(K)
        ctypes.memmove(res.value, r.ctypes.data, r.nbytes)
        return 0

    def _vps_pan_instances(self, order, flags, rows, tbox, n, cm, nclass, inst, keep_out, k_out, stream):
        ko = (ctypes.c_int32 * 4).from_address(k_out.value)
        kp = (ctypes.c_int32 * n).from_address(keep_out.value)
        k = 0
        for p in range(n):
            if flags is None or flags.value is None or (ctypes.c_int32 * 1).from_address(flags.value + 4 * p)[0]:
                kp[k] = p if order is None or order.value is None else (ctypes.c_int32 * 1).from_address(order.value + 4 * p)[0]
                k += 1
        ko[0], ko[1], ko[2] = max(k, 1), int(k > 0), 0
        return 0

    def _vps_track_assign(self, comp, K, M, emb, E, box, ldb, label, pe, pb, pl, scratch, ids, m_out, stream):
        idv = (ctypes.c_int32 * K).from_address(ids.value)
        for i in range(K):
            idv[i] = i if i < M else M + (i - M)
        (ctypes.c_int32 * 1).from_address(m_out.value)[0] = max(M, K)
        return 0

    def _vps_frame_tail(self, kinfo, keep, ids, mem_count, f16_status, nslots, K, kcap, tail, stream):
        t = (ctypes.c_int32 * (8 + 2 * kcap)).from_address(tail.value)
        ki = (ctypes.c_int32 * 4).from_address(kinfo.value)
        for i in range(4):
            t[i] = ki[i]
        t[4] = (ctypes.c_int32 * 1).from_address(mem_count.value)[0] if mem_count is not None and mem_count.value else 0
        t[5] = 0
        kp = (ctypes.c_int32 * max(K, 1)).from_address(keep.value)
        for i in range(K):
            t[8 + i] = kp[i]
            if ids is not None and ids.value:
                t[8 + kcap + i] = (ctypes.c_int32 * 1).from_address(ids.value + 4 * i)[0]
        return 0

    def _vps_mask_removal_dep(self, *a):
        flags, n = a[11], a[5]
        for i in range(n):
            (ctypes.c_int32 * 1).from_address(flags.value + 4 * i)[0] = 1
        return 0

    def _vps_mask_removal_hist(self, *a):
        flags, n = a[14], a[7]
        for i in range(n):
            (ctypes.c_int32 * 1).from_address(flags.value + 4 * i)[0] = 1
        return 0

    def _vps_mask_level(self, *a):
        flags, nlevel, level = a[-2], a[6], a[5]
        lv = (ctypes.c_int32 * nlevel).from_address(level.value)
        for i in range(nlevel):
            (ctypes.c_int32 * 1).from_address(flags.value + 4 * lv[i])[0] = 1
        return 0


class _FakeCuda(torch.Tensor):
    is_cuda = property(lambda s: True)


@pytest.mark.parametrize('variant,keys', [
    ('fusetrack', ['fcn_outputs', 'panoptic_cls_inds', 'panoptic_cls_prob', 'panoptic_det_labels', 'panoptic_det_obj_ids', 'panoptic_outputs']),
    ('fuse', ['fcn_outputs', 'panoptic_cls_inds', 'panoptic_cls_prob', 'panoptic_outputs']),
    ('track', ['fcn_outputs', 'panoptic_cls_inds', 'panoptic_cls_prob', 'panoptic_det_labels', 'panoptic_det_obj_ids', 'panoptic_outputs'])])
def test_host_control_flow_with_stubbed_launches(monkeypatch, variant, keys):
    lib = _RecordingLib()
    monkeypatch.setattr(hip, 'load', lambda: lib)
    monkeypatch.setattr(hip, 'ptr', lambda t: None if t is None else ctypes.c_void_p(t.data_ptr()))
    monkeypatch.setattr(hip, 'stream_ptr', lambda: None)
    monkeypatch.setattr(hip, 'conv2d', lambda d: lib.called.add('vps_conv2d'))
    cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'cityscapes', variant + '.py'))
    m = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    m.overlap_streams = False                      # torch.cuda streams need a device
    synth.load_synth(m, 0)
    H, W = 128, 256
    fr = synth.synth_clip(H, W, 3, 0)
    sizes, pools = [], []
    for t in range(3):
        out = m(return_loss=False, rescale=True, img=[fr[t].as_subclass(_FakeCuda)], img_meta=[[synth.img_meta(H, W, 10001 + t)]],
                ref_img=[fr[t - 1 if t else 0]])
        assert len(out) == 3 and sorted(out[2].keys()) == keys
        assert tuple(out[2]['panoptic_outputs'].shape) == (1, H, W) and tuple(out[2]['fcn_outputs'].shape) == (1, H, W)
        n = out[2]['panoptic_cls_inds'].numel()
        assert all(out[2][k].numel() == n for k in keys if k.startswith('panoptic_') and k != 'panoptic_outputs')
        assert isinstance(out[0], list if variant == 'fuse' else dict)
        sizes.append(m._ws.nbytes())
        pools.append((m._ws.pool.total, m._ws.pool.blocks))
        assert not m._ws._live, 'temporary maps still taken at the end of the frame: %s' % sorted(e[2] for e in m._ws._live.values())
    assert abs(sizes[2] - sizes[1]) < 1e-3 * sizes[1], 'the workspace must be persistent (only the per-detection buffers may resize)'
    # liveness-based reuse: the block pool reaches its size in the first frame and serves every later frame from
    # its free lists (same take / give sequence -> same blocks -> same addresses: the cached conv descriptors stay valid)
    assert pools[1] == pools[0] and pools[2] == pools[0], pools
    if variant == 'fusetrack':
        # one stream here (no device): persistent + pooled bytes of the frame graph, scaled from 128x256 to 1024x2048 (x64) and counted
        # over the image-proportional buffers only. Round 4 held 11.1 GB in the main workspace alone (35.7 GB with the prefetch ring);
        # the multi-stream figure is asserted on the GPU (tests/test_fullsize_gpu.py) and reported as config.workspace_GB by bench.py
        big = sum(t.numel() * t.element_size() for k, t in m._ws.bufs.items() if t.dim() == 4 and t.shape[1] * t.shape[2] >= 64 and not k.startswith('mask.'))
        est = 64 * (big + m._ws.pool.total) / 1e9
        print('single-stream workspace estimate at 1024x2048: %.2f GB (pool %.2f GB in %d blocks)' % (est, 64 * m._ws.pool.total / 1e9, m._ws.pool.blocks))
        assert est < 6.0, est
    undeclared = sorted(n for n in lib.called if n not in hip.SYMBOLS)
    assert not undeclared, undeclared
    assert 'vps_conv2d' in lib.called and ('vps_correlation' in lib.called) == (variant != 'track')


@pytest.mark.parametrize('prec,expect_fused', [('f16x3', True), ('f32', False)])
def test_semantic_head_hands_groupnorm_sums_to_the_unsplit_deformable_layers(monkeypatch, prec, expect_fused):
    """Host logic of upsnetFPN.py:39-81 on this package (kernel launches recorded, not executed): a deformable conv that is not
    split over K carries vps_conv_desc.gn_stats / gn_cpg / gn_rep (its own zeroed slot) and is followed by vps_groupnorm_apply on
    that slot; a split one (small levels) carries none and is followed by the two-pass vps_groupnorm_relu; the exact-fp32 mode never
    asks for the sums (its deformable kernel has no such epilogue)."""
    from vps_amd import nhwc
    lib = _RecordingLib()
    log = []
    monkeypatch.setattr(hip, 'load', lambda: lib)
    monkeypatch.setattr(hip, 'ptr', lambda t: None if t is None else ctypes.c_void_p(t.data_ptr()))
    monkeypatch.setattr(hip, 'stream_ptr', lambda: None)
    monkeypatch.setattr(hip, 'conv2d', lambda d: log.append(('conv', bool(d.offset), d.ksplit, d.gn_stats, d.gn_cpg, d.gn_rep, d.cout, d.tile_n, d.N * d.Ho * d.Wo)))
    monkeypatch.setattr(_RecordingLib, '_vps_groupnorm_apply', lambda self, *a: log.append(('apply', a[-3].value, a[-2])) or 0, raising=False)
    monkeypatch.setattr(_RecordingLib, '_vps_groupnorm_relu', lambda self, *a: log.append(('relu', a[-2].value)) or 0, raising=False)
    monkeypatch.setattr(nhwc, 'DEFAULT_PREC', nhwc.PREC_NAMES[prec])
    monkeypatch.setattr(nhwc, 'f16_status', lambda device: torch.zeros(nhwc.F16_SLOTS, dtype=torch.int32))
    cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'cityscapes', 'fusetrack.py'))
    head = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg).panopticFPN
    ws = nhwc.Workspace(torch.device('cpu'))
    levels = [nhwc.FMap(torch.zeros(1, 128 >> l, 256 >> l, 256)) for l in range(4)]       # P2 128x256 (256 tiles x 2: unsplit) ... P5 16x32
    head.run(levels, ws)
    dcn = [(i, e) for i, e in enumerate(log) if e[0] == 'conv' and e[1]]
    assert len(dcn) == 12 and any(e[7] == 256 for _, e in dcn) == (prec == 'f16x3')
    slots = set()
    for i, (_, _, ksplit, gn_stats, gn_cpg, gn_rep, cout, tile_n, M) in dcn:
        nxt = log[i + 1]
        # 256 output channels on >= 256 pixel tiles: the 128 x 256 block of the split-fp16 mode (nhwc.DCN256); the exact mode and the
        # 128-channel layers keep the 128-column tile
        assert tile_n == (256 if (prec == 'f16x3' and cout == 256 and (M + 127) // 128 >= 256) else 128), (cout, M, tile_n)
        if expect_fused and ksplit == 1:
            assert gn_stats and gn_cpg == cout // 32 and gn_rep == nhwc.GN_REP
            assert nxt[0] == 'apply' and nxt[1] == gn_stats and nxt[2] == nhwc.GN_REP
            slots.add(gn_stats)
        else:
            assert not gn_stats and nxt[0] == 'relu'
    unsplit = sum(1 for _, e in dcn if e[2] == 1)
    assert unsplit >= 3                                              # at least the three layers of the P2 tower
    assert len(slots) == (unsplit if expect_fused else 0)            # one slot per layer, never shared
    if expect_fused:
        stats = ws.bufs['sem.gnstats']
        assert stats.dtype == torch.float64 and tuple(stats.shape) == (12, nhwc.GN_REP, 64)
        assert all(stats.data_ptr() <= s < stats.data_ptr() + stats.numel() * 8 and (s - stats.data_ptr()) % (nhwc.GN_REP * 64 * 8) == 0 for s in slots)
