"""vps_amd.dataloader: the caller side of the import switch (INTEGRATION.md Level 1, round 6) - LookaheadLoader around the unmodified
tools/test_vpq.py loop, MMDataParallel / DataContainer stand-ins, build_dataloader. CPU: the protocol; GPU: the loop's outputs are
bitwise the plain per-frame calls' and every frame's image-only stages are enqueued exactly once, ahead of its call."""
import os

import numpy as np
import pytest
import torch

import vps_amd
from vps_amd import detector as D
from vps_amd import synth
from vps_amd.dataloader import DataContainer, LookaheadLoader, MMDataParallel, build_dataloader, scatter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Frames:
    """test-mode batches like the reference's loader yields them (one sample, img / img_meta / ref_img)"""

    def __init__(self, frames, H, W):
        self.frames, self.H, self.W = frames, H, W
        self.dataset = list(range(len(frames)))

    def __len__(self):
        return len(self.frames)

    def __iter__(self):
        for t, f in enumerate(self.frames):
            ref = (self.frames[t - 1] if t else f).clone()
            yield dict(img=[f], img_meta=[DataContainer([[synth.img_meta(self.H, self.W, 10001 + t)]], cpu_only=True)], ref_img=[ref])


def test_lookahead_loader_yields_in_order_and_announces_the_following_frames():
    frames = [torch.full((1, 3, 4, 8), float(t)) for t in range(5)]
    seen = []
    ld = LookaheadLoader(_Frames(frames, 4, 8), depth=2, device=None)
    assert len(ld) == 5 and ld.dataset == list(range(5))
    for t, data in enumerate(ld):
        assert data['img'][0] is frames[t]                                   # no device: the loader's own tensors
        board = list(D._BOARD)
        seen.append([int(a[0, 0, 0, 0]) for a, b in board])
        for a, b in board:
            assert int(b[0, 0, 0, 0]) == int(a[0, 0, 0, 0]) - 1             # each announced pair is (frame, its reference frame)
    assert seen == [[1, 2], [2, 3], [3, 4], [4], []]
    assert D._BOARD == []                                                     # cleared at the end of the iteration
    # depth 0 announces nothing; an exception in the consumer clears the board
    assert [list(D._BOARD) for _ in LookaheadLoader(_Frames(frames, 4, 8), depth=0, device=None)] == [[]] * 5
    with pytest.raises(RuntimeError):
        for data in LookaheadLoader(_Frames(frames, 4, 8), depth=2, device=None):
            raise RuntimeError('consumer failed')
    assert D._BOARD == []


def test_scatter_and_data_parallel_standins():
    dc = DataContainer([[dict(a=1)]], cpu_only=True)
    t = torch.zeros(2)
    out = scatter(dict(img=[t], img_meta=[dc], k=(t, 3)), torch.device('cpu'))
    assert out['img'][0] is t and out['img_meta'][0] == [dict(a=1)] and out['k'][0] is t and out['k'][1] == 3

    class Echo(torch.nn.Module):
        def forward(self, **kw):
            return kw
    got = MMDataParallel(Echo(), device_ids=[0])(return_loss=False, img=[t], img_meta=[dc])
    assert got['return_loss'] is False and got['img_meta'] == [[dict(a=1)]]
    assert got['img'][0].shape == t.shape


def test_build_dataloader_collates_one_sample_per_batch_and_looks_ahead():
    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 3

        def __getitem__(self, i):                                             # what MultiScaleFlipAug + Collect produce in test mode
            return dict(img=[torch.full((3, 4, 8), float(i))], img_meta=[DataContainer(dict(iid=10001 + i), cpu_only=True)],
                        ref_img=[torch.full((3, 4, 8), float(max(i - 1, 0)))])
    ld = build_dataloader(DS(), imgs_per_gpu=1, workers_per_gpu=0, dist=False, shuffle=False)
    assert isinstance(ld, LookaheadLoader) and len(ld) == 3
    items = list(ld)
    assert [tuple(d['img'][0].shape) for d in items] == [(1, 3, 4, 8)] * 3
    assert [float(d['img'][0].flatten()[0]) for d in items] == [0.0, 1.0, 2.0]
    assert items[1]['img_meta'][0].data[0][0]['iid'] == 10002 and items[1]['img_meta'][0].cpu_only     # test_vpq.py:42-43 reads it this way
    with pytest.raises(AssertionError):
        build_dataloader(DS(), imgs_per_gpu=2)


@pytest.mark.gpu
def test_unmodified_vpq_loop_gets_the_cross_frame_pipeline(dev):
    """the loop of tools/test_vpq.py:41-63 over LookaheadLoader + MMDataParallel: outputs bitwise the plain per-frame calls', and
    every frame's image-only stages are enqueued ONCE and BEFORE its own call (frames 1.. are prefetched; only frame 0 is not)"""
    from vps_amd import hip, nhwc
    H, W, n = 128, 256, 7
    cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'cityscapes', 'fusetrack.py'))
    old = nhwc.DEFAULT_PREC
    nhwc.DEFAULT_PREC = hip.PREC_F16X3
    try:
        m = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
        synth.load_synth(m, 0)
        m.ensure_packed(dev)
    finally:
        nhwc.DEFAULT_PREC = old
    fr = synth.synth_clip(H, W, 3, 0)
    frames = [fr[t % 3].clone().pin_memory() for t in range(n)]              # HOST tensors, like a DataLoader with pin_memory yields them
    keys = ('panoptic_det_obj_ids', 'panoptic_outputs', 'fcn_outputs', 'panoptic_cls_prob', 'panoptic_cls_inds')
    seq = []
    for t in range(n):
        out = m(return_loss=False, rescale=True, img=[frames[t].to(dev)], img_meta=[[synth.img_meta(H, W, 10001 + t)]],
                ref_img=[frames[t - 1 if t else 0].to(dev)])
        seq.append({k: out[2][k].cpu().numpy().copy() for k in keys})
    m._cache = None; m._pf = None; m.reset_tracker()
    calls = []
    orig = m._enqueue_image_stages

    def counted(nimg, nref, main, **kw):
        calls.append((float(nimg[0, 0, 0, 0]), len(got)))                    # which frame (by content), during which loop iteration
        return orig(nimg, nref, main, **kw)
    m._enqueue_image_stages = counted
    wrapped = MMDataParallel(m, device_ids=[dev.index or 0])
    got = []
    try:
        for data in LookaheadLoader(_Frames(frames, H, W), depth=2, device=dev):
            with torch.no_grad():
                result = wrapped(return_loss=False, rescale=True, **data)       # tools/test_vpq.py:46
            got.append({k: result[2][k].cpu().numpy().copy() for k in keys})
    finally:
        del m._enqueue_image_stages
    assert len(got) == n and m._pf is None and D._BOARD == []
    for t in range(n):
        for k in keys:
            assert np.array_equal(seq[t][k], got[t][k]), (t, k)
    assert len(calls) == n, calls                                              # one enqueue per frame: nothing computed twice, nothing wasted
    ahead = [it for _, it in calls]
    assert ahead[0] == 0 and all(it < t for t, it in enumerate(ahead) if t >= 1), calls      # frame t >= 1 was enqueued during an EARLIER call
