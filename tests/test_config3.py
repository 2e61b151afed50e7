"""BASELINE config 3 plumbing (VERDICT r4 missing #1, #3): `load_checkpoint` with mmcv's semantics, tools/run_config3.py on synthetic
artefacts in the reference's file layout, and the REFERENCE'S OWN CALLER - tools/test_vpq.py:28-69 `single_gpu_test`, imported from
/root/reference under the golden-vector import shims - driving the vps_amd detector through an MMDataParallel-style wrapper."""
import ctypes
import importlib.util
import io
import json
import os
import subprocess
import sys
from collections import OrderedDict
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

import vps_amd
from vps_amd import hip, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'


def _model():
    cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'cityscapes', 'fusetrack.py'))
    return vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)


def test_load_checkpoint_semantics(tmp_path):
    """mmcv.runner.load_checkpoint as tools/test_vpq.py:135-137 uses it: 'state_dict' entry or bare OrderedDict, 'module.' prefix,
    strict=False reports (not raises) missing / unexpected keys, a shape mismatch raises, the checkpoint object is returned, and the
    packed HIP weights are invalidated by the in-place copy"""
    m = _model()
    sd = synth.synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, 3)
    f1 = str(tmp_path / 'latest.pth')
    torch.save({'meta': {'CLASSES': ('a', 'b')}, 'state_dict': OrderedDict(('module.' + k, v) for k, v in sd.items()), 'optimizer': {}}, f1)
    m.backbone._packed_device = torch.device('cpu')                  # as if it had been packed before
    ck = vps_amd.load_checkpoint(m, f1, map_location='cpu')
    assert ck['meta']['CLASSES'] == ('a', 'b')
    assert ck['_vps_load_report'] == dict(unexpected=[], missing=[], loaded=len(sd))
    assert m.backbone._packed_device is None
    got = m.state_dict()
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    # a bare OrderedDict without prefix; one key missing, one unexpected: reported, not fatal
    f2 = str(tmp_path / 'bare.pth')
    part = OrderedDict((k, v) for k, v in sd.items() if k != 'bbox_head.fc_cls.bias')
    part['not.in.the.model'] = torch.zeros(1)
    torch.save(part, f2)
    buf = io.StringIO()
    with redirect_stdout(buf):
        vps_amd.load_checkpoint(_model(), f2, map_location='cpu')
    assert 'unexpected key in source state_dict: not.in.the.model' in buf.getvalue()
    assert 'missing keys in source state_dict: bbox_head.fc_cls.bias' in buf.getvalue()
    with pytest.raises(RuntimeError, match='missing keys'):
        vps_amd.load_checkpoint(_model(), f2, map_location='cpu', strict=True)
    # no state_dict at all / a shape mismatch
    f3 = str(tmp_path / 'junk.pth')
    torch.save({'meta': {}}, f3)
    with pytest.raises(RuntimeError, match='No state_dict found'):
        vps_amd.load_checkpoint(_model(), f3)
    bad = OrderedDict(sd); bad['bbox_head.fc_cls.bias'] = torch.zeros(3)
    f4 = str(tmp_path / 'bad.pth')
    torch.save({'state_dict': bad}, f4)
    with pytest.raises(RuntimeError, match='While copying the parameter named bbox_head.fc_cls.bias'):
        vps_amd.load_checkpoint(_model(), f4)


def test_run_config3_check_only_on_synthetic_artefacts():
    """tools/run_config3.py --dry-run --check-only: a synthetic latest.pth ('module.' prefix + meta) and FlowNet2_checkpoint.pth.tar in
    the reference's layout load with no missing / unexpected key, the dataset json is listed; a missing artefact is ONE JSON error"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'run_config3.py'), '--dry-run', '--check-only'], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads(p.stdout.strip().splitlines()[-1])
    assert j['checkpoint']['missing'] == [] and j['checkpoint']['unexpected'] == [] and j['checkpoint']['loaded'] == 629
    assert j['dataset'] == dict(frames=16, videos=1, labelled_frames=4, img_prefix=j['dataset']['img_prefix'])
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'run_config3.py'), '--check-only', '--checkpoint', '/nonexistent/latest.pth'],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 2 and 'missing artefact' in json.loads(p.stdout.strip().splitlines()[-1])['error']


@pytest.mark.gpu
def test_run_config3_dry_run_end_to_end(dev):
    """the whole chain of BASELINE config 3 on the synthetic stand-ins (files -> ClipFeeder -> detector with the loaded checkpoint ->
    unifier -> PNGs + pred.json -> eval_vpq): the prediction scored against itself is 100 in the reference's metric"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'run_config3.py'), '--dry-run'], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    j = json.loads(p.stdout.strip().splitlines()[-1])
    assert j['run']['frames'] == 16 and j['run']['decodes'] == 16 and j['run']['png_files'] == 4
    assert j['vpq_final'][0] == 'vpq_all:100.0000', j['vpq_final']


# ---------------------------------------------------------------------------------------------------------------------------------
# The reference's own caller. mmcv is absent: `MMDataParallel` / `DataContainer` are restated here with the behaviour the call
# relies on (mmcv 0.2.x parallel/data_container.py, scatter_gather.py: a cpu_only container is unwrapped to its per-GPU list, tensors
# are moved to the device, lists are mapped element-wise); every kernel launch is the recording stub of tests/test_host_dryrun.py.
# ---------------------------------------------------------------------------------------------------------------------------------
class DataContainer:
    def __init__(self, data, stack=False, padding_value=0, cpu_only=False):
        self._data, self.stack, self.cpu_only = data, stack, cpu_only

    data = property(lambda s: s._data)


class MMDataParallel(torch.nn.Module):
    def __init__(self, module, device_ids=None, to_device=None):
        super().__init__()
        self.module, self.device_ids, self._to = module, device_ids, to_device

    def _scatter(self, obj):
        if isinstance(obj, DataContainer):
            return obj.data[0] if obj.cpu_only else self._scatter(obj.data[0])
        if torch.is_tensor(obj):
            return self._to(obj)
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._scatter(o) for o in obj)
        if isinstance(obj, dict):
            return {k: self._scatter(v) for k, v in obj.items()}
        return obj

    def forward(self, *inputs, **kwargs):
        return self.module(*self._scatter(inputs), **self._scatter(kwargs))


@pytest.mark.skipif(not os.path.isdir(REF), reason='the reference tree exists in the build container only')
def test_reference_single_gpu_test_drives_the_plugin():
    """in a child process: the import shims patch torch globally (`.cuda()` no-ops, ...)"""
    p = subprocess.run([sys.executable, os.path.abspath(__file__), 'ref_caller'], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, 'tests')])))
    assert p.returncode == 0 and 'REF_CALLER_OK' in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


def _ref_caller_main():
    from test_host_dryrun import _FakeCuda, _RecordingLib
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import ref_shims
    ref_shims.install()
    # INTEGRATION.md Level 1: the maintainer's one-line change - build_detector comes from vps_amd
    sys.modules['mmdet.models'].build_detector = vps_amd.build_detector
    sys.modules['mmcv.parallel'].MMDataParallel = MMDataParallel
    sys.modules['mmcv.runner'].load_checkpoint = vps_amd.load_checkpoint
    ref_shims._stub('tools.dataset')                                 # `from tools.dataset import *`: the evaluation helpers (main() only)
    spec = importlib.util.spec_from_file_location('ref_test_vpq', os.path.join(REF, 'tools', 'test_vpq.py'))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)                                     # the REAL tools/test_vpq.py
    assert ref.build_detector is vps_amd.build_detector and ref.MMDataParallel is MMDataParallel

    lib = _RecordingLib()
    hip.load = lambda: lib
    hip.ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    hip.stream_ptr = lambda: None
    hip.conv2d = lambda d: lib.called.add('vps_conv2d')
    cfg = vps_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'cityscapes', 'fusetrack.py'))
    model = ref.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)          # test_vpq.py:129-131
    model.overlap_streams = False
    synth.load_synth(model, 0)
    model.CLASSES = ('person', 'rider', 'car', 'truck', 'bus', 'train', 'motorcycle', 'bicycle')
    H, W = 128, 256
    frames = synth.synth_clip(H, W, 3, 0)

    class Loader:                         # what build_dataloader(...) yields in test mode (MultiScaleFlipAug + Collect, imgs_per_gpu=1)
        dataset = list(range(3))

        def __iter__(self):
            for t in range(3):
                meta = synth.img_meta(H, W, 10001 + t)
                meta['filename'] = 'data/cityscapes_vps/val/img_all/0000_%04d_frankfurt_000000_%06d_newImg8bit.png' % (t, t)
                yield dict(img=[frames[t]], img_meta=[DataContainer([[meta]], cpu_only=True)], ref_img=[frames[t - 1 if t else 0]])

    wrapped = ref.MMDataParallel(model, device_ids=[0], to_device=lambda t: t.as_subclass(_FakeCuda))      # test_vpq.py:149
    # (the loader wrapped the way vps_amd.build_dataloader does it: the reference's loop iterates the look-ahead loader unchanged)
    results, pano = ref.single_gpu_test(wrapped, vps_amd.LookaheadLoader(Loader(), depth=2, device=None))       # test_vpq.py:28-69
    assert len(results) == 3 and all(len(r) == 2 for r in results)
    assert pano['all_names'] == ['0000_%04d_frankfurt_000000_%06d_newImg8bit.png' % (t, t) for t in range(3)]
    for k in ('all_ssegs', 'all_panos'):
        assert len(pano[k]) == 3 and all(a.dtype == np.uint8 and a.shape == (H, W) for a in pano[k])
    assert len(pano['all_pano_cls_inds']) == 3 and len(pano['all_pano_obj_ids']) == 3
    assert all(len(a) == len(b) for a, b in zip(pano['all_pano_cls_inds'], pano['all_pano_obj_ids']))
    assert isinstance(results[0][0], dict)                            # bbox2result_with_id: id-keyed boxes
    assert 'vps_conv2d' in lib.called and 'vps_panoptic_combine_dev' in lib.called
    print('REF_CALLER_OK')


if __name__ == '__main__' and sys.argv[1:] == ['ref_caller']:
    _ref_caller_main()
