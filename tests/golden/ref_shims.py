"""Import shims that let the REAL reference code (/root/reference, read-only) run on this CPU-only container.

Used only by tests/golden/make_golden.py to generate golden vectors; never imported by the product or by the tests
that run on the GPU box (the reference does not exist there).

What is shimmed, and with what:
  * missing third-party packages (mmcv, pycocotools, cv2, easydict-based tools.config, terminaltables, ...): inert
    stubs, plus the handful of functions the inference path really calls (mmcv.cnn init helpers, cv2.resize ->
    oracle.ops.cv2_resize_linear).
  * the reference's CUDA-only extension modules (correlation_cuda, resample2d_cuda, channelnorm_cuda, roi_align_cuda,
    deform_conv_cuda, nms_cuda, gpu_nms): Python classes with the reference's signatures backed by oracle.ops — these
    are the "parity unpinned" operators; everything ABOVE them in the reference (all Python module code) runs as is.
  * `.cuda()` / torch.cuda.current_device(): no-ops / 'cpu' so the hard-coded device moves in flow_modules.py:132-146,
    track_head.py:77-81,124 and anchor_generator.py:55 work on the CPU.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Anything:
    """callable / subclassable placeholder"""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]          # used as a bare decorator
        return _Anything()

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Anything()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Anything


def _stub(name, **attrs):
    m = _StubModule(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    parent, _, child = name.rpartition('.')
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def _pkg(name, path, lenient=False):
    """a package object whose __init__ is NOT executed but whose submodules resolve to the real reference files"""
    m = (_StubModule if lenient else types.ModuleType)(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    parent, _, child = name.rpartition('.')
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def install():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import ops as O

    # ---- device no-ops -----------------------------------------------------------------------------------
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.current_device = lambda: 0

    def _cpu_factory(fn):
        def f(*a, **k):
            if isinstance(k.get('device'), int):
                k['device'] = 'cpu'
            return fn(*a, **k)
        return f
    torch.ones = _cpu_factory(torch.ones)
    torch.zeros = _cpu_factory(torch.zeros)
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    _orig_get_device = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: 'cpu'
    _orig_to = torch.Tensor.to

    def _to(self, *a, **k):
        k.pop('non_blocking', None)
        a = tuple(x for x in a if not (isinstance(x, bool)))
        return _orig_to(self, *a, **k)
    torch.Tensor.to = _to

    # ---- third-party stubs ---------------------------------------------------------------------------------
    def _init(fn):
        def f(module, *a, **k):
            return None
        return f
    mmcv = _stub('mmcv')
    _stub('mmcv.cnn', xavier_init=_init('x'), kaiming_init=_init('k'), normal_init=_init('n'), constant_init=_init('c'),
          uniform_init=_init('u'))
    _stub('mmcv.runner', load_checkpoint=lambda *a, **k: None)
    _stub('mmcv.parallel'); _stub('mmcv.utils')
    _stub('pycocotools'); _stub('pycocotools.mask'); _stub('pycocotools.coco'); _stub('pycocotools.cocoeval')
    _stub('terminaltables'); _stub('imagecorruptions'); _stub('matplotlib'); _stub('matplotlib.pyplot')
    _stub('cv2', resize=lambda src, dsize, **k: O.cv2_resize_linear(src, dsize))

    class _Cfg:
        class network:
            bbox_reg_weights = (10., 10., 5., 5.)      # tools/config/config.py:47

        class test:
            max_det = 100                              # tools/config/config.py:169
    _stub('tools'); _stub('tools.config'); _stub('tools.config.config', config=_Cfg)

    # ---- the reference package, real files, no package-level __init__ side effects -------------------------
    if REF not in sys.path:
        sys.path.insert(1, REF)
    _pkg('mmdet', REF + '/mmdet')
    _pkg('mmdet.utils', REF + '/mmdet/utils')
    importlib.import_module('mmdet.utils.registry')
    sys.modules['mmdet.utils'].Registry = sys.modules['mmdet.utils.registry'].Registry
    sys.modules['mmdet.utils'].build_from_cfg = sys.modules['mmdet.utils.registry'].build_from_cfg

    # mmdet.ops: CUDA extensions replaced by oracle-backed modules with the reference's Python signatures
    class RoIAlign(nn.Module):
        def __init__(self, out_size, spatial_scale, sample_num=0, use_torchvision=False):
            super().__init__()
            self.out_size = (out_size, out_size) if isinstance(out_size, int) else tuple(out_size)
            self.spatial_scale, self.sample_num = float(spatial_scale), int(sample_num)

        def forward(self, features, rois):
            return O.roi_align(features, rois, self.out_size[0], self.spatial_scale, self.sample_num)

    class DeformConv(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                     deformable_groups=1, bias=False):
            super().__init__()
            assert not bias and groups == 1 and deformable_groups == 1
            self.stride, self.padding, self.dilation = stride, padding, dilation
            self.weight = nn.Parameter(torch.zeros(out_channels, in_channels, kernel_size, kernel_size))

        def forward(self, x, offset):
            return O.deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation)

    def nms(dets, iou_thr, device_id=None):
        return O.nms_mmdet(dets, iou_thr)
    ops = _stub('mmdet.ops', RoIAlign=RoIAlign, DeformConv=DeformConv, nms=nms)

    # mmdet.core: real transforms / geometry / anchors / misc, decorators as no-ops
    core = _pkg('mmdet.core', REF + '/mmdet/core', lenient=True)
    _pkg('mmdet.core.bbox', REF + '/mmdet/core/bbox')
    _pkg('mmdet.core.anchor', REF + '/mmdet/core/anchor')
    _pkg('mmdet.core.utils', REF + '/mmdet/core/utils')
    tr = importlib.import_module('mmdet.core.bbox.transforms')
    geo = importlib.import_module('mmdet.core.bbox.geometry')
    ag = importlib.import_module('mmdet.core.anchor.anchor_generator')
    misc = importlib.import_module('mmdet.core.utils.misc')
    # anchor_generator.py:55 hard-codes device='cuda'
    ag.AnchorGenerator.grid_anchors.__defaults__ = (16, 'cpu')
    for n in ('delta2bbox', 'bbox2roi', 'roi2bbox', 'bbox2result', 'bbox2result_with_id', 'bbox2delta'):
        setattr(core, n, getattr(tr, n))
    core.bbox_overlaps = geo.bbox_overlaps
    core.AnchorGenerator = ag.AnchorGenerator
    core.multi_apply = misc.multi_apply

    def _deco(*a, **k):
        def wrap(fn):
            return fn
        return wrap
    core.auto_fp16 = _deco
    core.force_fp32 = _deco
    for n in ('build_assigner', 'build_sampler', 'anchor_target', 'multiclass_nms', 'mask_target', 'get_classes',
              'tensor2imgs', 'bbox_target', 'merge_aug_bboxes', 'merge_aug_masks', 'merge_aug_proposals', 'bbox_mapping'):
        setattr(core, n, _Anything())

    _stub('mmdet.datasets'); _stub('mmdet.datasets.pipelines'); _stub('mmdet.datasets.pipelines.flow_utils')

    # mmdet.models: real registry/builder + the hot-path files
    models = _pkg('mmdet.models', REF + '/mmdet/models')
    importlib.import_module('mmdet.models.registry')
    _stub('mmdet.models.losses')
    _stub('mmdet.models.plugins')
    importlib.import_module('mmdet.models.builder').build_loss = lambda cfg: _Anything()
    # flownet CUDA packages
    fm = _pkg('mmdet.models.flow_modules', REF + '/mmdet/models/flow_modules')

    class Correlation(nn.Module):
        def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1):
            super().__init__()
            self.a = (pad_size, kernel_size, max_displacement, stride1, stride2)

        def forward(self, in1, in2):
            return O.correlation(in1, in2, *self.a)

    class Resample2d(nn.Module):
        def __init__(self, kernel_size=1, bilinear=True):
            super().__init__()
            assert kernel_size == 1 and bilinear

        def forward(self, in1, in2):
            return O.resample2d(in1.contiguous(), in2)

    class ChannelNorm(nn.Module):
        def __init__(self, norm_deg=2):
            super().__init__()

        def forward(self, x):
            return O.channelnorm(x)
    for pk, cls in (('correlation', Correlation), ('resample2d', Resample2d), ('channelnorm', ChannelNorm)):
        _stub('mmdet.models.flow_modules.%s_package' % pk)
        _stub('mmdet.models.flow_modules.%s_package.%s' % (pk, pk), **{cls.__name__: cls})
    # utils package: real files, cython/CUDA nms replaced
    ut = _pkg('mmdet.models.utils', REF + '/mmdet/models/utils')
    _pkg('mmdet.models.utils.upsnet', REF + '/mmdet/models/utils/upsnet')
    _pkg('mmdet.models.utils.upsnet.bbox', REF + '/mmdet/models/utils/upsnet/bbox')
    _stub('mmdet.models.utils.upsnet.bbox.bbox')
    _stub('mmdet.models.utils.upsnet.nms')
    _stub('mmdet.models.utils.upsnet.nms.nms',
          gpu_nms_wrapper=lambda thresh, device_id: (lambda dets: O.nms_upsnet(dets, thresh)),
          py_nms_wrapper=_Anything(), cpu_nms_wrapper=_Anything())
    for sub, names in (('weight_init', ['bias_init_with_prob', 'kaiming_init', 'normal_init', 'uniform_init', 'xavier_init']),
                       ('norm', ['build_norm_layer']), ('conv_ws', ['ConvWS2d', 'conv_ws_2d']),
                       ('conv_module', ['ConvModule', 'build_conv_layer']), ('scale', ['Scale']),
                       ('deform_conv_with_offset', ['DeformConvWithOffset'])):
        mod = importlib.import_module('mmdet.models.utils.' + sub)
        for n in names:
            setattr(ut, n, getattr(mod, n))
    _stub('mmdet.models.utils.attention')
    ut.TCEA_Fusion = importlib.import_module('mmdet.models.utils.tcea_modules').TCEA_Fusion
    # flow modules __init__ content
    fmm = importlib.import_module('mmdet.models.flow_modules.flow_modules')
    fn2 = importlib.import_module('mmdet.models.flow_modules.flownet2')
    for n in ('LiteFlowNetCorr', 'WarpingLayer'):
        setattr(fm, n, getattr(fmm, n))
    fm.FlowNet2 = fn2.FlowNet2
    # component packages (no __init__ execution; import the files the config selects)
    for pkg in ('backbones', 'necks', 'extra_necks', 'panoptic', 'anchor_heads', 'roi_extractors', 'bbox_heads',
                'track_heads', 'mask_heads', 'detectors'):
        _pkg('mmdet.models.' + pkg, REF + '/mmdet/models/' + pkg)
    bb = importlib.import_module('mmdet.models.bbox_heads.bbox_head')
    sys.modules['mmdet.models.bbox_heads'].BBoxHead = bb.BBoxHead
    mods = {}
    for name in ('backbones.resnet', 'necks.fpn', 'extra_necks.bfp_tcea', 'panoptic.upsnetFPN', 'anchor_heads.anchor_head',
                 'anchor_heads.rpn_head', 'roi_extractors.single_level', 'bbox_heads.convfc_bbox_head',
                 'track_heads.track_head', 'mask_heads.fcn_mask_head', 'detectors.base', 'detectors.test_mixins',
                 'detectors.two_stage', 'detectors.panoptic_fusetrack', 'detectors.panoptic_fuse', 'detectors.panoptic_track'):
        mods[name] = importlib.import_module('mmdet.models.' + name)
    return mods
