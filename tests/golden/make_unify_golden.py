"""Generates tests/golden/unify_cases.npz by running the REAL reference post-processing function
(/root/reference/tools/dataset/cityscapes_vps.py:162-226, CityscapesVPS.get_unified_pan_result) in the build container.

Only import shims are installed (easydict, cv2, pycocotools and whatever else the dataset package imports at module level
but the function never touches); the function body that runs is the reference's own. Run from the repo root:
    python tests/golden/make_unify_golden.py
The GPU box has no /root/reference: tests read the committed .npz only."""
import importlib
import os
import sys
import types
import warnings

import numpy as np

REF = '/root/reference'


class _EasyDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__getattr__ = lambda k: (_ for _ in ()).throw(AttributeError(k)) if k.startswith('__') else types.SimpleNamespace()
    sys.modules[name] = m
    return m


def load_reference_function(module='tools.dataset.cityscapes_vps', num_seg_classes=19, num_classes=9):
    sys.path.insert(0, REF)
    import collections, collections.abc
    for n in ('Sequence', 'Mapping', 'Iterable', 'MutableMapping'):     # the reference targets Python < 3.10
        if not hasattr(collections, n):
            setattr(collections, n, getattr(collections.abc, n))
    _stub('easydict', EasyDict=_EasyDict)
    for _ in range(40):                       # stub every missing third-party module the package imports at module level
        try:
            mod = importlib.import_module(module)
            break
        except ModuleNotFoundError as e:
            if e.name.startswith('tools'):
                raise
            _stub(e.name)
            for k in [k for k in sys.modules if k.startswith('tools')]:
                del sys.modules[k]
    cfg = importlib.import_module('tools.config.config').config
    cfg.dataset.num_seg_classes = num_seg_classes      # configs/cityscapes/test_cityscapes_1gpu.yaml:7-8; viper.py:100-101 (23 / 11)
    cfg.dataset.num_classes = num_classes
    cls = [v for k, v in vars(mod).items() if isinstance(v, type) and v.__module__ == mod.__name__ and
           'get_unified_pan_result' in vars(v)][0]
    return cls.get_unified_pan_result


def make_case(rng, H, W, k, with_obj, dup=False, void=False, big_ids=False, nstuff=11, nthing=8):
    """a panoptic map with k instances (ids nstuff..nstuff-1+k) over a stuff background, a semantic map that agrees / disagrees with
    the instances in controlled ways, class indices and object ids. nstuff = id_last_stuff + 1 (11 Cityscapes-VPS, 13 VIPER)."""
    seg = np.zeros((H, W), np.uint8)
    bs = max(8, H // 4)
    for y in range(0, H, bs):                 # stuff background in blocks (some classes end up below the area limit)
        for x in range(0, W, bs * 2):
            seg[y:y + bs, x:x + bs * 2] = rng.integers(0, nstuff)
    seg[:3, :5] = 7                           # a tiny stuff region
    pan = seg.copy()
    cls_ind = rng.integers(0, nthing, size=k)
    for i in range(k):
        h, w = int(rng.integers(4, max(5, H // 3))), int(rng.integers(4, max(5, W // 3)))
        y, x = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
        if i % 5 == 4 and i > 0:
            continue                          # an instance id that is absent from the map (idx != id - 11 afterwards)
        pan[y:y + h, x:x + w] = nstuff + i
        mode = i % 4
        if mode == 0:
            seg[y:y + h, x:x + w] = nstuff + cls_ind[i]          # semantic agrees
        elif mode == 1:
            seg[y:y + h, x:x + w] = rng.integers(0, nstuff)      # one stuff class covers it -> becomes stuff
        elif mode == 2:
            seg[y:y + h, x:x + w] = rng.integers(0, nstuff + nthing, size=(h, w))   # no majority
        else:
            seg[y:y + h, x:x + w] = nstuff + (cls_ind[i] + 1) % nthing    # another thing class
    if void:
        pan[H - 6:, W - 9:] = 255
    obj = None
    if with_obj:
        obj = rng.permutation(60)[:k].astype(np.int64) + (200 if big_ids else 0)
        if dup and k >= 4:
            obj[1] = obj[0]; obj[3] = obj[0]; obj[k - 1] = obj[2]
    return seg, pan, cls_ind.astype(np.int64), obj


def main(dataset='cityscapes_vps'):
    warnings.simplefilter('ignore')
    if dataset == 'viper':
        # tools/dataset/viper.py:93-130 (Viper.__init__ re-configures the dataset: 23 segmentation classes, 11 of them things ->
        # id_last_stuff = 12) and :661-727 (get_unified_pan_result)
        fn = load_reference_function('tools.dataset.viper', 23, 11)
        nstuff, nthing, fname = 13, 10, 'unify_cases_viper.npz'
    else:
        fn = load_reference_function()
        nstuff, nthing, fname = 11, 8, 'unify_cases.npz'
    rng = np.random.default_rng(0 if dataset != 'viper' else 7)
    out = {}
    clips = [
        dict(H=64, W=128, ks=[6, 9, 12], with_obj=True, dup=True, void=True, limit=4 * 64 * 64),
        dict(H=64, W=128, ks=[5, 0, 7], with_obj=False, dup=False, void=False, limit=200),
        dict(H=96, W=160, ks=[20, 33], with_obj=True, dup=True, void=False, limit=1500),
        dict(H=48, W=64, ks=[8], with_obj=True, dup=False, void=True, limit=100, big_ids=True),
        dict(H=256, W=512, ks=[40, 45], with_obj=True, dup=True, void=True, limit=4 * 64 * 64),
    ]
    for ci, c in enumerate(clips):
        segs, pans, clss, objs, names = [], [], [], [], []
        for fi, k in enumerate(c['ks']):
            seg, pan, cls_ind, obj = make_case(rng, c['H'], c['W'], k, c['with_obj'], c['dup'], c['void'], c.get('big_ids', False), nstuff, nthing)
            segs.append(seg); pans.append(pan); clss.append(cls_ind); objs.append(obj); names.append('f%d' % fi)
        res = fn(None, [s.copy() for s in segs], [p.copy() for p in pans], [c_.copy() for c_ in clss],
                 [o.copy() for o in objs] if c['with_obj'] else None, c['limit'], names)
        out['clip%d_limit' % ci] = np.int64(c['limit'])
        out['clip%d_with_obj' % ci] = np.int64(1 if c['with_obj'] else 0)
        out['clip%d_n' % ci] = np.int64(len(names))
        for fi, n in enumerate(names):
            out['clip%d_f%d_seg' % (ci, fi)] = segs[fi]; out['clip%d_f%d_pan' % (ci, fi)] = pans[fi]
            out['clip%d_f%d_cls' % (ci, fi)] = clss[fi]
            if c['with_obj']:
                out['clip%d_f%d_obj' % (ci, fi)] = objs[fi]
            out['clip%d_f%d_out' % (ci, fi)] = res[n]
    out['nclips'] = np.int64(len(clips))
    out['id_last_stuff'] = np.int64(nstuff - 1)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), fname)
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'cityscapes_vps')
