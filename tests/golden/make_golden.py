"""Generate the golden vectors of tests/golden/*.npz by running the REAL reference detector code
(/root/reference/mmdet/models/detectors/panoptic_fusetrack.py and everything it builds) on the CPU of the build
container, with the import shims of ref_shims.py (third-party stubs + oracle-backed stand-ins for the CUDA-only ops).

    python tests/golden/make_golden.py [fusetrack|fuse|track]   # needs /root/reference; writes tests/golden/<variant>_clip.npz
    python tests/golden/make_golden.py fullsize                  # 2 frames at 1024x2048 -> tests/golden/fusetrack_fullsize.npz
    python tests/golden/make_golden.py fullsize_cond             # the same on the well-conditioned synthetic checkpoint (every stage held to 1e-4)
    python tests/golden/make_golden.py fullsize_sep              # 4 frames at 1024x2048, fitted well-separated box classifier (strict fixture)
    python tests/golden/make_golden.py fullsize_dense            # 6 frames at 1024x2048, fitted dense box classifier (32..53 detections per frame, strict)
    python tests/golden/make_golden.py r101                      # ResNet-101 variant (BASELINE config 5), 2 frames at 128x256
    python tests/golden/make_golden.py config5_sep               # ResNet-101 at 1088x1920, 3 frames, fitted box classifier (strict fixture of config 5)
    python tests/golden/make_golden.py seed1                     # FuseTrack, weight / clip seed 1, 3 frames at 128x192

The reference cannot travel to the GPU box; the vectors do. tests/test_oracle_golden.py checks the oracle against them
(CPU), tests/test_fusetrack_gpu.py checks the HIP path against the oracle and against these vectors (GPU).
Weights: vps_amd.synth.synth_state_dict (per-key seeded, no checkpoint exists offline). Inputs: vps_amd.synth.synth_clip.
"""
import os
import sys
import tempfile
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
warnings.simplefilter('ignore')

H, W, NFRAMES, SEED = 128, 256, 3, 0
FULL_H, FULL_W, FULL_NFRAMES = 1024, 2048, 2          # `fullsize`: the BASELINE frame size (configs[1]), FuseTrack only


def main(variant='fusetrack', H=H, W=W, NFRAMES=NFRAMES, out_name=None, full=False, depth=None, seed=SEED, separated=False, head='separated_fc_cls.npz', map_stride=1, conditioned=False):
    """full=True: the 1024x2048 golden — same quantities, the dense stage tensors strided so the file stays a few MB"""
    import ref_shims
    mods = ref_shims.install()
    from vps_amd import synth
    from vps_amd.registry import Config, ConfigDict
    import vps_amd

    cfg = Config.fromfile('/root/reference/configs/cityscapes/%s.py' % variant)
    if depth is not None:
        cfg.model['backbone']['depth'] = depth          # BASELINE config 5: the ResNet-101 variant (resnet.py:361)
    has_flow, has_track = variant != 'track', variant != 'fuse'
    # --- shapes of every parameter, from OUR containers; the reference model must expose exactly the same keys ---
    ours = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    shapes = {k: tuple(v.shape) for k, v in ours.state_dict().items()}
    # separated=True: the box classification layer fitted by search_separated.py (every listing decision has a margin)
    over = synth.separated_overrides(os.path.join(HERE, head)) if separated else {}
    if conditioned:
        # the WELL-CONDITIONED synthetic checkpoint (vps_amd.synth.conditioned_overrides, tools/condition_search.py): fine flow, deformable
        # offsets and attention logits scaled down so that the fp32 reference arithmetic itself is within ~1e-5 of its float64 evaluation
        over.update(synth.conditioned_overrides(shapes, seed))
    sd = synth.synth_state_dict(shapes, seed, overrides=over or None)

    # --- the reference detector; its __init__ loads FlowNet2 from cwd/work_dirs/flownet/FlowNet2_checkpoint.pth.tar ---
    tmp = tempfile.mkdtemp(prefix='vps_golden_')
    os.makedirs(os.path.join(tmp, 'work_dirs', 'flownet'))
    torch.save({'state_dict': {k[len('flownet2.'):]: v for k, v in sd.items() if k.startswith('flownet2.')}},
               os.path.join(tmp, 'work_dirs', 'flownet', 'FlowNet2_checkpoint.pth.tar'))
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        builder = sys.modules['mmdet.models.builder']
        model_cfg = ConfigDict.wrap(dict(cfg.model))
        model_cfg['pretrained'] = None
        ref = builder.build_detector(model_cfg, train_cfg=None, test_cfg=cfg.test_cfg)
    finally:
        os.chdir(cwd)
    ref.eval()
    ref_shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    missing = sorted(set(ref_shapes) - set(shapes)); extra = sorted(set(shapes) - set(ref_shapes))
    assert not missing and not extra, ('state_dict key mismatch', missing[:10], extra[:10])
    bad = [k for k in shapes if shapes[k] != ref_shapes[k]]
    assert not bad, ('shape mismatch', bad[:10])
    print('state_dict: %d keys identical to the reference module tree' % len(shapes))
    ref.load_state_dict(sd)
    # the key/shape manifest of the REFERENCE module tree travels in the .npz: tests/test_oracle_golden.py compares it
    # with vps_amd's state_dict on every run (the drop-in checkpoint contract)
    import json
    manifest = json.dumps({k: list(v) for k, v in sorted(ref_shapes.items())})

    # --- hooks on the stage boundaries ---
    cap = {}

    def hook(name):
        def f(mod, inp, out):
            cap.setdefault(name, []).append(out)
        return f
    if has_flow:
        ref.flownet2.register_forward_hook(hook('flownet2'))
        ref.extra_neck.register_forward_hook(hook('extra_neck'))
    ref.panopticFPN.register_forward_hook(hook('panopticFPN'))
    ref.bbox_head.register_forward_hook(hook('bbox_head'))
    ref.mask_head.register_forward_hook(hook('mask_head'))
    ref.neck.register_forward_hook(hook('neck'))
    orig_rpn = ref.simple_test_rpn

    def rpn_wrap(*a, **k):
        r = orig_rpn(*a, **k)
        cap.setdefault('proposals', []).append(r[0])
        return r
    ref.simple_test_rpn = rpn_wrap

    frames = synth.synth_clip(H, W, NFRAMES, seed)
    out = {}
    with torch.no_grad():
        for t in range(NFRAMES):
            img = frames[t]
            ref_img = frames[t - 1] if t > 0 else frames[0]
            meta = synth.img_meta(H, W, 10000 + t + 1)
            cap.clear()
            bbox_res, mask_res, pano = ref(return_loss=False, rescale=True, img=[img], img_meta=[[meta]], ref_img=[ref_img])
            p = 'f%d.' % t
            # map_stride > 1 (the 6-frame dense fixture): every map_stride-th pixel of the two maps, so the file stays ~10 MB
            out[p + 'fcn_outputs'] = pano['fcn_outputs'].numpy().astype(np.uint8)[..., ::map_stride, ::map_stride]
            out[p + 'panoptic_outputs'] = pano['panoptic_outputs'].numpy().astype(np.uint8)[..., ::map_stride, ::map_stride]
            out[p + 'panoptic_cls_inds'] = pano['panoptic_cls_inds'].numpy()
            out[p + 'panoptic_cls_prob'] = pano['panoptic_cls_prob'].numpy()
            if has_track:
                out[p + 'panoptic_det_labels'] = pano['panoptic_det_labels'].numpy()
                out[p + 'panoptic_det_obj_ids'] = np.asarray(pano['panoptic_det_obj_ids'].numpy())
                out[p + 'bbox_ids'] = np.array(sorted(int(k) for k in bbox_res.keys()), dtype=np.int64)
            else:
                assert sorted(pano.keys()) == ['fcn_outputs', 'panoptic_cls_inds', 'panoptic_cls_prob', 'panoptic_outputs']
                out[p + 'bbox_counts'] = np.array([len(b) for b in bbox_res], dtype=np.int64)
            s1, s2 = (8, 4) if full else (2, 1)                              # spatial strides of the stored stage tensors
            c5 = 32 if full else None                                        # channels kept of the low-resolution levels
            if has_flow:
                out[p + 'flow_full'] = cap['flownet2'][0][0][:, ::s1, ::s1].numpy()
            # neck is called twice (img, ref_img): first call = target frame
            out[p + 'fpn_p2'] = cap['neck'][0][0][0, :8, ::s2, ::s2].numpy()          # first 8 channels of P2
            out[p + 'fpn_p5'] = cap['neck'][0][3][0, :c5].numpy()
            if has_flow:
                out[p + 'neck_out_p2'] = cap['extra_neck'][0][0][0, :8, ::s2, ::s2].numpy()
                out[p + 'neck_out_p6'] = cap['extra_neck'][0][4][0, :c5].numpy()
            out[p + 'fcn_score'] = cap['panopticFPN'][0][1][0, :, ::s2, ::s2].numpy()
            out[p + 'proposals'] = cap['proposals'][0].numpy()
            out[p + 'cls_score'] = cap['bbox_head'][0][0].numpy()
            out[p + 'bbox_pred'] = cap['bbox_head'][0][1].numpy()
            out[p + 'mask_pred'] = cap['mask_head'][0][:8].numpy()          # first 8 detections
            print('%s frame %d: K=%d kept=%d ids=%s' % (variant, t, cap['mask_head'][0].shape[0], len(out[p + 'panoptic_cls_inds']),
                                                       out[p + 'panoptic_det_obj_ids'][:8] if has_track else None))
    out['meta'] = np.array([H, W, NFRAMES, seed], dtype=np.int64)
    out['strides'] = np.array([s1, s2, c5 or 0], dtype=np.int64)
    out['map_stride'] = np.int64(map_stride)
    out['state_dict_manifest'] = np.frombuffer(manifest.encode(), dtype=np.uint8)
    path = os.path.join(HERE, out_name or '%s_clip.npz' % variant)
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) / 1e6, 'MB')


if __name__ == '__main__':
    v = sys.argv[1] if len(sys.argv) > 1 else 'fusetrack'
    if v == 'fullsize':
        main('fusetrack', FULL_H, FULL_W, FULL_NFRAMES, 'fusetrack_fullsize.npz', full=True)
    elif v == 'fullsize_sep':
        # the STRICT full-size fixture: 4 frames at 1024x2048, well-separated detections (tests/golden/separated_fc_cls.npz)
        main('fusetrack', FULL_H, FULL_W, 4, 'fusetrack_fullsize_sep.npz', full=True, separated=True)
    elif v == 'fullsize_dense':
        # the DENSE strict fixture: 6 frames at 1024x2048, 32..53 well-separated detections per frame (tests/golden/search_dense.py ->
        # dense_fc_cls.npz, chosen by oracle margins only)
        main('fusetrack', FULL_H, FULL_W, 6, 'fusetrack_fullsize_dense.npz', full=True, separated=True, head='dense_fc_cls.npz', map_stride=2)
    elif v == 'fullsize_cond':
        # 2 frames at 1024x2048 on the well-conditioned synthetic checkpoint: the fixture on which EVERY stage is held to 1e-4
        main('fusetrack', FULL_H, FULL_W, FULL_NFRAMES, 'fusetrack_fullsize_cond.npz', full=True, conditioned=True)
    elif v == 'r101':
        main('fusetrack', H, W, 2, 'fusetrack_r101_clip.npz', depth=101)
    elif v == 'config5_sep':
        # BASELINE config 5, strict: the ResNet-101 model on 3 frames at 1088x1920 with the fitted classification layer of
        # VPS_SEP_CONFIG5=1 search_separated.py (config5_fc_cls.npz, chosen by oracle margins only)
        main('fusetrack', 1088, 1920, 3, 'fusetrack_config5_sep.npz', full=True, separated=True, head='config5_fc_cls.npz', depth=101, map_stride=2)
    elif v == 'seed1':
        main('fusetrack', 128, 192, 3, 'fusetrack_clip_seed1.npz', seed=1)      # second weight / clip seed, another aspect ratio
    else:
        main(v)
