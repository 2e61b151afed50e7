"""Deterministic synthetic weights and inputs (there is no network for checkpoints or datasets).

`synth_state_dict(shapes, seed)` fills every tensor of a state_dict from a per-key seeded generator, with scales that keep
activations O(1) through the 50-100 layer stacks (variance-preserving conv init, mild BatchNorm statistics) and make the
heads produce a realistic number of detections (logit std ~3 on fc_cls). The same function is used for the HIP path, the
CPU oracle and the golden-vector script, so all three see bit-identical weights.

Input recipes follow SURVEY §8d: low-pass filtered random frames, the second frame a translated + noisy copy.
"""
import math
import re
import zlib

import numpy as np
import torch


def _gen(key, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_tensor(key, shape, seed=0):
    g = _gen(key, seed)
    shape = tuple(shape)
    leaf = key.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=torch.long)
    if leaf == 'running_mean':
        return torch.randn(shape, generator=g) * 0.1
    if leaf == 'running_var':
        return torch.rand(shape, generator=g) * 1.0 + 0.5
    is_norm = len(shape) == 1 and ('.bn' in key or 'downsample.1' in key or
                                   any(('deform_convs.0.%d.' % i) in key for i in (1, 4, 7)))
    if is_norm:
        if leaf == 'weight':
            w = torch.rand(shape, generator=g) * 1.0 + 0.5
            if '.bn3.' in key:
                w = w * 0.35          # keep the residual branches from blowing up over 16 blocks
                m = re.search(r'layer3\.(\d+)\.bn3', key)
                if m and int(m.group(1)) >= 6:
                    w = w * 0.4       # ResNet-101: 17 more blocks in layer3 (ResNet-50 has no such keys: its tensors are unchanged)
            return w
        return torch.randn(shape, generator=g) * 0.1
    if leaf == 'bias':
        b = torch.randn(shape, generator=g) * 0.05
        if key.endswith('bbox_head.fc_reg.bias'):
            b[2::4] -= 4.0; b[3::4] -= 4.0        # dw, dh ~ -0.8 after the /5 weights: detections shrink like refined boxes
        return b
    # weights
    if len(shape) == 4:
        if 'deconv' in key or 'upsampled_flow' in key or key.endswith('mask_head.upsample.weight'):
            fan_in = shape[0] * shape[2] * shape[3] / 4.0      # ConvTranspose2d [I,O,k,k], stride 2
        else:
            fan_in = shape[1] * shape[2] * shape[3]
        std = math.sqrt(2.0 / fan_in)
        if 'flownetfusion.predict_flow' in key:
            std *= 1.0                                        # final full-resolution flow of a few pixels
        elif 'predict_flow' in key:
            std *= 0.3                                        # 1/4-res flows of a few tenths -> x20 = pixels
        elif 'upsampled_flow' in key:
            std = math.sqrt(1.0 / fan_in)
        elif 'conv_offset' in key:
            std *= 0.5                                        # DCN offsets ~ N(0, <1 px)
        elif 'rpn_reg' in key:
            std *= 0.03                                       # anchor deltas ~ N(0, 0.4)
        elif 'rpn_cls' in key:
            std *= 0.15                                       # objectness logits ~ N(0, 1.5): no saturated (tied) scores
        elif 'flow_estimator.convs.3' in key:
            std *= 0.3
        elif 'conv_logits' in key or 'conv_pred' in key:
            std *= 1.5
        return torch.randn(shape, generator=g) * std
    if len(shape) == 2:
        std = math.sqrt(2.0 / shape[1])
        if 'fc_cls' in key:
            std *= 0.25                                       # logit std ~3 on the O(10) synthetic FPN features
            w = torch.randn(shape, generator=g) * std
            return w - w.mean(dim=0, keepdim=True)           # no class is favoured by the (mostly positive) features
        elif 'fc_reg' in key:
            std *= 0.05
        elif 'track_head.fcs.1' in key:
            std *= 0.5
        return torch.randn(shape, generator=g) * std
    return torch.randn(shape, generator=g) * 0.1


def synth_state_dict(shapes, seed=0, prefix='', overrides=None):
    """shapes: dict key -> shape (e.g. {k: v.shape for k, v in model.state_dict().items()}). `overrides`: key -> tensor replacing
    the seeded one (tests/golden/separated_fc_cls.npz: a classification layer fitted so that every listing decision of the
    full-size golden clip has a margin — tests/golden/search_separated.py)"""
    sd = {k: synth_tensor(prefix + k, tuple(s), seed) for k, s in shapes.items()}
    for k, v in (overrides or {}).items():
        assert prefix + k in sd or k in sd, k
        kk = k if k in sd else prefix + k
        assert tuple(v.shape) == tuple(sd[kk].shape), (k, tuple(v.shape), tuple(sd[kk].shape))
        sd[kk] = torch.as_tensor(v).float().clone()
    return sd


def separated_overrides(path):
    """-> overrides dict of the fitted, well-separated box classification layer stored at `path` (.npz: weight, bias) plus any other
    state_dict tensor the file carries under its key (config5_fc_cls.npz: the rescaled `rpn_head.rpn_cls.*`)"""
    z = np.load(path)
    out = {'bbox_head.fc_cls.weight': torch.from_numpy(z['weight']), 'bbox_head.fc_cls.bias': torch.from_numpy(z['bias'])}
    for k in z.files:
        if k.endswith('.weight') or k.endswith('.bias'):
            out[k] = torch.from_numpy(z[k])
    return out


def conditioned_overrides(shapes, seed=0, lite=0.02, tatt=0.1, satt=0.1, off=0.05):
    """overrides that make the synthetic network behind the FPN WELL-CONDITIONED (round 6; tools/condition_search.py measures the effect):
    exact scalings of the seeded tensors of the layers whose outputs are used as sampling positions or as sigmoid logits. With the plain
    seeded weights the fine flow is tens of pixels, the deformable offsets several pixels, the temporal-attention logits dozens: the
    rounding noise of the fp32 REFERENCE arithmetic itself is then amplified to 2-4e-4 at the neck (fp32 oracle against its float64
    evaluation), and no implementation can be held to less. A trained checkpoint has small residual flows and offsets (both layers
    start from zero): the scaled network is the more faithful stand-in, not the less."""
    out = {}
    def scale(prefix, f):
        for k in shapes:
            if k.startswith(prefix) and (k.endswith('.weight') or k.endswith('.bias')):
                out[k] = synth_tensor(k, tuple(shapes[k]), seed) * f
    scale('extra_neck.liteflownet.flow_estimator.convs.3', lite)
    scale('extra_neck.tcea_fusion.tAtt_1', tatt); scale('extra_neck.tcea_fusion.tAtt_2', tatt)
    scale('extra_neck.tcea_fusion.sAtt_4', satt); scale('extra_neck.tcea_fusion.sAtt_add_2', satt)
    for k in shapes:
        if k.startswith('panopticFPN.') and '.conv_offset.' in k:
            out[k] = synth_tensor(k, tuple(shapes[k]), seed) * off
    return out


def load_synth(model, seed=0, overrides=None):
    sd = synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, seed, overrides=overrides)
    model.load_state_dict(sd)
    return sd


# ------------------------------------------------------------------------------------------------ inputs
MEAN = np.array([123.675, 116.28, 103.53], dtype=np.float32)
STD = np.array([58.395, 57.12, 57.375], dtype=np.float32)


def _box_blur(a, k):
    """separable box filter (edge-replicated) on an HxWxC float array"""
    pad = k // 2
    for axis in (0, 1):
        ap = np.pad(a, [(pad, pad) if ax == axis else (0, 0) for ax in range(a.ndim)], mode='edge')
        c = np.cumsum(ap, axis=axis, dtype=np.float64)
        z = np.zeros_like(np.take(c, [0], axis=axis))
        c = np.concatenate([z, c], axis=axis)
        n = a.shape[axis]
        a = (np.take(c, np.arange(k, k + n), axis=axis) - np.take(c, np.arange(0, n), axis=axis)) / k
    return a.astype(np.float32)


def synth_frame(H, W, seed=0, shift=(0, 0), noise=0.0):
    """uint8-valued BGR frame [H,W,3] (float32): low-pass filtered noise, optionally translated by (dx,dy) + N(0,noise)"""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(H + 64, W + 64, 3)).astype(np.float32)
    base = _box_blur(base, 9)
    base = (base - base.min()) / max(base.max() - base.min(), 1e-6) * 255.0
    dx, dy = shift
    img = base[32 + dy:32 + dy + H, 32 + dx:32 + dx + W]
    if noise > 0:
        img = img + np.random.default_rng(seed * 7919 + 131 * dx + dy + 17).normal(0, noise, img.shape).astype(np.float32)
    return np.clip(np.round(img), 0, 255).astype(np.float32)


def normalize_frame(bgr):
    """pipeline Normalize(to_rgb=True) + ImageToTensor: -> torch [1,3,H,W] fp32 (fusetrack.py:153-154,176-191)"""
    rgb = bgr[..., ::-1]
    x = (rgb - MEAN) / STD
    return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))[None].float()


def synth_pair(H, W, seed=0):
    """(img, ref_img) normalised tensors: ref = base frame, img = ref translated by (+3,-2) px + N(0,2) noise"""
    ref = synth_frame(H, W, seed=seed)
    img = synth_frame(H, W, seed=seed, shift=(3, -2), noise=2.0)
    return normalize_frame(img), normalize_frame(ref)


def synth_clip(H, W, nframes, seed=0):
    """frame_t = base translated by t*(2,1) px + noise(seed t)"""
    return [normalize_frame(synth_frame(H, W, seed=seed, shift=(2 * t, 1 * t), noise=2.0 if t else 0.0)) for t in range(nframes)]


def img_meta(H, W, iid, filename='synthetic_city_000000.png'):
    return dict(filename=filename, iid=iid, img_shape=(H, W, 3), ori_shape=(H, W, 3), pad_shape=(H, W, 3), scale_factor=1.0,
                flip=False)
