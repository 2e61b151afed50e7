"""Oracle FlowNet2 forward (TEST INFRASTRUCTURE ONLY). Functional restatement over a state_dict with the
reference's key names (`<prefix>flownetc.conv1.0.weight`, ...). Paths relative to
/root/reference/mmdet/models/flow_modules.
"""
import torch
import torch.nn.functional as F

from . import ops


def _conv(sd, p, x, k=3, stride=1):
    # submodules.py:6-17 conv(batchNorm=False): Conv2d(pad=(k-1)//2, bias) + LeakyReLU(0.1)
    y = F.conv2d(x, sd[p + '.0.weight'], sd[p + '.0.bias'], stride=stride, padding=(k - 1) // 2)
    return F.leaky_relu(y, 0.1)


def _iconv(sd, p, x):
    # submodules.py:19-29 i_conv: conv 3x3 + bias, NO activation
    return F.conv2d(x, sd[p + '.0.weight'], sd[p + '.0.bias'], stride=1, padding=1)


def _pred(sd, p, x):
    # submodules.py:31-32 predict_flow
    return F.conv2d(x, sd[p + '.weight'], sd[p + '.bias'], stride=1, padding=1)


def _deconv(sd, p, x):
    # submodules.py:34-38 deconv: ConvTranspose2d(4, 2, 1, bias) + LeakyReLU(0.1)
    return F.leaky_relu(F.conv_transpose2d(x, sd[p + '.0.weight'], sd[p + '.0.bias'], stride=2, padding=1), 0.1)


def _upflow(sd, p, x):
    return F.conv_transpose2d(x, sd[p + '.weight'], sd.get(p + '.bias'), stride=2, padding=1)


def _decoder_s(sd, p, c2, c3, c4, c5, c6):
    """Shared refinement of FlowNetC/FlowNetS (FlowNetC.py:104-123, FlowNetS.py:69-90)."""
    flow6 = _pred(sd, p + 'predict_flow6', c6)
    flow6_up = _upflow(sd, p + 'upsampled_flow6_to_5', flow6)
    d5 = _deconv(sd, p + 'deconv5', c6)
    cat5 = torch.cat((c5, d5, flow6_up), 1)
    flow5 = _pred(sd, p + 'predict_flow5', cat5)
    flow5_up = _upflow(sd, p + 'upsampled_flow5_to_4', flow5)
    d4 = _deconv(sd, p + 'deconv4', cat5)
    cat4 = torch.cat((c4, d4, flow5_up), 1)
    flow4 = _pred(sd, p + 'predict_flow4', cat4)
    flow4_up = _upflow(sd, p + 'upsampled_flow4_to_3', flow4)
    d3 = _deconv(sd, p + 'deconv3', cat4)
    cat3 = torch.cat((c3, d3, flow4_up), 1)
    flow3 = _pred(sd, p + 'predict_flow3', cat3)
    flow3_up = _upflow(sd, p + 'upsampled_flow3_to_2', flow3)
    d2 = _deconv(sd, p + 'deconv2', cat3)
    cat2 = torch.cat((c2, d2, flow3_up), 1)
    return _pred(sd, p + 'predict_flow2', cat2)


def flownetc(sd, p, x):
    """FlowNetC.py:71-128 (eval: returns flow2 only)."""
    x1, x2 = x[:, 0:3], x[:, 3:]
    c1a = _conv(sd, p + 'conv1', x1, 7, 2)
    c2a = _conv(sd, p + 'conv2', c1a, 5, 2)
    c3a = _conv(sd, p + 'conv3', c2a, 5, 2)
    c1b = _conv(sd, p + 'conv1', x2, 7, 2)
    c2b = _conv(sd, p + 'conv2', c1b, 5, 2)
    c3b = _conv(sd, p + 'conv3', c2b, 5, 2)
    corr = ops.correlation(c3a, c3b, pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2)
    corr = F.leaky_relu(corr, 0.1)
    redir = _conv(sd, p + 'conv_redir', c3a, 1, 1)
    c31 = _conv(sd, p + 'conv3_1', torch.cat((redir, corr), 1))
    c4 = _conv(sd, p + 'conv4_1', _conv(sd, p + 'conv4', c31, 3, 2))
    c5 = _conv(sd, p + 'conv5_1', _conv(sd, p + 'conv5', c4, 3, 2))
    c6 = _conv(sd, p + 'conv6_1', _conv(sd, p + 'conv6', c5, 3, 2))
    return _decoder_s(sd, p, c2a, c31, c4, c5, c6)


def flownets(sd, p, x):
    """FlowNetS.py:59-94."""
    c1 = _conv(sd, p + 'conv1', x, 7, 2)
    c2 = _conv(sd, p + 'conv2', c1, 5, 2)
    c3 = _conv(sd, p + 'conv3_1', _conv(sd, p + 'conv3', c2, 5, 2))
    c4 = _conv(sd, p + 'conv4_1', _conv(sd, p + 'conv4', c3, 3, 2))
    c5 = _conv(sd, p + 'conv5_1', _conv(sd, p + 'conv5', c4, 3, 2))
    c6 = _conv(sd, p + 'conv6_1', _conv(sd, p + 'conv6', c5, 3, 2))
    return _decoder_s(sd, p, c2, c3, c4, c5, c6)


def flownetsd(sd, p, x):
    """FlowNetSD.py:66-105 (inter_conv* have no activation)."""
    c0 = _conv(sd, p + 'conv0', x)
    c1 = _conv(sd, p + 'conv1_1', _conv(sd, p + 'conv1', c0, 3, 2))
    c2 = _conv(sd, p + 'conv2_1', _conv(sd, p + 'conv2', c1, 3, 2))
    c3 = _conv(sd, p + 'conv3_1', _conv(sd, p + 'conv3', c2, 3, 2))
    c4 = _conv(sd, p + 'conv4_1', _conv(sd, p + 'conv4', c3, 3, 2))
    c5 = _conv(sd, p + 'conv5_1', _conv(sd, p + 'conv5', c4, 3, 2))
    c6 = _conv(sd, p + 'conv6_1', _conv(sd, p + 'conv6', c5, 3, 2))
    flow6 = _pred(sd, p + 'predict_flow6', c6)
    flow6_up = _upflow(sd, p + 'upsampled_flow6_to_5', flow6)
    d5 = _deconv(sd, p + 'deconv5', c6)
    cat5 = torch.cat((c5, d5, flow6_up), 1)
    flow5 = _pred(sd, p + 'predict_flow5', _iconv(sd, p + 'inter_conv5', cat5))
    flow5_up = _upflow(sd, p + 'upsampled_flow5_to_4', flow5)
    d4 = _deconv(sd, p + 'deconv4', cat5)
    cat4 = torch.cat((c4, d4, flow5_up), 1)
    flow4 = _pred(sd, p + 'predict_flow4', _iconv(sd, p + 'inter_conv4', cat4))
    flow4_up = _upflow(sd, p + 'upsampled_flow4_to_3', flow4)
    d3 = _deconv(sd, p + 'deconv3', cat4)
    cat3 = torch.cat((c3, d3, flow4_up), 1)
    flow3 = _pred(sd, p + 'predict_flow3', _iconv(sd, p + 'inter_conv3', cat3))
    flow3_up = _upflow(sd, p + 'upsampled_flow3_to_2', flow3)
    d2 = _deconv(sd, p + 'deconv2', cat3)
    cat2 = torch.cat((c2, d2, flow3_up), 1)
    return _pred(sd, p + 'predict_flow2', _iconv(sd, p + 'inter_conv2', cat2))


def flownetfusion(sd, p, x):
    """FlowNetFusion.py:47-66."""
    c0 = _conv(sd, p + 'conv0', x)
    c1 = _conv(sd, p + 'conv1_1', _conv(sd, p + 'conv1', c0, 3, 2))
    c2 = _conv(sd, p + 'conv2_1', _conv(sd, p + 'conv2', c1, 3, 2))
    flow2 = _pred(sd, p + 'predict_flow2', c2)
    flow2_up = _upflow(sd, p + 'upsampled_flow2_to_1', flow2)
    d1 = _deconv(sd, p + 'deconv1', c2)
    cat1 = torch.cat((c1, d1, flow2_up), 1)
    flow1 = _pred(sd, p + 'predict_flow1', _iconv(sd, p + 'inter_conv1', cat1))
    flow1_up = _upflow(sd, p + 'upsampled_flow1_to_0', flow1)
    d0 = _deconv(sd, p + 'deconv0', cat1)
    cat0 = torch.cat((c0, d0, flow1_up), 1)
    return _pred(sd, p + 'predict_flow0', _iconv(sd, p + 'inter_conv0', cat0))


def flow_input(img, ref_img, mean, std, rgb_max=255.0):
    """The 6-channel tensor FlowNetC / FlowNetSD see, from the detector's normalised frames:
    utils/flow_utils.py:5-10 `denormalize` (x * std + mean per channel, both frames), panoptic_fusetrack.py:119-128 (stack, zero pad
    bottom / right of the two special sizes - in 0..255 RGB space, i.e. BEFORE the mean is taken), flow_modules/flownet2.py:135-139
    (mean over both frames and all padded pixels, / rgb_max, frames concatenated along the channels). -> [B, 6, Hp, Wp]"""
    def denorm(t):
        t = t.clone()
        for c in range(3):
            t[:, c] = t[:, c] * std[c] + mean[c]
        return t
    rgbs = torch.stack([denorm(img), denorm(ref_img)], dim=2)
    H, W = rgbs.size(-2), rgbs.size(-1)
    if H == 800 and W == 1600:
        rgbs = F.pad(rgbs, (0, 64, 0, 32))
    elif H == 200 and W == 400:
        rgbs = F.pad(rgbs, (0, 48, 0, 56))
    rgb_mean = rgbs.contiguous().view(rgbs.shape[:2] + (-1,)).mean(dim=-1).view(rgbs.shape[:2] + (1, 1, 1))
    x = (rgbs - rgb_mean) / rgb_max
    return torch.cat((x[:, :, 0], x[:, :, 1]), dim=1)


def flownet2(sd, p, inputs, rgb_max=255.0, div_flow=20.0, return_stages=False):
    """flownet2.py:133-198. inputs [B,3,2,H,W] RGB 0..255."""
    b = inputs.shape[0]
    rgb_mean = inputs.contiguous().view(inputs.shape[:2] + (-1,)).mean(dim=-1).view(inputs.shape[:2] + (1, 1, 1))
    x = (inputs - rgb_mean) / rgb_max
    x = torch.cat((x[:, :, 0], x[:, :, 1]), dim=1)
    up_bil = lambda t: F.interpolate(t, scale_factor=4, mode='bilinear', align_corners=False)
    up_near = lambda t: F.interpolate(t, scale_factor=4, mode='nearest')

    c_flow2 = flownetc(sd, p + 'flownetc.', x)
    c_flow = up_bil(c_flow2 * div_flow)
    res1 = ops.resample2d(x[:, 3:], c_flow)
    norm1 = ops.channelnorm(x[:, :3] - res1)
    concat1 = torch.cat((x, res1, c_flow / div_flow, norm1), dim=1)

    s1_flow2 = flownets(sd, p + 'flownets_1.', concat1)
    s1_flow = up_bil(s1_flow2 * div_flow)
    res2 = ops.resample2d(x[:, 3:], s1_flow)
    norm2 = ops.channelnorm(x[:, :3] - res2)
    concat2 = torch.cat((x, res2, s1_flow / div_flow, norm2), dim=1)

    s2_flow2 = flownets(sd, p + 'flownets_2.', concat2)
    s2_flow = up_near(s2_flow2 * div_flow)
    norm_s2 = ops.channelnorm(s2_flow)
    diff_s2 = ops.channelnorm(x[:, :3] - ops.resample2d(x[:, 3:], s2_flow))

    sd_flow2 = flownetsd(sd, p + 'flownets_d.', x)
    sd_flow = up_near(sd_flow2 / div_flow)      # sic: divided (flownet2.py:180)
    norm_sd = ops.channelnorm(sd_flow)
    diff_sd = ops.channelnorm(x[:, :3] - ops.resample2d(x[:, 3:], sd_flow))

    concat3 = torch.cat((x[:, :3], sd_flow, s2_flow, norm_sd, norm_s2, diff_sd, diff_s2), dim=1)
    out = flownetfusion(sd, p + 'flownetfusion.', concat3)
    if return_stages:
        return out, dict(x=x, c_flow2=c_flow2, concat1=concat1, s1_flow2=s1_flow2, concat2=concat2,
                         s2_flow2=s2_flow2, sd_flow2=sd_flow2, concat3=concat3)
    return out
