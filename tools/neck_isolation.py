"""Where does the 100x jump of the stage error between the FPN output (2e-6 of max|ref|) and the fusion-neck output (2.5e-4) come from
? The HIP BFP-TCEA neck is run on four combinations of inputs - its own FPN levels / flow, or the ORACLE's - and
every intermediate tensor is compared with the oracle's neck on the oracle's inputs:

    hip levels + hip flow      the product path (the error the stage tests see)
    hip levels + oracle flow   the flow error removed: if the neck error falls to the level error the jump is flow -> warp amplification
    oracle levels + hip flow   the level error removed
    oracle levels + oracle flow   the neck kernels alone (their own summation-order error)

and the ORACLE's own neck is evaluated in float64 on the same inputs: the distance of the fp32 oracle from it is the noise floor of fp32
arithmetic on this (synthetic-weight) network - an error of that size in the HIP neck is conditioning, not a semantic difference.

The HIP cases need the GPU box; the oracle side runs on the host cores (about a minute at 1024x2048; `--no-hip`: the fp64 leg alone,
anywhere). Writes gpurun_out/neck_isolation.json.
    python tools/neck_isolation.py [--height 1024 --width 2048 --prec f16x3] [--no-hip]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vps_amd
from oracle.fusetrack import FuseTrackOracle, bfp_tcea
from vps_amd import nhwc, synth


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / max(float(b.abs().max()), 1e-12))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--height', type=int, default=1024)
    ap.add_argument('--width', type=int, default=2048)
    ap.add_argument('--prec', default='f16x3')
    ap.add_argument('--out', default='gpurun_out/neck_isolation.json')
    ap.add_argument('--no-hip', action='store_true')
    a = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = torch.device('cuda:0') if not a.no_hip else None
    nhwc.DEFAULT_PREC = nhwc.PREC_NAMES[a.prec]
    cfg = vps_amd.Config.fromfile(os.path.join(root, 'configs', 'cityscapes', 'fusetrack.py'))
    m = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    sd = synth.load_synth(m, 0)
    if dev is not None:
        m.ensure_packed(dev)
    fr = synth.synth_clip(a.height, a.width, 2, 0)
    o = FuseTrackOracle(sd)
    torch.set_num_threads(max(1, min(64, (os.cpu_count() or 2) // 2)))
    with torch.no_grad():
        lv_o = o.extract_feat(fr[1]); ref_o = o.extract_feat(fr[0])
        flow_o = o.compute_flow(fr[1], fr[0], 0.25)
        outs_o, aux_o = bfp_tcea(o.sd, 'extra_neck.', lv_o, ref_o, flow_o, return_aux=True)
        # the oracle's neck in float64 on the SAME fp32 inputs: how far the fp32 reference arithmetic itself is from the exact map
        sd64 = {k: v.double() for k, v in o.sd.items()}
        outs64, aux64 = bfp_tcea(sd64, 'extra_neck.', [t.double() for t in lv_o], [t.double() for t in ref_o], flow_o.double(), return_aux=True)
    floor = {k: rel(aux_o[k], aux64[k]) for k in ('flow_fine', 'warp', 'fused', 'refined')}
    floor.update(neck_p2=rel(outs_o[0], outs64[0]), neck_p6=rel(outs_o[4], outs64[4]))
    print('fp32 oracle vs fp64 oracle (noise floor of the reference arithmetic):', floor, flush=True)
    if a.no_hip:
        os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
        json.dump(dict(size=[a.height, a.width], fp32_oracle_vs_fp64_oracle=floor), open(a.out, 'w'), indent=1)
        return
    lv_h = [t.cpu() for t in m.extract_feat(fr[1].to(dev))]; ref_h = [t.cpu() for t in m.extract_feat(fr[0].to(dev))]
    flow_h = m.compute_flow(fr[1].to(dev), fr[0].to(dev), 0.25)[0].cpu()
    en = m.extra_neck
    C = en.in_channels
    res = dict(size=[a.height, a.width], prec=a.prec, fp32_oracle_vs_fp64_oracle=floor,
               inputs=dict(levels_p2=rel(lv_h[0], lv_o[0]), levels_p5=rel(lv_h[3], lv_o[3]), ref_levels_p2=rel(ref_h[0], ref_o[0]),
                           flow_quarter=rel(flow_h, flow_o), flow_quarter_abs_max_err=float((flow_h - flow_o).abs().max()),
                           flow_quarter_abs_max=float(flow_o.abs().max())), cases={})
    for name, lv, rlv, fl in (('hip_levels+hip_flow', lv_h, ref_h, flow_h), ('hip_levels+oracle_flow', lv_h, ref_h, flow_o),
                              ('oracle_levels+hip_flow', lv_o, ref_o, flow_h), ('oracle_levels+oracle_flow', lv_o, ref_o, flow_o)):
        ws = nhwc.Workspace(dev)
        ws.pooling = False                # the intermediates (warp, fused, refined) are read back: plain buffers
        L = [nhwc.from_nchw(t.to(dev), ws, 'l%d' % i) for i, t in enumerate(lv)]
        R = [nhwc.from_nchw(t.to(dev), ws, 'r%d' % i) for i, t in enumerate(rlv)]
        cat = en.gather(L, ws, 'cat'); refcat = en.gather(R, ws, 'refcat')
        nhwc.resize(nhwc.from_nchw(fl.to(dev), ws, 'fi'), cat.window(C + 81, 2), 'nearest')        # same size: a copy into the window
        outs, aux = en.run(L, cat, refcat.window(0, C), ws, 'n.')
        torch.cuda.synchronize()
        res['cases'][name] = dict(
            bsf=rel(cat.window(0, C).to_nchw().cpu(), aux_o['bsf']), flow_fine=rel(aux['flow_fine'].to_nchw().cpu(), aux_o['flow_fine']),
            flow_fine_abs_max_err=float((aux['flow_fine'].to_nchw().cpu() - aux_o['flow_fine']).abs().max()),
            warp2=rel(aux['warp'].to_nchw().cpu(), aux_o['warp']), fused=rel(aux['fused'].to_nchw().cpu(), aux_o['fused']),
            refined=rel(aux['refined'].to_nchw().cpu(), aux_o['refined']), neck_p2=rel(outs[0].to_nchw().cpu(), outs_o[0]),
            neck_p6=rel(outs[4].to_nchw().cpu(), outs_o[4]))
        print(name, res['cases'][name], flush=True)
    os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
    json.dump(res, open(a.out, 'w'), indent=1)
    print(json.dumps(res['inputs']))


if __name__ == '__main__':
    main()
