"""How ill-conditioned is the synthetic-weight network behind the FPN, and which weight RESCALING makes it well-conditioned?
(VERDICT r5 next #4b). For a set of scale factors on the order-defining layers the fp32 oracle is compared with its own float64
evaluation on the same fp32 inputs (FPN levels, flow): the distance is the noise floor of fp32 arithmetic on that network - what NO
implementation of the reference arithmetic can beat. CPU only.

    python tools/condition_search.py [--height 256 --width 512]

Knobs (all exact scalings of the seeded synthetic tensors, vps_amd.synth.conditioned_overrides):
  lite   extra_neck.liteflownet.flow_estimator.convs.3   (the fine flow: pixels of displacement the second warp samples at)
  tatt   extra_neck.tcea_fusion.tAtt_1 / tAtt_2          (the embeddings whose 256-term dot product feeds a sigmoid)
  satt   extra_neck.tcea_fusion.sAtt_4 / sAtt_add_2      (spatial attention logits / additive term)
  off    panopticFPN deform_convs.*.conv_offset          (sampling offsets of the deformable towers)
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vps_amd
from oracle.fusetrack import FuseTrackOracle, bfp_tcea, upsnet_fpn
from vps_amd import synth


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / max(float(b.abs().max()), 1e-30))


def floors(sd, lv, ref, flow):
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        o32, a32 = bfp_tcea(sd, 'extra_neck.', lv, ref, flow, return_aux=True)
        o64, a64 = bfp_tcea(sd64, 'extra_neck.', [t.double() for t in lv], [t.double() for t in ref], flow.double(), return_aux=True)
        f32 = upsnet_fpn(sd, 'panopticFPN.', o32[:4])[1]
        f64 = upsnet_fpn(sd64, 'panopticFPN.', [t.double() for t in o32[:4]])[1]         # the head alone, on the fp32 neck outputs
        f64c = upsnet_fpn(sd64, 'panopticFPN.', o64[:4])[1]                               # neck + head
    out = {k: rel(a32[k], a64[k]) for k in ('flow_fine', 'warp', 'fused', 'refined')}
    out.update(neck_p2=rel(o32[0], o64[0]), neck_p6=rel(o32[4], o64[4]), fcn_head_alone=rel(f32, f64), fcn_score=rel(f32, f64c),
               flow_fine_max=float(a32['flow_fine'].abs().max()), fused_max=float(a32['fused'].abs().max()))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--height', type=int, default=256)
    ap.add_argument('--width', type=int, default=512)
    ap.add_argument('--out', default='gpurun_out/condition_search.json')
    a = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = vps_amd.Config.fromfile(os.path.join(root, 'configs', 'cityscapes', 'fusetrack.py'))
    m = vps_amd.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    shapes = {k: v.shape for k, v in m.state_dict().items()}
    fr = synth.synth_clip(a.height, a.width, 2, 0)
    base = FuseTrackOracle(synth.synth_state_dict(shapes, 0))
    torch.set_num_threads(max(1, (os.cpu_count() or 2)))
    with torch.no_grad():
        lv = base.extract_feat(fr[1]); ref = base.extract_feat(fr[0])
        flow = base.compute_flow(fr[1], fr[0], 0.25)
    res = {}
    for name, kn in (('seeded', {}), ('lite0.05', dict(lite=0.05)), ('lite0.05+tatt0.25', dict(lite=0.05, tatt=0.25)),
                     ('lite0.05+tatt0.25+satt0.25', dict(lite=0.05, tatt=0.25, satt=0.25)),
                     ('lite0.05+tatt0.25+satt0.25+off0.1', dict(lite=0.05, tatt=0.25, satt=0.25, off=0.1)),
                     ('lite0.02+tatt0.1+satt0.1+off0.05', dict(lite=0.02, tatt=0.1, satt=0.1, off=0.05))):
        sd = synth.synth_state_dict(shapes, 0, overrides=synth.conditioned_overrides(shapes, 0, **kn) if kn else None)
        res[name] = floors(sd, lv, ref, flow)
        print(name, {k: '%.2e' % v for k, v in res[name].items()}, flush=True)
    os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
    json.dump(dict(size=[a.height, a.width], fp32_vs_fp64_oracle=res), open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
