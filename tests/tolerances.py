"""The stated fp32 tolerance of the path (north_star: "within a stated fp32 logit tolerance"), ONE number per stage - DESIGN.md §4 quotes
this file, the GPU parity tests import it (no per-fixture widening of a stage tolerance anywhere else).

Every number bounds  max|hip - ref| / max|ref|  of a stage tensor against the REAL reference's golden values (or the oracle), in every
fp32-grade arithmetic mode (f32, bf16x6, f16x3).

TWO checkpoints, two levels (round 6):

* the WELL-CONDITIONED synthetic checkpoint (vps_amd.synth.conditioned_overrides; golden of the real reference at 1024x2048:
  tests/golden/fusetrack_fullsize_cond.npz): CONDITIONED = 1e-4 for EVERY stage tensor and every detection score - the number
  BASELINE.md section 3 names. Measured: flow 5e-6, FPN 2.5e-6, neck 6e-6 .. 1e-5, semantic logits 7e-6, box-head logits <= 3.7e-5, scores
  1e-6 (profiles/r06_fullsize_conditioned_report.txt). bf16x3 (bf16 operands: what BASELINE config 5 words): 5e-4, measured <= 2.5e-4.

* the plain seeded weights (STAGE below). Behind the FPN that network amplifies rounding noise: fine flows of tens of pixels, deformable
  offsets of several pixels, attention logits of dozens - the fp32 ORACLE itself is 1.9e-4 (1024x2048) from its float64 evaluation at the
  neck (tools/neck_isolation.py, tools/condition_search.py), and the EXACT-fp32 kernels reach 1.7e-3 on one frame of the dense fixture
  (frame 4: neck 1.7e-3, semantic logits 1.9e-3 - in round 5 and in round 6, profiles/r0[56]_fullsize_sep_report.txt). The numbers
  below are what that noise allows, not what the kernels deliver; round 6 tried 1e-3 for the neck and the exact-fp32 mode
  failed it on that frame - the tight gate is the conditioned checkpoint.

  stage                     tolerance   measured (ResNet-50)     note
  flow (FlowNet2, full res)   5e-5      1.5e-6 .. 7.1e-6         5 networks, 60 convolutions: plain summation-order noise
  FPN levels P2..P6           5e-5      1.0e-6 .. 2.9e-6         (same for ResNet-101)
  fusion-neck outputs         2e-3      2.1e-5 .. 1.7e-3
  semantic logits fcn_score   2e-3      6.2e-5 .. 1.9e-3
  cls_score / bbox_pred       2e-3      5.0e-5 .. 9e-4
  detection scores            2e-3      <= 1.54e-3 (absolute: they are probabilities behind a softmax)
  ResNet-101 (config 5): the stages BEHIND the FPN 1e-2: the 101-layer synthetic network amplifies the same noise ~5x, in the exact-fp32
  kernels as much as in the split modes (neck <= 4.5e-3, fcn_score <= 8.2e-3, scores <= 8.3e-3 - all three maxima in bf16x6, the mode
  with the SMALLEST per-product error: summation-order noise, it moves with every change of a split-K partition)

Maps (fraction of differing pixels): semantic 1e-3; panoptic 1e-3, except the two fixtures whose margins cover every LISTING decision but
not which source proposal stands behind a detection - a boundary strip of single instances differs there in every mode, the exact-fp32
kernels included: dense 5e-3 (measured <= 2.6e-3), config5 1.5e-2 (measured <= 6e-3 in round 5, 1.08e-2 in bf16x6 after round 6's uneven
split-K changed the summation order of four decoder layers)."""

STAGE = dict(flow=5e-5, fpn=5e-5, neck=2e-3, fcn_score=2e-3, cls_score=2e-3, bbox_pred=2e-3, score=2e-3)
STAGE_R101 = dict(neck=1e-2, fcn_score=1e-2, cls_score=1e-2, bbox_pred=1e-2, score=1e-2)      # the stages behind the FPN of the 101-layer model
# every stage tensor and every detection score on the WELL-CONDITIONED synthetic checkpoint (vps_amd.synth.conditioned_overrides; golden of the
# real reference: tests/golden/fusetrack_fullsize_cond.npz) - the 1e-4 of BASELINE.md section 3
CONDITIONED = 1e-4
# the same for bf16x3 - bf16 operands, 2^-16 per product: the arithmetic BASELINE config 5 names, at the tolerance it can hold (measured <= 2.5e-4)
CONDITIONED_BF16X3 = 5e-4
MAP = dict(sem=1e-3, pan=1e-3, pan_dense=5e-3, pan_config5=1.5e-2)


def stage_tol(name, depth=50):
    """tolerance of a stage tensor by the name the tests report it under (flow, fpn_p2, neck_p6, fcn_score, cls_score, ...)"""
    key = name.split('_p')[0] if name.startswith(('fpn', 'neck')) else name
    return STAGE_R101[key] if depth > 50 and key in STAGE_R101 else STAGE[key]
