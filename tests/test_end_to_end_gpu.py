"""GPU: the whole drop-in chained like tools/test_vpq.py + tools/eval_vpq.py on a synthetic video (tools/run_vps_synthetic.py):
uint8 frames -> device input preparation -> detector -> device unifier -> device converter + asynchronous PNG / json writer ->
device-counted VPQ. With the prediction as its own ground truth every window length must score exactly 100; with the exact-fp32
kernels as ground truth the benchmarked f16x3 arithmetic must score > 99 (the reference's own acceptance metric as the parity
number: the north star allows 0.1 VPQ on real data)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

pytestmark = pytest.mark.gpu


def _run(tmp_path, extra):
    import run_vps_synthetic as R
    argv = sys.argv
    sys.argv = ['run_vps_synthetic.py', '--videos', '2', '--frames', '30', '--height', '128', '--width', '256', '--out', str(tmp_path)] + extra
    try:
        return R.main()
    finally:
        sys.argv = argv


def test_synthetic_video_set_end_to_end_scores_100_against_itself(dev, tmp_path):
    rep = _run(tmp_path, [])
    assert rep['labelled_frames'] == 12 and rep['png_files'] == 12
    assert rep['vpq'] == 100.0 and all(v == 100.0 for v in rep['pq_per_window'].values()), rep
    for sub in ('pan_pred', 'pan_2ch'):
        assert len(os.listdir(tmp_path / 'pred' / sub)) == 12
    pj = json.load(open(tmp_path / 'pred' / 'pred.json'))
    assert len(pj['annotations']) == 12 and all(a['segments_info'] for a in pj['annotations'])


def test_f16x3_against_exact_fp32_in_the_reference_metric(dev, tmp_path):
    rep = _run(tmp_path, ['--prec', 'f16x3', '--gt-prec', 'f32'])
    print(rep)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'vpq_f16x3_vs_f32.json'), 'w') as f:
        json.dump(rep, f, indent=1)
    assert rep['vpq'] > 99.0, rep
