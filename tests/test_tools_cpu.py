"""CPU checks of the measurement tools that post-process GPU-side output (so a broken parser is found here, not on the GPU box)."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trace_gaps_on_a_synthetic_kernel_trace(tmp_path):
    """three frames of two overlapping 'streams': busy = union of the intervals, work = their sum, the gap classes add up to idle"""
    rows, t = [], 1000
    for fr in range(6):
        for k in range(10):
            rows.append((t, t + 90_000, 'conv'))                 # main stream: 90 us kernels, 10 us apart
            rows.append((t + 20_000, t + 60_000, 'side'))        # side stream: fully inside the main kernel
            t += 100_000
        rows.append((t, t + 50_000, 'void (anonymous namespace)::panoptic_combine_kernel(float const*)'))
        t += 50_000
    d = tmp_path / 'trace' / 'box'
    d.mkdir(parents=True)
    with open(d / '1_kernel_trace.csv', 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Kind', 'Start_Timestamp', 'End_Timestamp', 'Kernel_Name'])
        for a, b, n in rows:
            w.writerow(['KERNEL_DISPATCH', a, b, n])
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'trace_gaps.py'), str(tmp_path / 'trace')], capture_output=True, text=True, check=True)
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r['frames'] == 5 and abs(r['period_ms'] - 1.05) < 1e-6
    assert abs(r['busy_ms'] - 0.95) < 1e-6 and abs(r['idle_ms'] - 0.10) < 1e-6          # 10 gaps of 10 us
    assert abs(r['work_ms'] - (0.95 + 0.40)) < 1e-6 and r['launches_per_frame'] == 21
    assert r['idle_gaps']['10-50us'] == [10, 0.1] and r['idle_gaps']['<2us'][0] == 0
