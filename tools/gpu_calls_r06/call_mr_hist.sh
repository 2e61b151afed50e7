# MaskRemoval without a dependency chain (VPS_MASK_REMOVAL=hist, vps_mask_removal_hist) against the one-launch dependency kernel (dep): tests, the walk alone, frame A/B
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_config2_inject_gpu.py -m gpu -x -q -p no:cacheprovider -k "mask_removal or inject or mask" 2>&1 | tail -3
python tools/bench_mask_removal.py 2>&1 | grep -v amdgpu.ids
bash tools/gpu_calls_r06/call_ab.sh VPS_MASK_REMOVAL dep hist 40
python - <<'PY'
import json
for v in ('dep', 'hist'):
    d = json.loads(open('gpurun_out/ab_%s.json' % v).read().strip().splitlines()[-1])
    print(v, {k: d['roofline']['in_frame_launch_us'].get(k) for k in ('vps_mask_removal_dep', 'vps_mask_removal_hist', 'vps_panoptic_combine_dev', 'vps_pan_instances')})
PY
