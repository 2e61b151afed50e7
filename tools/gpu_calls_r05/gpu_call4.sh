mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "mask_removal or groupnorm" > gpurun_out/c4_pytest_ops.log 2>&1; tail -12 gpurun_out/c4_pytest_ops.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider --deselect tests/test_hip_ops.py > gpurun_out/c4_pytest.log 2>&1; tail -12 gpurun_out/c4_pytest.log
for mr in dep level; do
VPS_MASK_REMOVAL=$mr timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/c4_bench_$mr.json 2> gpurun_out/c4_bench_$mr.err
python -c "
import json;j=json.loads(open('gpurun_out/c4_bench_$mr.json').read().strip().splitlines()[-1]);r=j['roofline'];print('$mr', j['value'], 'frames/s', 'ws', j['config']['workspace_GB'], 'nonconv', r['in_frame_non_conv_ms'], {k:v for k,v in r['in_frame_launch_us'].items() if 'mask' in k or 'groupnorm' in k})"
done
