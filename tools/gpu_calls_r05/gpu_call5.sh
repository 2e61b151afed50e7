mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fusetrack_gpu.py tests/test_hip_ops.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "pooled or tcea or mask_removal or stage_tensors or outputs_match" > gpurun_out/c5_pytest.log 2>&1; tail -6 gpurun_out/c5_pytest.log
for sk in 512 256 128 1024; do
VPS_SPLITK_TARGET=$sk timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/c5_bench_sk$sk.json 2> gpurun_out/c5_bench_sk$sk.err
python -c "
import json;j=json.loads(open('gpurun_out/c5_bench_sk$sk.json').read().strip().splitlines()[-1]);r=j['roofline'];print('splitk $sk', j['value'], 'frames/s', 'conv_ms', r['conv_ms_per_frame'], 'nonconv', r['in_frame_non_conv_ms'], 'launches', r['launches_per_frame'], {k:v for k,v in r['in_frame_launch_us'].items() if 'mask_rem' in k or 'modulate' in k})"
done
