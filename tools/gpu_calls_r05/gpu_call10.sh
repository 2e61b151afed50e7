mkdir -p gpurun_out; R=$PWD
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q --tb=short -rf -p no:cacheprovider -s -k "correlation" > gpurun_out/c10_pytest_corr.log 2>&1; grep -E "correlation f16|passed|failed|Error" gpurun_out/c10_pytest_corr.log | tail -12
timeout 1200 python -m pytest tests/test_fusetrack_gpu.py tests/test_fullsize_gpu.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "stage_tensors or outputs_match or fan_out or golden or workspace" > gpurun_out/c10_pytest.log 2>&1; tail -6 gpurun_out/c10_pytest.log
for cf in 1 0 1 0; do
VPS_CORR_F16=$cf timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/c10_bench_cf$cf.json 2> gpurun_out/c10_bench_cf$cf.err
python -c "
import json;j=json.loads(open('gpurun_out/c10_bench_cf$cf.json').read().strip().splitlines()[-1]);r=j['roofline'];print('corr_f16 $cf', j['value'], 'frames/s', 'conv_ms', r['conv_ms_per_frame'], 'nonconv', r['in_frame_non_conv_ms'], r['in_frame_launch_us'].get('vps_correlation_f16'), r['in_frame_launch_us'].get('vps_correlation'), 'fallbacks', j['f16_fallbacks'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace10 -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/trace10.err
cd $R
timeout 120 python tools/trace_gaps.py gpurun_out/trace10 --out gpurun_out/c10_frame_occupancy.json
find gpurun_out/trace10 -name "*kernel_trace.csv" -delete
