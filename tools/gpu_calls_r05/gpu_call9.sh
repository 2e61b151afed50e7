mkdir -p gpurun_out; R=$PWD
timeout 900 python -m pytest tests/test_fusetrack_gpu.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "fan_out or clip_shard_backend or pooled or streamed" > gpurun_out/c9_pytest.log 2>&1; tail -5 gpurun_out/c9_pytest.log
for d in 1 2 1 2; do
VPS_PREFETCH_DEPTH=$d timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/c9_bench_d$d.json 2> gpurun_out/c9_bench_d$d.err
python -c "
import json;j=json.loads(open('gpurun_out/c9_bench_d$d.json').read().strip().splitlines()[-1]);r=j['roofline'];print('depth $d', j['value'], 'frames/s', 'conv_ms', r['conv_ms_per_frame'], 'nonconv', r['in_frame_non_conv_ms'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace9 -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/trace9.err
cd $R
timeout 120 python tools/trace_gaps.py gpurun_out/trace9 --out gpurun_out/c9_frame_occupancy.json
find gpurun_out/trace9 -name "*kernel_trace.csv" -delete
