mkdir -p gpurun_out; R=$PWD
timeout 900 python -m pytest tests/test_fusetrack_gpu.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "fan_out or clip_shard_backend or pooled" > gpurun_out/c12_pytest.log 2>&1; tail -3 gpurun_out/c12_pytest.log
for i in 1 2; do
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/c12_bench_$i.json 2> gpurun_out/c12_bench_$i.err
python -c "
import json;j=json.loads(open('gpurun_out/c12_bench_$i.json').read().strip().splitlines()[-1]);r=j['roofline'];print('run $i', j['value'], 'frames/s', 'conv_ms', r['conv_ms_per_frame'], 'nonconv', r['in_frame_non_conv_ms'])"
done
VPS_PREFETCH_DEPTH=1 timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/c12_bench_d1.json 2> gpurun_out/c12_bench_d1.err
python -c "
import json;j=json.loads(open('gpurun_out/c12_bench_d1.json').read().strip().splitlines()[-1]);print('depth 1', j['value'], 'frames/s')"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace12 -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/trace12.err
cd $R
timeout 120 python tools/trace_gaps.py gpurun_out/trace12 --out gpurun_out/c12_frame_occupancy.json
find gpurun_out/trace12 -name "*kernel_trace.csv" -delete
