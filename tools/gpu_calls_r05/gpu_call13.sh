mkdir -p gpurun_out; R=$PWD
timeout 900 python -m pytest tests/test_fusetrack_gpu.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "fan_out or clip_shard_backend or pooled" > gpurun_out/c13_pytest.log 2>&1; tail -3 gpurun_out/c13_pytest.log
for db in 1 0 1 0; do
VPS_DEFER_BACKBONE=$db timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/c13_bench_db$db.json 2> gpurun_out/c13_bench_db$db.err
python -c "
import json;j=json.loads(open('gpurun_out/c13_bench_db$db.json').read().strip().splitlines()[-1]);print('defer_backbone $db', j['value'], 'frames/s')"
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace13 -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/trace13.err
cd $R
timeout 120 python tools/trace_gaps.py gpurun_out/trace13 --out gpurun_out/c13_frame_occupancy.json
find gpurun_out/trace13 -name "*kernel_trace.csv" -delete
