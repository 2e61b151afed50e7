mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_fusetrack_gpu.py tests/test_fullsize_gpu.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "fan_out or clip_shard_backend or pooled or deterministic or workspace or outputs_match" > gpurun_out/c17_pytest.log 2>&1; tail -3 gpurun_out/c17_pytest.log
for i in 1 2; do
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/c17_bench_$i.json 2> gpurun_out/c17_bench_$i.err
python -c "
import json;j=json.loads(open('gpurun_out/c17_bench_$i.json').read().strip().splitlines()[-1]);print('run $i', j['value'], 'frames/s')"
done
