mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -x -p no:cacheprovider > gpurun_out/c3_pytest.log 2>&1; tail -15 gpurun_out/c3_pytest.log
timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err
python -c "
import json;j=json.loads(open('gpurun_out/c3_bench.json').read().strip().splitlines()[-1]);print(j['value'], 'frames/s', 'workspace_GB', j['config']['workspace_GB'], 'clip30', j.get('clip30'), 'vpq_loop', j.get('test_vpq_loop'), 'png', (j.get('from_png') or {}).get('ratio_to_resident'))"
tail -3 gpurun_out/c3_bench.err
