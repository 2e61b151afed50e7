mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/c16_pytest.log 2>&1; tail -4 gpurun_out/c16_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c16_smoke.log 2>&1; tail -2 gpurun_out/c16_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c16_bench_driver_cmd.json 2> gpurun_out/c16_bench.err
python -c "
import json;j=json.loads(open('gpurun_out/c16_bench_driver_cmd.json').read().strip().splitlines()[-1]);r=j['roofline'];print('driver cmd', j['value'], 'frames/s', j['ms_per_step'], 'frac', r['frac'], 'traffic stamped', r.get('traffic_measured_on_these_kernel_sources'), 'cpu', j['cpu_baseline']['value'])"
