mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_hip_ops.py tests/test_fullsize_sep_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/c23_bench.json 2> gpurun_out/c23.err; head -c 260 gpurun_out/c23_bench.json; echo
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c23_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['traffic_measured_on_these_kernel_sources'])
PY
