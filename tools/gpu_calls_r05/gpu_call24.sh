mkdir -p gpurun_out
for i in 1 2; do
timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/c24_bench_$i.json 2> gpurun_out/c24.err
python - <<PY
import json
d = json.loads(open('gpurun_out/c24_bench_$i.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['conv_ms_per_frame'], d['stage_ms'])
PY
done
