mkdir -p gpurun_out
timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --conv-table gpurun_out/c25_conv_table.txt > gpurun_out/c25_bench.json 2> gpurun_out/c25.err; tail -3 gpurun_out/c25.err
python - <<PY
import json
d = json.loads(open('gpurun_out/c25_bench.json').read().strip().splitlines()[-1])
r = d['roofline']
print(d['value'], r['frac'], r['conv_ms_per_frame'], r["instrumented_frames"], r["launches_per_frame"], r['in_frame_non_conv_ms'], d['stage_ms'])
print(len(json.load(open('gpurun_out/c25_conv_table.txt.ordered.json'))))
PY
head -4 gpurun_out/c25_conv_table.txt
