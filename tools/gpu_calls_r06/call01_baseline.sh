# round 6, call 1: same-box baseline of the round-5 tree (bench line + per-layer conv table)
mkdir -p gpurun_out
timeout 400 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --conv-table gpurun_out/c01_conv_table.txt > gpurun_out/c01_bench.json 2> gpurun_out/c01.err; tail -3 gpurun_out/c01.err
python - <<PY
import json
d = json.loads(open('gpurun_out/c01_bench.json').read().strip().splitlines()[-1])
r = d['roofline']
print(d['value'], r['frac'], r['conv_ms_per_frame'], r['in_frame_non_conv_ms'], d['stage_ms'])
PY
