# round 6: selected GPU tests + one bench line (used after host-side changes)
mkdir -p gpurun_out
timeout 1500 python -m pytest $TESTS -m gpu -x -q -p no:cacheprovider 2>&1 | tail -12
timeout 400 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --conv-table gpurun_out/cb_conv_table.txt > gpurun_out/cb_bench.json 2> gpurun_out/cb.err; tail -2 gpurun_out/cb.err
python - <<PY
import json
d = json.loads(open('gpurun_out/cb_bench.json').read().strip().splitlines()[-1])
r = d['roofline']
print(d['value'], 'frac', r['frac'], 'conv_ms', r['conv_ms_per_frame'], 'nonconv', r['in_frame_non_conv_ms'], d.get('test_vpq_loop'))
PY
