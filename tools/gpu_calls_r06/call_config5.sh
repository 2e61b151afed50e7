# BASELINE config 5 (ResNet-101, 1088x1920) in the two arithmetics: bf16 (as north_star words it) and f16x3 (the fp32-grade mode)
mkdir -p gpurun_out
for p in f16x3 bf16; do
timeout 400 python bench.py --model-config configs/viper/fusetrack_r101.py --height 1088 --width 1920 --prec $p --steps 40 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r06_bench_config5_r101_1088x1920_$p.json 2>> gpurun_out/r06_bench_config5.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r06_bench_config5_r101_1088x1920_$p.json').read().strip().splitlines()[-1])
print('$p', d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'])
PY
done
timeout 900 python -m pytest tests/test_config5_r101.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
