# A/B of one environment switch inside ONE call (boxes differ by +-1.3 %): usage: call_ab.sh VAR A_VALUE B_VALUE [steps]
mkdir -p gpurun_out
V=$1; A=$2; B=$3; S=${4:-40}
for rep in 1 2; do
for val in $A $B; do
env $V=$val timeout 400 python bench.py --gpus 1 --steps $S --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/ab_$val.json 2> gpurun_out/ab_$val.err
python - <<PY
import json
d = json.loads(open('gpurun_out/ab_$val.json').read().strip().splitlines()[-1])
r = d['roofline']
print('$V=$val', d['value'], 'frac', r['frac'], 'conv_ms', r['conv_ms_per_frame'], 'nonconv', r['in_frame_non_conv_ms'])
PY
done
done
