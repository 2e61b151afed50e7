mkdir -p gpurun_out
for mc in 0 3 5; do
VPS_H8_MIN_CHUNKS=$mc BENCH_CONV_FILTER="3x3" timeout 300 python tools/bench_conv.py 4 > gpurun_out/c7_conv_mc$mc.txt 2>&1
done
paste <(grep ms gpurun_out/c7_conv_mc0.txt | awk -F'  +' '{print $1}') <(grep ms gpurun_out/c7_conv_mc0.txt | awk '{for(i=1;i<=NF;i++) if($i=="ms") print $(i-1)}') <(grep ms gpurun_out/c7_conv_mc3.txt | awk '{for(i=1;i<=NF;i++) if($i=="ms") print $(i-1)}') <(grep ms gpurun_out/c7_conv_mc5.txt | awk '{for(i=1;i<=NF;i++) if($i=="ms") print $(i-1)}')
for mc in 0 3 5; do
VPS_H8_MIN_CHUNKS=$mc timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/c7_bench_mc$mc.json 2> gpurun_out/c7_bench_mc$mc.err
python -c "
import json;j=json.loads(open('gpurun_out/c7_bench_mc$mc.json').read().strip().splitlines()[-1]);r=j['roofline'];print('min_chunks $mc', j['value'], 'frames/s', 'conv_ms', r['conv_ms_per_frame'])"
done
