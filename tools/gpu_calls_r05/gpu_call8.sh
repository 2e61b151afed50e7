mkdir -p gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace8 -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/trace8.err
cd $R
timeout 120 python tools/trace_gaps.py gpurun_out/trace8 --out gpurun_out/c8_frame_occupancy.json
find gpurun_out/trace8 -name "*kernel_trace.csv" -delete
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/c8_bench.json 2> gpurun_out/c8_bench.err
python -c "
import json;j=json.loads(open('gpurun_out/c8_bench.json').read().strip().splitlines()[-1]);r=j['roofline'];print(j['value'], 'frames/s', 'conv_ms', r['conv_ms_per_frame'])"
