mkdir -p gpurun_out
VPS_GRAPH=1 timeout 500 python bench.py --steps 20 --warmup 6 --no-cpu-baseline > gpurun_out/g23_graph1.json 2> gpurun_out/g23_graph1.err; echo "rc=$?"; tail -3 gpurun_out/g23_graph1.err
VPS_GRAPH=0 timeout 500 python bench.py --steps 20 --warmup 6 --no-cpu-baseline > gpurun_out/g23_graph0.json 2> gpurun_out/g23_graph0.err; echo "rc=$?"
python - <<'PY'
import json
for n in ('1','0'):
    try:
        j=json.loads(open('gpurun_out/g23_graph%s.json'%n).read().strip().splitlines()[-1])
        print('VPS_GRAPH=%s'%n, j['value'], j['ms_per_step'], 'clip30', j['clip30']['frames_per_s'], j['clip30']['id_checksum'], 'vpq_loop', j['test_vpq_loop']['frames_per_s'], 'png', j['from_png']['frames_per_s'])
    except Exception as e:
        print('VPS_GRAPH=%s failed: %s'%(n,e))
PY
VPS_GRAPH=1 timeout 600 python -m pytest tests/test_fusetrack_gpu.py -q -x -k "clip_shard or pipelin or prefetch" > gpurun_out/g23_t.log 2>&1; tail -3 gpurun_out/g23_t.log
