mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_config2_inject_gpu.py tests/test_fusetrack_gpu.py -q -x -k "rpn or track or inject or graph or clip_shard or streamed or golden" > gpurun_out/g24_t.log 2>&1; tail -4 gpurun_out/g24_t.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/g24_b.json 2> gpurun_out/g24_b.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/g24_b.json').read().strip().splitlines()[-1])
u=j['roofline']['in_frame_launch_us']
print(j['value'], j['ms_per_step'], j['clip30']['id_checksum'], 'rpn_select', u.get('vps_rpn_select'), 'track', u.get('vps_track_assign'), 'non-conv ms', j['roofline']['in_frame_non_conv_ms'])
PY
