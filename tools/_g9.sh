mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_hip_ops.py tests/test_pipeline.py tests/test_fusetrack_gpu.py -x -q -k "rpn or png or feeder or combine or panoptic or clip" -p no:cacheprovider > gpurun_out/g9_t.log 2>&1; tail -3 gpurun_out/g9_t.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/g9_b.json 2> gpurun_out/g9_b.err; head -c 200 gpurun_out/g9_b.json; echo; tail -3 gpurun_out/g9_b.err
