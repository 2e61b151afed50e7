mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_hip_ops.py -x -q -k "rpn" -p no:cacheprovider > gpurun_out/g8_rpn.log 2>&1; tail -3 gpurun_out/g8_rpn.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --conv-table gpurun_out/g8_ct.txt > gpurun_out/g8_b.json 2> gpurun_out/g8_b.err; head -c 200 gpurun_out/g8_b.json; echo; tail -3 gpurun_out/g8_b.err
timeout 900 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider -x > gpurun_out/g8_pytest_gpu.log 2>&1; tail -8 gpurun_out/g8_pytest_gpu.log
