mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_hip_ops.py -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/c15_pytest.log 2>&1; tail -5 gpurun_out/c15_pytest.log
