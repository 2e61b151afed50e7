mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_rccl_gpu.py::test_bench_rccl_branch_at_world_1 tests/test_config3.py -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/c2_pytest.log 2>&1; tail -15 gpurun_out/c2_pytest.log
timeout 600 python tools/exp_two_clips.py --frames 40 > gpurun_out/c2_two_clips.json 2> gpurun_out/c2_two_clips.err; cat gpurun_out/c2_two_clips.json
