# round-5 GPU call 1: new stream schedule (tests + A/B), RCCL world-1 tests, pre-split knock-out, neck isolation
mkdir -p gpurun_out; R=$PWD
timeout 1200 python -m pytest tests/test_rccl_gpu.py "tests/test_fusetrack_gpu.py::test_image_stage_stream_fan_out_is_bitwise_the_single_prefetch_stream" "tests/test_fusetrack_gpu.py::test_clip_shard_backend_and_handoff_feature" -q --tb=short -rf -p no:cacheprovider > gpurun_out/c1_pytest.log 2>&1; tail -5 gpurun_out/c1_pytest.log
for cfg in "1 0" "2 0" "3 0" "3 1" "1 1" "3 0"; do
  set -- $cfg
  VPS_PRE_STREAMS=$1 VPS_MAIN_PRIO=$2 timeout 400 python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/c1_bench_s$1_p$2.json 2> gpurun_out/c1_bench_s$1_p$2.err
  python -c "import json;j=json.loads(open('gpurun_out/c1_bench_s$1_p$2.json').read().strip().splitlines()[-1]);print('streams $1 prio $2:', j['value'], 'frames/s', j['ms_per_step'], 'ms')"
done
timeout 300 python bench.py --gpus 2 > gpurun_out/c1_bench_gpus2.json 2>&1; echo "gpus2 rc=$?"; cat gpurun_out/c1_bench_gpus2.json
timeout 300 python tools/bench_conv.py 4 > gpurun_out/c1_conv_base.txt 2>&1; mv gpurun_out/bench_conv_p4.json gpurun_out/c1_conv_base.json
VPS_HIP_LIB=$R/build/nosplit/libvpship.so timeout 300 python tools/bench_conv.py 4 > gpurun_out/c1_conv_nosplit.txt 2>&1; mv gpurun_out/bench_conv_p4.json gpurun_out/c1_conv_nosplit.json
paste <(awk '{print $(NF-6), $(NF-5)}' gpurun_out/c1_conv_base.txt) gpurun_out/c1_conv_nosplit.txt | head -40
timeout 600 python tools/neck_isolation.py > gpurun_out/c1_neck_isolation.log 2>&1; tail -6 gpurun_out/c1_neck_isolation.log
