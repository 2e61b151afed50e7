# The judged measurement set of a round in one gpurun call (writes gpurun_out/; copy what is to be judged into profiles/).
#   gpurun --timeout 1500 -- 'bash tools/run_round_measurements.sh'
#   SKIP_PYTEST=1 skips the GPU test suite (run it in its own call: it takes ~10 of the 90 GPU-minutes), WITH_VPQ=1 adds the two
#   60-frame VPQ runs (profiles/r04_vpq_attribution.json holds the numbers from tools/vpq_attribution.py)
# Every command sits under its own `timeout`: a hung counter pass must not take the rest of the call with it.
mkdir -p gpurun_out; R=$PWD; T=${TAG:-r06}
if [ -z "$SKIP_PYTEST" ]; then timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/${T}_pytest_gpu.log 2>&1; tail -6 gpurun_out/${T}_pytest_gpu.log; fi
timeout 900 python bench.py --steps 100 --warmup 5 --conv-table gpurun_out/${T}_conv_table_f16x3.txt > gpurun_out/${T}_bench_default_f16x3.json 2> gpurun_out/${T}_bench.err; head -c 200 gpurun_out/${T}_bench_default_f16x3.json; echo
# BASELINE config 5 (ResNet-101, 1088x1920): the fp32-grade mode, the bf16-operand mode that holds a stated tolerance (bf16x3) and plain bf16
for P in f16x3 bf16x3 bf16; do timeout 300 python bench.py --model-config configs/viper/fusetrack_r101.py --height 1088 --width 1920 --prec $P --steps 40 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/${T}_bench_config5_r101_1088x1920_$P.json 2>> gpurun_out/${T}_bench.err; head -c 120 gpurun_out/${T}_bench_config5_r101_1088x1920_$P.json; echo; done
# the host side of an 8-GPU node: 8 rank processes x 4 decode threads, pinned (tools/bench_input_pipeline.py --node)
timeout 200 python tools/bench_input_pipeline.py --node 8 --threads 4 --seconds 8 > gpurun_out/${T}_node_decode_8x4.json 2>> gpurun_out/${T}_bench.err
# the 2-rank clip pipeline, functionally (bench.py launches itself under torch.distributed.run): two processes on the ONE GPU of this box over gloo (NOT RCCL, no scaling number): the
# streamed records / maps / hand-off with the real DetectorBackend; its clip30 id_checksum must equal the 1-rank run's
VPS_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_2rank_gloo_one_gpu.json 2>> gpurun_out/${T}_bench.err
python - <<PY
import json
try:
    a=json.loads(open('gpurun_out/${T}_bench_default_f16x3.json').read().strip().splitlines()[-1]); b=json.loads(open('gpurun_out/${T}_bench_2rank_gloo_one_gpu.json').read().strip().splitlines()[-1])
    print('clip30 id_checksum 1 rank', a['clip30']['id_checksum'], '2 ranks', b['clip30']['id_checksum'], 'EQUAL' if a['clip30']['id_checksum']==b['clip30']['id_checksum'] else 'DIFFERENT')
except Exception as e:
    print('checksum comparison failed:', e)
PY
if [ -n "$WITH_VPQ" ]; then
timeout 600 python tools/run_vps_synthetic.py --height 1024 --width 2048 --videos 2 --frames 30 --prec f16x3 --gt-prec f32 --out gpurun_out/vps_near_tied > gpurun_out/${T}_vpq_f16x3_vs_f32_1024x2048_near_tied.json 2>> gpurun_out/${T}_bench.err
timeout 600 python tools/run_vps_synthetic.py --height 1024 --width 2048 --videos 2 --frames 30 --prec f16x3 --gt-prec f32 --separated --out gpurun_out/vps_separated > gpurun_out/${T}_vpq_f16x3_vs_f32_1024x2048_separated.json 2>> gpurun_out/${T}_bench.err
rm -rf gpurun_out/vps_near_tied gpurun_out/vps_separated
fi
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ss -o ss -- python $R/bench.py --steps 40 --warmup 3 --single-stream --no-cpu-baseline --no-extras > $R/gpurun_out/prof_ss.json 2> $R/gpurun_out/prof_ss.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o pmc -- python $R/bench.py --steps 1 --warmup 1 --single-stream --no-cpu-baseline --no-extras --conv-table $R/gpurun_out/conv_table_pmc.txt > /dev/null 2> $R/gpurun_out/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o pmc -- python $R/bench.py --steps 1 --warmup 1 --single-stream --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/pmc_write.err
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -o pmc -- python $R/bench.py --steps 1 --warmup 1 --single-stream --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/pmc_mfma.err
# (7 = frames of that command: 1 warm-up + 1 timed + 2 set-up frames + bench.py's 3 instrumented frames)
# occupancy of the timed frames (three streams): union and sum of the kernel intervals per frame interval, steady state only
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/trace.err
cd $R
timeout 120 python tools/trace_gaps.py gpurun_out/trace --out gpurun_out/${T}_frame_occupancy_traced.json
timeout 120 python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 7 > gpurun_out/${T}_pmc_traffic_f16x3.json 2> gpurun_out/pmc_traffic.err
timeout 120 python tools/pmc_mfma.py gpurun_out/pmc_mfma > gpurun_out/${T}_pmc_mfma_busy_f16x3.json 2> gpurun_out/pmc_mfma_tool.err
timeout 120 python tools/pmc_per_layer.py gpurun_out/conv_table_pmc.txt.ordered.json gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/${T}_traffic_per_layer_f16x3.txt 2>> gpurun_out/pmc_traffic.err
cp $(find gpurun_out/prof_ss -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_fusetrack_kernel_stats_single_stream_f16x3.csv
# keep the merged-back payload small: drop the raw per-dispatch traces
find gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/prof_ss gpurun_out/trace -name "*kernel_trace.csv" -delete
find gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma -name "*counter_collection.csv" -size +20M -delete
du -sh gpurun_out
