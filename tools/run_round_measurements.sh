mkdir -p gpurun_out; R=$PWD
if [ -z "$SKIP_PYTEST" ]; then python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/pytest6.log 2>&1; tail -6 gpurun_out/pytest6.log; fi
python bench.py --steps 20 --warmup 5 --conv-table gpurun_out/conv_table_r02_f16x3.txt > gpurun_out/bench6_default.json 2> gpurun_out/bench6.err; head -c 260 gpurun_out/bench6_default.json; echo
python bench.py --model-config configs/viper/fusetrack_r101.py --height 1088 --width 1920 --prec bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/bench6_config5_bf16.json 2>> gpurun_out/bench6.err; head -c 260 gpurun_out/bench6_config5_bf16.json; echo
python bench.py --model-config configs/viper/fusetrack_r101.py --height 1088 --width 1920 --prec f16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/bench6_config5_f16x3.json 2>> gpurun_out/bench6.err; head -c 260 gpurun_out/bench6_config5_f16x3.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ss -o ss -- python $R/bench.py --steps 10 --warmup 3 --single-stream --no-cpu-baseline --no-extras > $R/gpurun_out/prof_ss.json 2> $R/gpurun_out/prof_ss.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o pmc -- python $R/bench.py --steps 1 --warmup 1 --single-stream --no-cpu-baseline --no-extras --conv-table $R/gpurun_out/conv_table_pmc.txt > /dev/null 2> $R/gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o pmc -- python $R/bench.py --steps 1 --warmup 1 --single-stream --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -o pmc -- python $R/bench.py --steps 1 --warmup 1 --single-stream --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/pmc_mfma.err
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 5 > gpurun_out/pmc_traffic_f16x3.json 2> gpurun_out/pmc_traffic.err
python tools/pmc_mfma.py gpurun_out/pmc_mfma > gpurun_out/pmc_mfma_f16x3.json 2> gpurun_out/pmc_mfma_tool.err
python tools/pmc_per_layer.py gpurun_out/conv_table_pmc.txt.ordered.json gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/traffic_per_layer_f16x3.txt 2>> gpurun_out/pmc_traffic.err
find gpurun_out/prof_ss -name "*kernel_stats.csv" | head -2; du -sh gpurun_out
# keep the merged-back payload small: drop the raw per-dispatch traces
find gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/prof_ss -name "*kernel_trace.csv" -delete
find gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma -name "*counter_collection.csv" -size +20M -delete
du -sh gpurun_out
