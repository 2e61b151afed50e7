mkdir -p gpurun_out; R=$PWD; T=r04
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o pmc -- python $R/bench.py --steps 1 --warmup 1 --single-stream --no-cpu-baseline --no-extras --conv-table $R/gpurun_out/conv_table_pmc.txt > /dev/null 2> $R/gpurun_out/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o pmc -- python $R/bench.py --steps 1 --warmup 1 --single-stream --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/pmc_write.err
cd $R
timeout 120 python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 5 > gpurun_out/${T}_pmc_traffic_f16x3.json 2> gpurun_out/pmc_traffic.err
timeout 120 python tools/pmc_per_layer.py gpurun_out/conv_table_pmc.txt.ordered.json gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/${T}_traffic_per_layer_f16x3.txt 2>> gpurun_out/pmc_traffic.err
cp gpurun_out/${T}_pmc_traffic_f16x3.json profiles/${T}_pmc_traffic_f16x3.json
timeout 600 python bench.py --steps 20 --warmup 5 --conv-table gpurun_out/${T}_conv_table_f16x3.txt > gpurun_out/${T}_bench_default_f16x3.json 2> gpurun_out/${T}_bench.err; head -c 160 gpurun_out/${T}_bench_default_f16x3.json; echo
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*kernel_trace.csv" -delete
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*counter_collection.csv" -size +20M -delete
