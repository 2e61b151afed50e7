#!/bin/bash
# GPU box: per-layer HBM-side traffic of the conv launches (two --pmc passes + the launch order of the instrumented frame)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-f16x3}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o pmc -- python $R/bench.py --steps 1 --warmup 1 --single-stream --no-cpu-baseline --no-extras --conv-table $R/gpurun_out/conv_table_pmc.txt > /dev/null 2> $R/gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o pmc -- python $R/bench.py --steps 1 --warmup 1 --single-stream --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/pmc_write.err
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 5 > gpurun_out/pmc_traffic_$TAG.json 2> gpurun_out/pmc_traffic.err
python tools/pmc_per_layer.py gpurun_out/conv_table_pmc.txt.ordered.json gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/traffic_per_layer_$TAG.txt 2>> gpurun_out/pmc_traffic.err
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*kernel_trace.csv" -delete
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*counter_collection.csv" -size +20M -delete
