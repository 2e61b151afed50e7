mkdir -p gpurun_out
timeout 400 python bench.py --steps 100 --warmup 5 --conv-table gpurun_out/r05_conv_table_f16x3.txt > gpurun_out/r05_bench_default_f16x3.json 2> gpurun_out/c22.err; head -c 300 gpurun_out/r05_bench_default_f16x3.json; echo
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_command_steps20.json 2>> gpurun_out/c22.err; head -c 300 gpurun_out/r05_bench_driver_command_steps20.json; echo
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
