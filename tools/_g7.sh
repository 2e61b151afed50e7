mkdir -p gpurun_out
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --conv-table gpurun_out/g7_ct.txt > gpurun_out/g7_b.json 2> gpurun_out/g7_b.err; head -c 250 gpurun_out/g7_b.json; echo; tail -3 gpurun_out/g7_b.err
timeout 1200 python tools/vpq_attribution.py --height 1024 --width 2048 --videos 2 --frames 30 > gpurun_out/g7_vpq_attribution.json 2> gpurun_out/g7_vpq.err; tail -4 gpurun_out/g7_vpq.err | cut -c1-1500
