mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_hip_ops.py -x -q -k "conv or linear or fpn or rpn or deform" -p no:cacheprovider > gpurun_out/g4_pytest.log 2>&1; tail -6 gpurun_out/g4_pytest.log
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --conv-table gpurun_out/g4_ct_q3.txt > gpurun_out/g4_b_q3.json 2> gpurun_out/g4_b_q3.err; head -c 200 gpurun_out/g4_b_q3.json; echo; tail -3 gpurun_out/g4_b_q3.err
VPS_UNIFORM_LEAD=0 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --conv-table gpurun_out/g4_ct_q0.txt > gpurun_out/g4_b_q0.json 2> gpurun_out/g4_b_q0.err; head -c 200 gpurun_out/g4_b_q0.json; echo
python tools/compare_conv_tables.py gpurun_out/g1_ct_q0.txt gpurun_out/g4_ct_q3.txt 0.03 > gpurun_out/g4_cmp.txt; tail -1 gpurun_out/g4_cmp.txt
python tools/compare_conv_tables.py gpurun_out/g1_ct_q0.txt gpurun_out/g4_ct_q0.txt 0.03 > gpurun_out/g4_cmp0.txt; tail -1 gpurun_out/g4_cmp0.txt
