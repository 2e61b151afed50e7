mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py -x -q -k "conv or linear or fpn" -p no:cacheprovider > gpurun_out/g1_pytest.log 2>&1; tail -4 gpurun_out/g1_pytest.log
VPS_DEBUG_OCC=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --conv-table gpurun_out/g1_ct_q3.txt > gpurun_out/g1_b_q3.json 2> gpurun_out/g1_b_q3.err; grep vps gpurun_out/g1_b_q3.err; head -c 300 gpurun_out/g1_b_q3.json; echo
VPS_UNIFORM_LEAD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --conv-table gpurun_out/g1_ct_q0.txt > gpurun_out/g1_b_q0.json 2> gpurun_out/g1_b_q0.err; head -c 300 gpurun_out/g1_b_q0.json; echo
VPS_UNIFORM_LEAD=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --conv-table gpurun_out/g1_ct_q1.txt > gpurun_out/g1_b_q1.json 2> gpurun_out/g1_b_q1.err; head -c 300 gpurun_out/g1_b_q1.json; echo
python tools/compare_conv_tables.py gpurun_out/g1_ct_q0.txt gpurun_out/g1_ct_q3.txt 0.03 > gpurun_out/g1_cmp.txt; tail -3 gpurun_out/g1_cmp.txt
