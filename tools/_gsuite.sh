mkdir -p gpurun_out; rm -f gpurun_out/fullsize_sep_report.txt
timeout 600 python -m pytest tests/test_fullsize_sep_gpu.py -m gpu -q --tb=line -k config5 > gpurun_out/g22_c5.log 2>&1; tail -3 gpurun_out/g22_c5.log
