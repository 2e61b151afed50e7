# the stride-1 correlation with half a wavefront per dot product (VPS_CORR_HALF) against correlation4_kernel: tests, the shape alone, the frame
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_ref_native_gpu.py -m gpu -x -q -p no:cacheprovider -k "correlation" 2>&1 | tail -2
for v in 1 0 1 0; do echo "== VPS_CORR_HALF=$v"; VPS_CORR_HALF=$v python tools/bench_corr.py 2>&1 | grep -v "Warning\|amdgpu.ids" | grep Lite; done
bash tools/gpu_calls_r06/call_ab.sh VPS_CORR_HALF 0 1 40
