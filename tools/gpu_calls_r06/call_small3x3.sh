# narrow 3x3 layers (predict_flow): tests + the layer shapes alone, the instruction-lean instance (VPS_SMALL3X3_V=1) against the kernel it replaces
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -p no:cacheprovider -k "conv2d_matches or conv_transpose or concat_window" 2>&1 | tail -2
for v in 1 0 1 0; do echo "== VPS_SMALL3X3_V=$v"; VPS_SMALL3X3_V=$v BENCH_CONV_FILTER=predict_flow BENCH_CONV_REPS=20 python tools/bench_conv.py 4 2>&1 | grep -v "Warning\|amdgpu.ids"; done
