# narrow 3x3 layers (predict_flow): tests + the layer shapes alone (the kernel this was A/B-ed against - VPS_SMALL3X3_V=0 - has been removed:
# 194->2 @256x512 55 -> 35 us, 16->2 @1024x2048 71 -> 42 us, frame 55.60 -> 55.87 frames/s in one call)
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -p no:cacheprovider -k "conv2d_matches or conv_transpose or concat_window" 2>&1 | tail -2
BENCH_CONV_FILTER=predict_flow BENCH_CONV_REPS=20 python tools/bench_conv.py 4 2>&1 | grep -v "Warning\|amdgpu.ids"
