set -x
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -p no:cacheprovider -k "conv2d_matches or conv_transpose or concat_window" 2>&1 | tail -5
for m in 1 0; do echo "== VPS_SMALL_BATCHED=$m"; VPS_SMALL_BATCHED=$m BENCH_CONV_FILTER=narrow BENCH_CONV_REPS=20 python tools/bench_conv.py 4 2>&1 | grep -v Warning; done
