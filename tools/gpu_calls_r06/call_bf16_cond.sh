# plain bf16 / bf16x3 (config 5's wording) on the well-conditioned checkpoint against the real-reference golden at 1024x2048: stage errors only
mkdir -p gpurun_out; rm -f gpurun_out/fullsize_conditioned_report.txt
python - <<'PY'
import sys, os
sys.path.insert(0, 'tests')
import torch
import test_fullsize_gpu as T
dev = torch.device('cuda:0')
for p in ('bf16', 'bf16x3'):
    try:
        T.test_every_stage_within_1e_4_of_the_reference_on_the_conditioned_checkpoint(dev, p)
    except AssertionError as e:
        print(p, 'exceeds 1e-4 (expected):', str(e)[:300])
PY
cat gpurun_out/fullsize_conditioned_report.txt
