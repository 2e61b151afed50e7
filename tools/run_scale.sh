#!/bin/bash
# The multi-GPU measurement of a node in ONE command (BASELINE config 4 + the 1/2/4/8 curve of north_star), for the first box that has
# more than one MI355X:
#     bash tools/run_scale.sh            # N = 1 2 4 8 (those the node has), weak scaling (K frames per rank) + the fixed 30-frame clip
#     NS="1 2" STEPS=50 bash tools/run_scale.sh
# Every run is `python bench.py --gpus N`: for N > 1 bench.py re-launches itself under torch.distributed.run (one rank per GPU, RCCL
# over xGMI, rendezvous on 127.0.0.1) and rank 0 prints ONE JSON line carrying `n_gpus`, `rccl_ranks` (all-reduce of ones), `value`
# (whole-job frames/s, weak) and `clip30` (the 30-frame clip sharded over the N GPUs: strong scaling, id checksum). Lines go to
# gpurun_out/scale_N<N>.json; the table at the end is computed from them (efficiency = value_N / (N * value_1)). The clip30 id
# checksum must be the same for every N: the sharded pipeline reproduces the sequential ids.
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
NS=${NS:-"1 2 4 8"}; STEPS=${STEPS:-100}; WARMUP=${WARMUP:-5}
NDEV=$(python -c "import torch; print(torch.cuda.device_count() if torch.cuda.is_available() else 0)")
echo "devices visible: $NDEV"
for N in $NS; do
    if [ "$N" -gt "$NDEV" ]; then echo "N=$N skipped: the node has $NDEV GPU(s)"; continue; fi
    extra="--no-cpu-baseline"; [ "$N" = "1" ] && extra=""
    timeout 1200 python bench.py --gpus $N --steps $STEPS --warmup $WARMUP $extra > gpurun_out/scale_N$N.json 2> gpurun_out/scale_N$N.err
    echo "N=$N rc=$? $(head -c 160 gpurun_out/scale_N$N.json)"
done
python - <<'PY'
import glob, json, re
rows = {}
for f in sorted(glob.glob('gpurun_out/scale_N*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable:', e); continue
    if 'error' in j:
        print(f, j['error']); continue
    rows[j['n_gpus']] = j
if 1 in rows:
    v1 = rows[1]['value']
    print('%4s %12s %10s %12s %14s %10s' % ('N', 'frames/s', 'weak eff', 'clip30 f/s', 'clip30 speedup', 'ranks'))
    for n, j in sorted(rows.items()):
        c = j.get('clip30') or {}
        c1 = (rows[1].get('clip30') or {}).get('frames_per_s')
        print('%4d %12.2f %10.3f %12s %14s %10s' % (n, j['value'], j['value'] / (n * v1), c.get('frames_per_s', '-'),
              ('%.2f' % (c['frames_per_s'] / c1)) if c and c1 else '-', j.get('rccl_ranks', 1)))
    sums = {n: (j.get('clip30') or {}).get('id_checksum') for n, j in rows.items()}
    print('clip30 id checksums:', sums, 'EQUAL' if len(set(sums.values())) == 1 else 'DIFFERENT')
PY
