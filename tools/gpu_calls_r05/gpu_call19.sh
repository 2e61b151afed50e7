mkdir -p gpurun_out
timeout 200 python - <<'PY' 2>&1 | tail -8
import torch
from vps_amd import hip, nhwc
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
for (H, W) in ((256, 512), (128, 256), (100, 333)):
    w = torch.randn(256, 256, 3, 3, generator=g) * 0.05
    pc = nhwc.PackedConv(w, torch.randn(256, generator=g), None, stride=1, padding=1, act=hip.ACT_NONE, deform=True, device=dev, prec=hip.PREC_F16X3)
    x = nhwc.FMap(torch.randn(1, H, W, 256, device=dev), 256, 0)
    off = nhwc.FMap(torch.randn(1, H, W, 20, device=dev) * 2.0, 18, 0)
    outs = []
    for flag in (False, True):
        nhwc.DCN256[0] = flag
        ws = nhwc.Workspace(dev)
        st = torch.zeros(nhwc.GN_REP * 2 * 32, dtype=torch.float64, device=dev)
        o = pc(x, ws=ws, name='o', offset=off, gn=(st, 32))
        torch.cuda.synchronize()
        outs.append((o.t.clone(), st.view(nhwc.GN_REP, -1).sum(0).clone(), pc.gn_fused))
    print(H, W, 'bitwise equal', bool(torch.equal(outs[0][0], outs[1][0])), 'gn fused', outs[0][2], outs[1][2],
          'gn sums rel diff', float(((outs[0][1] - outs[1][1]).abs() / outs[0][1].abs().clamp_min(1e-9)).max()))
PY
for f in 0 1; do VPS_DCN256=$f BENCH_CONV_FILTER='dcn 256' timeout 120 python tools/bench_conv.py 4 2>&1 | tail -1; done
for i in 1 2; do for f in 0 1; do
VPS_DCN256=$f timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/c19_$f.json 2> gpurun_out/c19_$f.err
python - <<PY
import json
d = json.loads(open('gpurun_out/c19_$f.json').read().strip().splitlines()[-1])
print('dcn256=$f', d['value'], 'frames/s', 'conv_ms', d.get('extra', {}).get('conv_ms_single_stream'), d['roofline']['frac'])
PY
done; done
