"""The measurement tooling under tools/ on synthetic rocprofv3 output (CPU): the numbers under profiles/ are only as good as these
parsers — kernel-name handling ("void (anonymous namespace)::name<...>(args)"), the XCD normalisation of GRBM_GUI_ACTIVE, the
gfx950 FETCH_SIZE correction, the join of the per-dispatch rows with the launch order of the instrumented frame."""
import csv
import importlib.util
import io
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = ['Correlation_Id', 'Dispatch_Id', 'Agent_Id', 'Queue_Id', 'Process_Id', 'Thread_Id', 'Grid_Size', 'Kernel_Id', 'Kernel_Name', 'Workgroup_Size',
       'LDS_Block_Size', 'Scratch_Size', 'VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'Counter_Name', 'Counter_Value', 'Start_Timestamp', 'End_Timestamp']
H8 = 'void (anonymous namespace)::conv_mfma_h8_kernel<4, 3, 3>(vps_conv_desc, int, int, int)'
BP = 'void (anonymous namespace)::conv_mfma_bf16p_kernel<2, 2, 2, 2, 4>(vps_conv_desc, int, int, int, int)'
RED = 'void (anonymous namespace)::conv_splitk_reduce_kernel<4>(vps_conv_desc, int)'
OTHER = 'void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float>, std::array<char*, 1ul> >(int, at::native::FillFunctor<float>, std::array<char*, 1ul>)'


def write_csv(path, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w', newline='') as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(HDR)
        for i, (disp, name, counter, value) in enumerate(rows):
            w.writerow([i + 1, disp, 'Agent 2', 1, 7, 7, 1024, 3, name, 256, 0, 0, 64, 0, 32, counter, float(value), 10 * i, 10 * i + 5])


def run(tool, *args):
    return subprocess.check_output([sys.executable, os.path.join(ROOT, 'tools', tool)] + [str(a) for a in args]).decode()


def test_pmc_mfma_busy_fraction(tmp_path):
    # one h8 dispatch: 1 000 000 active cycles per XCD (reported x8), the matrix pipes of 1024 SIMDs busy half of them
    rows = [(1, H8, 'SQ_VALU_MFMA_BUSY_CYCLES', 0.5 * 1e6 * 1024), (1, H8, 'GRBM_GUI_ACTIVE', 8e6),
            (2, BP, 'SQ_VALU_MFMA_BUSY_CYCLES', 0.25 * 2e6 * 1024), (2, BP, 'GRBM_GUI_ACTIVE', 16e6),
            (3, OTHER, 'SQ_VALU_MFMA_BUSY_CYCLES', 0.0), (3, OTHER, 'GRBM_GUI_ACTIVE', 8e6)]
    write_csv(str(tmp_path / 'm' / 'x' / 'pmc_counter_collection.csv'), rows)
    out = json.loads(run('pmc_mfma.py', tmp_path / 'm'))
    assert out['kernels']['conv_mfma_h8_kernel<4, 3, 3>']['mfma_busy_frac'] == 0.5
    assert out['kernels']['conv_mfma_bf16p_kernel<2, 2, 2, 2, 4>']['mfma_busy_frac'] == 0.25
    assert len(out['kernels']) == 2
    assert abs(out['conv_family_mfma_busy_frac'] - (0.5 * 1e6 + 0.25 * 2e6) / 3e6) < 1e-4


def test_pmc_traffic_and_per_layer_join(tmp_path):
    # two frames; the instrumented (last) frame launches h8, bf16p (+ its split-K reduce), h8. FETCH_SIZE / WRITE_SIZE are in KiB.
    seq = [H8, BP, RED, OTHER, H8]
    fetch, write = [], []
    d = 0
    for frame in range(2):
        for name in seq:
            d += 1
            fetch.append((d, name, 'FETCH_SIZE', {H8: 1000.0, BP: 500.0, RED: 50.0, OTHER: 7.0}[name]))
            write.append((d, name, 'WRITE_SIZE', {H8: 400.0, BP: 300.0, RED: 20.0, OTHER: 3.0}[name]))
    write_csv(str(tmp_path / 'f' / 'pmc_counter_collection.csv'), fetch)
    write_csv(str(tmp_path / 'w' / 'pmc_counter_collection.csv'), write)
    tr = json.loads(run('pmc_traffic.py', tmp_path / 'f', tmp_path / 'w', 2))
    k = tr['kernels']
    assert k['conv_mfma_h8_kernel']['launches_per_frame'] == 2.0
    assert k['conv_mfma_h8_kernel']['fetch_bytes_per_frame'] == 2 * 2 * 1000.0 * 1024          # x2: the gfx950 wide-read correction
    assert k['conv_mfma_h8_kernel']['write_bytes_per_frame'] == 2 * 400.0 * 1024
    assert k['conv_splitk_reduce_kernel']['fetch_bytes_per_frame'] == 2 * 50.0 * 1024
    assert tr['conv_hbm_bytes_per_frame'] == sum(v['fetch_bytes_per_frame'] + v['write_bytes_per_frame'] for v in k.values())
    order = [dict(layer='A 3x3', flops=1, ms=0.1, algorithmic_bytes=1.0e6), dict(layer='B 1x1 ksplit4', flops=1, ms=0.2, algorithmic_bytes=0.5e6),
             dict(layer='A 3x3', flops=1, ms=0.1, algorithmic_bytes=1.0e6)]
    oj = tmp_path / 'order.json'
    oj.write_text(json.dumps(order))
    txt = run('pmc_per_layer.py', oj, tmp_path / 'f', tmp_path / 'w')
    lines = {l.split('  ')[0].strip(): l for l in txt.splitlines()}
    a = [float(v) for v in lines['A 3x3'].split()[3:]]                    # kernel, calls, ms, algo, fetch, write, reduce, ratio, excess (MB)
    assert lines['A 3x3'].split()[2] == 'mfma_h8' and a[0] == 2
    assert abs(a[2] - 2.0) < 1e-6 and abs(a[3] - 2 * 2 * 1000 * 1024 / 1e6) < 0.06 and abs(a[4] - 2 * 400 * 1024 / 1e6) < 0.06 and a[5] == 0.0
    b = [float(v) for v in lines['B 1x1 ksplit4'].split()[4:]]
    assert abs(b[5] - (2 * 50 + 20) * 1024 / 1e6) < 0.06                  # the reduce kernel's traffic is charged to the conv in front of it
    assert 'total: algorithmic 0.00 GB' in txt


def test_check_isa_reports_resources_and_mix():
    spec = importlib.util.spec_from_file_location('check_isa', os.path.join(ROOT, 'tools', 'check_isa.py'))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    if not (os.path.exists(mod.OBJDUMP) and os.path.exists(mod.READELF)):
        import pytest
        pytest.skip('llvm tools not available')
    lib = os.path.join(ROOT, 'vps_amd', 'csrc', 'libvpship.so')
    mix = mod.instruction_mix(lib)
    h8 = mix['void conv_mfma_h8_kernel<4, 3, 3>']
    assert h8['mfma'] == 216 and h8['barrier'] == 9                       # 9 taps x (2 slabs x 3 products x 2 x 2 fragments), one barrier per tap
    assert h8['valu'] / h8['mfma'] < 5 and h8['lds_read'] / h8['mfma'] < 2        # the issue budget of DESIGN.md 3.1
    assert mix['void conv_mfma_h8s2_kernel<4, 5, 128>']['mfma'] == 600    # 25 taps of the phase-split 5x5 stride-2 kernel
    assert mix['void conv_mfma_h8s2_kernel<4, 3, 64>']['mfma'] == 108     # the 64-column instance: one column fragment per wave
    n16 = mix['void conv_mfma_n16_kernel<4, 3, 3>']
    assert n16['mfma'] == 54 and n16['barrier'] == 2                      # 9 taps x 3 products x 2 pixel groups of 16x16x32, two barriers per chunk


def test_bench_multi_gpu_launch_fails_with_one_json_line_not_an_assert():
    """`python bench.py --gpus N` without a launcher re-launches itself under torch.distributed.run; on a box with
    fewer than N GPUs (this container has none) it prints ONE JSON line with an `error` field and exits 2 - no assert, no traceback.
    The same for a launcher whose WORLD_SIZE disagrees with --gpus, and for N = 1 without a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    for args, extra in ((['--gpus', '8'], {}), (['--gpus', '2'], {'WORLD_SIZE': '1'}), ([], {})):
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=dict(env, **extra), capture_output=True, text=True, timeout=300)
        import torch
        if torch.cuda.is_available() and not args:
            continue                     # on a GPU box the N = 1 run would be the real benchmark
        lines = [l for l in p.stdout.strip().splitlines() if l.strip()]
        assert p.returncode == 2 and len(lines) == 1, (args, p.returncode, p.stdout, p.stderr[-500:])
        j = json.loads(lines[0])
        assert 'error' in j and j['n_gpus'] == (int(args[1]) if args else 1)
        assert 'Traceback' not in p.stderr
