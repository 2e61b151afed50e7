"""Matrix-pipe busy fraction per conv kernel from a rocprofv3 --pmc pass over bench.py (GPU box).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -o pmc -- \
        python $R/bench.py --steps 1 --warmup 1 --single-stream --no-cpu-baseline --no-extras
    python $R/tools/pmc_mfma.py $R/gpurun_out/pmc_mfma > $R/gpurun_out/pmc_mfma.json

SQ_VALU_MFMA_BUSY_CYCLES counts busy cycles of the matrix pipes summed over the chip's SIMDs (MI355X_MICROARCH.md: 32 per
v_mfma_f32_32x32x16_{bf16,f16}; calibrated: the 256->256 3x3 @256x512 layer reports exactly 32 x its 14 155 776 MFMAs);
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (6.6 M counts for a 0.44 ms dispatch = 8 x 1.87 GHz), so the active cycles
of the dispatch are GUI_ACTIVE / 8. busy fraction = MFMA_BUSY / (GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs). Collected in its own
pass (no other trace domain)."""
import csv
import glob
import json
import os
import sys

NSIMD = 256 * 4
NXCD = 8


def main():
    d = sys.argv[1]
    per = {}
    for fn in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        rows = {}
        for row in csv.DictReader(open(fn)):
            key = (row.get('Dispatch_Id') or row.get('Dispatch_ID'), row['Kernel_Name'])
            rows.setdefault(key, {})[row['Counter_Name']] = float(row['Counter_Value'])
        for (_, name), c in rows.items():
            if 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or 'GRBM_GUI_ACTIVE' not in c:
                continue
            short = name.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0]
            a = per.setdefault(short, [0, 0.0, 0.0])
            a[0] += 1; a[1] += c['SQ_VALU_MFMA_BUSY_CYCLES']; a[2] += c['GRBM_GUI_ACTIVE'] / NXCD
    out = {'kernels': {}, 'note': 'busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / %d XCDs * %d SIMDs); conv family only' % (NXCD, NSIMD)}
    tb = ta = 0.0
    for k, (n, busy, act) in sorted(per.items(), key=lambda kv: -kv[1][2]):
        if 'conv_mfma' not in k and 'conv_thin' not in k and 'conv_pw' not in k:
            continue
        out['kernels'][k] = {'dispatches': n, 'mfma_busy_cycles': busy, 'gui_active_cycles': act, 'mfma_busy_frac': round(busy / max(act * NSIMD, 1.0), 4)}
        tb += busy; ta += act
    out['conv_family_mfma_busy_frac'] = round(tb / max(ta * NSIMD, 1.0), 4)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
