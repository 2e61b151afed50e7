"""Per-layer HBM-side traffic of the conv launches of ONE frame: joins the launch order of bench.py's instrumented frame
(`--conv-table X` writes X.ordered.json) with the per-dispatch rows of the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
of `bench.py --steps 1 --warmup 1 --single-stream` — the instrumented frame is the last frame of that command, one stream, so
its conv dispatches are the last len(order) conv dispatches of the trace, in order; a split-K reduce kernel is charged to the conv
launch in front of it. Units and the gfx950 FETCH_SIZE correction as in tools/pmc_traffic.py.

    python tools/pmc_per_layer.py gpurun_out/conv_table.txt.ordered.json gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_traffic_per_layer.txt
"""
import csv
import glob
import json
import os
import re
import sys

CONV = re.compile(r'conv_mfma_\w+_kernel|conv_thin_kernel|conv_small\w*_kernel|conv_pw_kernel')


def dispatches(directory, counter):
    rows = []
    for fn in glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(fn)):
            if row['Counter_Name'] == counter and ('conv_' in row['Kernel_Name']):
                rows.append((int(row['Dispatch_Id']), row['Kernel_Name'], float(row['Counter_Value'])))
    rows.sort()
    out = []   # [kernel, bytes of the conv launch, bytes of its reduce]
    for _, name, v in rows:
        if 'conv_splitk_reduce_kernel' in name:
            out[-1][2] += v
        elif CONV.search(name):
            out.append([CONV.search(name).group(0), v, 0.0])
    return out


def main():
    order = json.load(open(sys.argv[1]))
    fe = dispatches(sys.argv[2], 'FETCH_SIZE')[-len(order):]
    wr = dispatches(sys.argv[3], 'WRITE_SIZE')[-len(order):]
    assert len(fe) == len(order) == len(wr), (len(fe), len(wr), len(order))
    agg = {}
    for o, f, w in zip(order, fe, wr):
        assert f[0] == w[0], (f[0], w[0])
        a = agg.setdefault((o['layer'], f[0].replace('conv_', '').replace('_kernel', '')), [0, 0.0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += o['algorithmic_bytes']; a[2] += 2.0 * 1024 * f[1]; a[3] += 1024 * w[1]; a[4] += 2.0 * 1024 * f[2] + 1024 * w[2]; a[5] += o['ms']
    print('%-58s %-10s %5s %8s %9s %9s %9s %9s %7s %8s' % ('layer shape', 'kernel', 'calls', 'ms', 'algo MB', 'fetch MB', 'write MB', 'reduce MB', 'ratio', 'excess MB'))
    tot = [0.0, 0.0]
    for (layer, kern), a in sorted(agg.items(), key=lambda kv: -(kv[1][2] + kv[1][3] + kv[1][4] - kv[1][1])):
        t = a[2] + a[3] + a[4]
        tot[0] += a[1]; tot[1] += t
        print('%-58s %-10s %5d %8.3f %9.1f %9.1f %9.1f %9.1f %7.2f %9.1f' % (layer, kern, a[0], a[5], a[1] / 1e6, a[2] / 1e6, a[3] / 1e6, a[4] / 1e6, t / max(a[1], 1), (t - a[1]) / 1e6))
    print('total: algorithmic %.2f GB, counted %.2f GB, ratio %.3f' % (tot[0] / 1e9, tot[1] / 1e9, tot[1] / tot[0]))


if __name__ == '__main__':
    main()
