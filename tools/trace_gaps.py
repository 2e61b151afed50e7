"""How busy is the GPU inside one frame of the two-stream clip pipeline? Reads a `rocprofv3 --kernel-trace` CSV of

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras

and, for every frame interval (end of one `panoptic_combine_kernel` to the end of the next), reports
  period      the interval, ms
  busy        the union of all kernel [start, end) intervals inside it (any stream), ms  -> idle = period - busy
  work        the SUM of kernel durations inside it, ms                                 -> work / busy = mean concurrency
  gaps        idle gaps by size class: count and total ms
Only intervals whose period is within 15 % of the median are kept (the timed frames; warm-up, the instrumented single-stream
frame and the per-frame-call loop have other periods). Prints one JSON line.

    python tools/trace_gaps.py gpurun_out/trace [--out profiles/rNN_frame_occupancy.json]
"""
import argparse
import csv
import glob
import json
import os
import sys

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('--out', default=None)
    ap.add_argument('--marker', default='panoptic_combine_kernel')
    args = ap.parse_args()
    files = glob.glob(os.path.join(args.dir, '**', '*kernel_trace.csv'), recursive=True)
    assert files, 'no *kernel_trace.csv under %s' % args.dir
    rows = []
    for fn in files:
        with open(fn) as f:
            for r in csv.DictReader(f):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    short = lambda n: n.split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')[:60]
    st = np.array([r[0] for r in rows], np.int64); en = np.array([r[1] for r in rows], np.int64)
    marks = np.array([r[1] for r in rows if args.marker in r[2]], np.int64)
    assert marks.size > 4, 'marker kernel %s not found' % args.marker
    per = np.diff(marks)
    med = float(np.median(per))
    keep = [i for i in range(per.size) if abs(per[i] - med) <= 0.15 * med]
    res = []
    classes = [(0, 2e3), (2e3, 10e3), (10e3, 50e3), (50e3, 1e12)]
    for i in keep:
        a, b = marks[i], marks[i + 1]
        sel = (en > a) & (st < b)
        s = np.clip(st[sel], a, b); e = np.clip(en[sel], a, b)
        order = np.argsort(s)
        s, e = s[order], e[order]
        names = [rows[k][2] for k in np.nonzero(sel)[0][order]]
        busy, gaps, cur, big, last = 0, [], a, [], 'frame start'
        for x, y, nm in zip(s, e, names):
            if x > cur:
                gaps.append(x - cur)
                if x - cur >= 20e3:
                    big.append((float(x - cur) / 1e3, round(float(cur - a) / 1e6, 3), short(last), short(nm)))
                busy += y - x
                cur = y; last = nm
            elif y > cur:
                busy += y - cur
                cur = y; last = nm
        if b > cur:
            gaps.append(b - cur)
        g = np.array(gaps, np.float64)
        res.append(dict(big=sorted(big, reverse=True)[:8], period=(b - a) / 1e6, busy=busy / 1e6, work=float((e - s).sum()) / 1e6, launches=int(sel.sum()),
                        gaps=[[int(((g >= lo) & (g < hi)).sum()), float(g[(g >= lo) & (g < hi)].sum()) / 1e6] for lo, hi in classes]))
    m = lambda k: round(float(np.median([r[k] for r in res])), 3)
    out = dict(frames=len(res), period_ms=m('period'), busy_ms=m('busy'), idle_ms=round(m('period') - m('busy'), 3), work_ms=m('work'),
               mean_concurrency=round(m('work') / m('busy'), 3), launches_per_frame=int(np.median([r['launches'] for r in res])),
               idle_gaps={'<2us': None, '2-10us': None, '10-50us': None, '>50us': None},
               note='median over the kept frame intervals; gaps: [count, total ms] per frame (median)')
    for j, k in enumerate(out['idle_gaps']):
        out['idle_gaps'][k] = [int(np.median([r['gaps'][j][0] for r in res])), round(float(np.median([r['gaps'][j][1] for r in res])), 3)]
    # where the idle time sits: the largest gaps of the median-period frame, [us, ms after the frame start, kernel that ended before it,
    # kernel that started after it]
    mid = sorted(res, key=lambda r: r['period'])[len(res) // 2]
    out['largest_gaps_of_a_median_frame'] = [[round(g[0], 1), g[1], g[2], g[3]] for g in mid['big']]
    line = json.dumps(out)
    print(line)
    if args.out:
        with open(args.out, 'w') as f:
            f.write(line + '\n')


if __name__ == '__main__':
    sys.exit(main())
