"""Per-layer roofline of the conv family from artefacts already in profiles/ (no GPU): joins the per-layer timing table of the
bench (`--conv-table`: calls, ms, GFLOP) with the per-layer algorithmic / measured bytes of the PMC passes
(`tools/pmc_per_layer.py`) and prints, per layer shape,

    mfma floor   = 3 x FLOP / 2.5 PFLOP/s   (f16x3: three fp16 MFMAs per fp32 product; p0 = exact-fp32 vector kernel: FLOP / 78 TFLOP/s)
    hbm floor    = algorithmic bytes / 8 TB/s
    floor        = max of the two (which roof bounds the layer), ms / floor = how far the layer is from its own roofline

and the totals: the speed of light of the frame's conv work under this arithmetic (sum of the floors) against the measured sum.

    python tools/roofline_per_layer.py profiles/r03_conv_table_f16x3.txt profiles/r03_traffic_per_layer_f16x3.txt > profiles/r03_conv_roofline_per_layer.txt
"""
import re
import sys

MFMA_PEAK, VALU_F32_PEAK, HBM_PEAK = 2.5e15, 78.6e12, 8.0e12


def rows(fn, ncol):
    out = {}
    for line in open(fn).read().splitlines()[1:]:
        p = line.split()
        if len(p) <= ncol:
            continue
        out[' '.join(p[:-ncol])] = p[-ncol:]
    return out


def main():
    table, traffic = rows(sys.argv[1], 4), rows(sys.argv[2], 9)
    algo = {}
    for k, v in traffic.items():              # kernel calls ms algoMB fetchMB writeMB reduceMB ratio excessMB
        algo[k] = (float(v[3]), float(v[4]) + float(v[5]) + float(v[6]), v[0])
    out, tot = [], dict(ms=0.0, floor=0.0, mfma=0.0, hbm=0.0, nomatch=0.0)
    for k, (calls, ms, gflop, tf) in table.items():
        ms, gflop = float(ms), float(gflop)
        exact = re.search(r' p0$', k) is not None
        mfma = gflop * 1e9 / VALU_F32_PEAK * 1e3 if exact else 3 * gflop * 1e9 / MFMA_PEAK * 1e3
        a = algo.get(k)
        hbm = a[0] * 1e6 / HBM_PEAK * 1e3 if a else 0.0
        floor = max(mfma, hbm)
        tot['ms'] += ms; tot['floor'] += floor; tot['mfma'] += mfma; tot['hbm'] += hbm
        if not a:
            tot['nomatch'] += ms
        out.append((ms - floor, k, int(calls), ms, gflop, mfma, hbm, 'mfma' if mfma >= hbm else 'hbm', ms / floor if floor > 0 else 0.0,
                    (a[1] / a[0]) if a and a[0] else 0.0, a[2] if a else '-'))
    out.sort(reverse=True)
    print('conv family per layer shape against its own roofline (f16x3: 3 MFMAs per product at 2.5 PFLOP/s dense fp16; HBM 8 TB/s). Sorted by ms above the floor.')
    print('%-58s %5s %8s %9s %9s %9s %5s %8s %8s  %s' % ('layer shape', 'calls', 'ms', 'GFLOP', 'mfma ms', 'hbm ms', 'bound', 'ms/floor', 'traffic/', 'kernel'))
    for d, k, calls, ms, gflop, mfma, hbm, b, r, tr, kern in out:
        print('%-58s %5d %8.3f %9.2f %9.3f %9.3f %5s %8.1f %8.2f  %s' % (k, calls, ms, gflop, mfma, hbm, b, r, tr, kern))
    print()
    print('sum of the layers: measured %.2f ms, sum of floors %.2f ms (x%.1f); all-MFMA floor %.2f ms, all-HBM floor %.2f ms; layers without a traffic row: %.2f ms'
          % (tot['ms'], tot['floor'], tot['ms'] / tot['floor'], tot['mfma'], tot['hbm'], tot['nomatch']))


if __name__ == '__main__':
    main()
