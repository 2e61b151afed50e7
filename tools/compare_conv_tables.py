"""Two `bench.py --conv-table` files side by side (A/B of a kernel change on the layers of the real frame; no GPU).

    python tools/compare_conv_tables.py before.txt after.txt [min_ms]
"""
import sys


def rows(fn):
    out = {}
    for line in open(fn).read().splitlines()[1:]:
        p = line.split()
        if len(p) > 4:
            out[' '.join(p[:-4])] = (int(p[-4]), float(p[-3]), float(p[-2]))
    return out


def main():
    a, b = rows(sys.argv[1]), rows(sys.argv[2])
    min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    print('%-58s %5s %9s %9s %8s' % ('layer shape', 'calls', 'before ms', 'after ms', 'ratio'))
    ta = tb = 0.0
    for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0, 0))[1])):
        x, y = a.get(k), b.get(k)
        ta += x[1] if x else 0.0; tb += y[1] if y else 0.0
        if x and y and x[1] >= min_ms:
            print('%-58s %5d %9.3f %9.3f %8.2f' % (k, x[0], x[1], y[1], y[1] / x[1] if x[1] else 0.0))
        elif (x or y) and (x or y)[1] >= min_ms:
            print('%-58s %5d %9s %9s' % (k, (x or y)[0], '%.3f' % x[1] if x else '-', '%.3f' % y[1] if y else '-'))
    print('total %.3f -> %.3f ms' % (ta, tb))


if __name__ == '__main__':
    main()
