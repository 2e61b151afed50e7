import csv, glob, os, sys
for d in sys.argv[1:]:
    per = {}
    for fn in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(fn)):
            if 'conv_mfma' not in row['Kernel_Name']:
                continue
            a = per.setdefault(row['Counter_Name'], [0, 0.0])
            a[0] += 1; a[1] += float(row['Counter_Value'])
    print(d)
    for k, (n, v) in sorted(per.items()):
        print('   %-36s n=%d  per dispatch %.4g' % (k, n, v / n))
