import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/root/repo/gpurun_out/pmc_%s_*/*/*counter_collection.csv' % tag):
    for r in csv.DictReader(open(f)):
        nm = r['Kernel_Name']
        if 'k_decode_quad' in nm or 'k_decode_fine' in nm:
            k = 'quad' if 'quad' in nm else 'fine'
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print('==', k)
    for c, v in sorted(d.items()):
        print('  %-28s n=%d mean %.4g' % (c, len(v), sum(v) / len(v)))
