import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/root/repo/gpurun_out/pmc_%s_*/*/*counter_collection.csv' % tag):
    for r in csv.DictReader(open(f)):
        nm = r['Kernel_Name']
        for key in ('k_decode_region', 'k_decode_quad', 'k_decode_fine'):
            if key in nm:
                agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print('==', k)
    for c, v in sorted(d.items()):
        print('  %-28s n=%d mean %.4g' % (c, len(v), sum(v) / len(v)))
    if 'SQ_WAVE_CYCLES' in d:
        m = {c: sum(v) / len(v) for c, v in d.items()}
        wc = m['SQ_WAVE_CYCLES']
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_VMEM', 'SQ_WAIT_INST_LDS'):
            if c in m: print('  %-22s / WAVE_CYCLES = %.3f' % (c, m[c] / wc))
        if 'SQ_BUSY_CYCLES' in m and 'SQ_ACTIVE_INST_VALU' in m:
            # BUSY_CYCLES is per SE (32): cycles the SQ had waves; SIMD-cycles = 4 * busy-cycles-per-CU ...
            print('  VALU busy (ACTIVE_INST_VALU*4 / (BUSY_CYCLES/32*1024 simd)) ~ %.3f' % (m['SQ_ACTIVE_INST_VALU'] * 4 / (m['SQ_BUSY_CYCLES'] / 32 * 1024)))
        if 'SQ_LDS_IDX_ACTIVE' in m: print('  LDS bank conflict share %.3f' % (m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']))
