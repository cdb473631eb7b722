#!/bin/bash
# HBM traffic of the decode kernel (two separate PMC passes).  usage: profiles/tools/pmc_traffic.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/pt_$c
  rocprofv3 --pmc $c --output-format csv -d $R/pt_$c -- python3 /root/repo/profiles/tools/dec_time.py > $R/pt_$c.log 2>&1 || echo "pass $c failed"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob('/root/repo/gpurun_out/pt_%s/*/*counter_collection.csv' % c):
        for r in csv.DictReader(open(f)):
            if 'k_decode_quad' in r['Kernel_Name'] and r['Counter_Name'] == c:
                acc[c].append(float(r['Counter_Value']))
fe = sum(acc['FETCH_SIZE']) / len(acc['FETCH_SIZE']) * 1024
wr = sum(acc['WRITE_SIZE']) / len(acc['WRITE_SIZE']) * 1024
print("k_decode_quad per launch: FETCH_SIZE %.3f GB (x2 = %.3f) WRITE_SIZE %.3f GB -> traffic %.3f GB" % (fe / 1e9, 2 * fe / 1e9, wr / 1e9, (2 * fe + wr) / 1e9))
PY
rm -rf $R/pt_FETCH_SIZE $R/pt_WRITE_SIZE
