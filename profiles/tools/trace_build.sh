cd /tmp && export TMPDIR=/tmp
R=/root/repo/gpurun_out
rm -rf $R/r3_tr
rocprofv3 --kernel-trace --output-format csv -d $R/r3_tr -- python3 /root/repo/profiles/tools/enc_time.py > $R/r3_tr.log 2>&1 || echo failed
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/root/repo/gpurun_out/r3_tr/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last build: find last k_pyramid12
idx = [i for i, r in enumerate(rows) if 'k_pyramid12' in r['Kernel_Name']]
seg = rows[idx[-1]:]
out = []
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    nm = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('vr::', '')
    out.append("%9.1f %8.1f %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, nm))
open('/root/repo/gpurun_out/r3_tr_lastbuild.txt', 'w').write("\n".join(out))
PY
rm -rf $R/r3_tr
