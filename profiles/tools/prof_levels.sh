#!/bin/bash
# per-launch durations of the level-loop kernels of the last build, by kernel and launch order
cd /tmp && export TMPDIR=/tmp
R=/root/repo/gpurun_out
rm -rf $R/p_lv
rocprofv3 --kernel-trace --output-format csv -d $R/p_lv -- python3 /root/repo/profiles/tools/enc_time.py > $R/p_lv.log 2>&1 || echo "trace failed"
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/root/repo/gpurun_out/p_lv/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'vr::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# split into builds at k_pyramid12
builds = []
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('vr::', '')
    if n.startswith('k_pyramid12'): builds.append([])
    if builds: builds[-1].append((n, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Start_Timestamp']), int(r['End_Timestamp'])))
b = builds[-1]
tot = collections.defaultdict(float); cnt = collections.Counter()
for n, us, s, e in b: tot[n] += us; cnt[n] += 1
for n in sorted(tot, key=lambda k: -tot[k]): print("%-26s %4d launches %9.1f us" % (n[:26], cnt[n], tot[n]))
print("span %.1f us, sum %.1f us" % ((b[-1][3] - b[0][2]) / 1e3, sum(tot.values())))
for name in ('k_fill16', 'k_est_summ', 'k_est_walk', 'k_control', 'k_est_head', 'k_level_end'):
    xs = [us for n, us, s, e in b if n.startswith(name)]
    print(name, ' '.join('%.0f' % x for x in xs[-24:]))
PY
rm -rf $R/p_lv
