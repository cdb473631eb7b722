#!/bin/bash
# PMC passes over the encode-only script (full bench volume), per-kernel means for the large kernels.  usage: profiles/tools/pmc_enc.sh TAG
TAG=${1:-e}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES" "SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ENC_REPS=1 rocprofv3 --pmc $set --output-format csv -d /root/repo/gpurun_out/pmce_${TAG}_$i -- python3 /root/repo/profiles/tools/enc_time.py > /root/repo/gpurun_out/pmce_${TAG}_$i.log 2>&1 || echo "pass $i failed"
  tail -1 /root/repo/gpurun_out/pmce_${TAG}_$i.log
done
python3 - $TAG <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/root/repo/gpurun_out/pmce_%s_*/*/*counter_collection.csv' % tag):
    for r in csv.DictReader(open(f)):
        nm = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('vr::', '')
        if nm.startswith(('k_prune_emit12', 'k_pyramid12', 'k_index12')):
            agg[nm][r['Counter_Name']].append(float(r['Counter_Value']))
        elif nm.startswith(('k_est_summ', 'k_fill16')):
            agg[nm + ' (largest launch)'][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print('==', k)
    for c, v in sorted(d.items()):
        if 'largest' in k: print('  %-28s n=%d max %.4g' % (c, len(v), max(v)))
        else: print('  %-28s n=%d mean %.4g' % (c, len(v), sum(v) / len(v)))
PY
rm -rf /root/repo/gpurun_out/pmce_${TAG}_*
