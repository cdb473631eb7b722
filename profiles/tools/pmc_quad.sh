#!/bin/bash
# PMC passes over the decode-only script (full bench volume).  usage: profiles/tools/pmc_quad.sh TAG
TAG=${1:-q}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" "SQ_WAVES SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /root/repo/gpurun_out/pmc_${TAG}_$i -- python3 /root/repo/profiles/tools/dec_time.py > /root/repo/gpurun_out/pmc_${TAG}_$i.log 2>&1 || echo "pass $i failed"
  tail -2 /root/repo/gpurun_out/pmc_${TAG}_$i.log
done
python3 /root/repo/profiles/tools/pmc_parse.py $TAG
