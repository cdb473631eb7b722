#!/bin/bash
# usage: [DEC_SCRIPT=profiles/tools/dec_split.py] profiles/tools/variants.sh "flagsA" "flagsB" ...   (each variant rebuilds libvrhip.so and times the decode)
cd /root/repo
for f in "$@"; do
  echo "=== variant: $f"
  VRHIP_EXTRA_HIPCC_FLAGS="$f" TAG="$f" QUICK=1 timeout -k 10 400 python ${DEC_SCRIPT:-profiles/tools/dec_time.py} 2>&1 | grep -v "amdgpu.ids\|warning\|^ *[0-9]* |\|^ *|\|generated"
done
