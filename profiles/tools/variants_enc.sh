#!/bin/bash
# usage: profiles/tools/variants_enc.sh "flagsA" "flagsB" ...   (each variant rebuilds libvrhip.so and times the build phases)
cd /root/repo
for f in "$@"; do
  echo "=== variant: $f"
  VRHIP_EXTRA_HIPCC_FLAGS="$f" TAG="$f" timeout -k 10 400 python profiles/tools/enc_time.py 2>&1 | grep -v "amdgpu.ids\|warning\|^ *[0-9]* |\|^ *|\|generated"
done
