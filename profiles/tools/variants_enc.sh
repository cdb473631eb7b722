#!/bin/bash
# usage: profiles/tools/variants_enc.sh "<hipcc flags>" ... : encode timing per build variant
for v in "$@"; do
  export VRHIP_EXTRA_HIPCC_FLAGS="$v"
  rm -f /root/repo/volumerenderer_amd/libvrhip.so
  TAG="[$v]" python /root/repo/profiles/tools/enc_time.py 2>&1 | grep -v amdgpu.ids | tail -2
done
