#!/bin/bash
# usage: profiles/tools/variants_pipe.sh "<hipcc flags>" ... : serial encode phases + pipelined step per build variant
for v in "$@"; do
  export VRHIP_EXTRA_HIPCC_FLAGS="$v"
  rm -f /root/repo/volumerenderer_amd/libvrhip.so
  echo "=== [$v]"
  python /root/repo/profiles/tools/enc_time.py 2>&1 | grep -v amdgpu.ids | tail -2
  python /root/repo/bench.py --no-cpu --no-render --no-stream 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipelined', d['value'], d['ms_per_step'], 'serial', d['serial_ms_per_step'])"
done
