#!/bin/bash
cd /root/repo
for f in "" "-DEST_LANE_BOUNDS" "" "-DEST_LANE_BOUNDS"; do
  echo "=== variant: $f"
  VRHIP_EXTRA_HIPCC_FLAGS="$f" timeout -k 10 400 python bench.py --no-cpu --no-render --no-stream --no-extra-timing --steps 10 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j=json.loads(line); print(j['value'], j['ms_per_step'], j['serial_ms_per_step'], j['phases_ms'])
"
done
