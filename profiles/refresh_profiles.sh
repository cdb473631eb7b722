#!/bin/bash
# Run on the GPU box (gpurun -- profiles/refresh_profiles.sh): kernel statistics of the default bench command and of the
# serial one, and the two HBM-traffic PMC passes (separate --pmc runs, no tracing combined with counters).
# profiles/make_profiles.py then turns what gets merged into gpurun_out/ into the committed r03_* files.
cd /tmp && export TMPDIR=/tmp
R=/root/repo/gpurun_out
rm -rf $R/r3_stats $R/r3_stats_serial $R/r3_fetch $R/r3_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/r3_stats -- python3 /root/repo/bench.py --no-cpu --no-stream > $R/r3_stats.log 2>&1 || echo "stats pass failed"
rm -f $R/r3_stats/*/*kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/r3_stats_serial -- python3 /root/repo/bench.py --pipeline 1 --level-loop-streams 1 --no-cpu --no-stream > $R/r3_stats_serial.log 2>&1 || echo "serial stats pass failed"
rm -f $R/r3_stats_serial/*/*kernel_trace.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/r3_fetch -- python3 /root/repo/bench.py --steps 1 --warmup 0 --level-loop-streams 1 --no-cpu --no-render --no-stream > $R/r3_fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/r3_write -- python3 /root/repo/bench.py --steps 1 --warmup 0 --level-loop-streams 1 --no-cpu --no-render --no-stream > $R/r3_write.log 2>&1 || echo "write pass failed"
tail -1 $R/r3_stats.log | cut -c1-600
