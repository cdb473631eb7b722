#!/bin/bash
# run on the GPU box: kernel stats of the default bench + the two HBM-traffic PMC passes
cd /tmp && export TMPDIR=/tmp
R=/root/repo/gpurun_out
rm -rf $R/r1_stats $R/r1_fetch $R/r1_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/r1_stats -- python3 /root/repo/bench.py > $R/r1_stats.log 2>&1 || echo "stats pass failed"
rm -f $R/r1_stats/*/*kernel_trace.csv
rm -rf $R/r1_stats_serial
rocprofv3 --kernel-trace --stats --output-format csv -d $R/r1_stats_serial -- python3 /root/repo/bench.py --pipeline 1 > $R/r1_stats_serial.log 2>&1 || echo "serial stats pass failed"
rm -f $R/r1_stats_serial/*/*kernel_trace.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/r1_fetch -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu --no-render > $R/r1_fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/r1_write -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu --no-render > $R/r1_write.log 2>&1 || echo "write pass failed"
tail -1 $R/r1_stats.log | cut -c1-400
