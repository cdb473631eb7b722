cd /tmp && export TMPDIR=/tmp
R=/root/repo/gpurun_out
rm -rf $R/r3_ps
rocprofv3 --kernel-trace --stats --output-format csv -d $R/r3_ps -- python3 /root/repo/bench.py --pipeline 1 --level-loop-streams 1 --no-cpu --no-stream --no-render --no-extra-timing > $R/r3_ps.log 2>&1 || echo failed
rm -f $R/r3_ps/*/*kernel_trace.csv
