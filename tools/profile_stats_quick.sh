#!/bin/bash
# kernel histogram of the default bench command (hipGraph replay) only: tools/profile_stats_quick.sh TAG
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
set -e
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-sampler --no-fp32-trunk-leg --steps 10 --warmup 3 > $R/gpurun_out/${tag}_stats.log 2>&1
python3 $R/tools/kernel_hist.py $R/gpurun_out/${tag}_stats/s_kernel_trace.csv 10 90 > $R/gpurun_out/${tag}_kernel_hist.txt
rm -f $R/gpurun_out/${tag}_stats/s_kernel_trace.csv
