#!/bin/bash
# Pass 1 of tools/profile_round2.sh alone (kernel stats + per-step histogram / queue report / timeline of the DEFAULT replayed step): tools/profile_stats_only.sh TAG
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
set -e
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-sampler --no-fp32-trunk-leg --steps 10 --warmup 3 > $R/gpurun_out/${tag}_stats.log 2>&1
python3 $R/tools/kernel_hist.py $R/gpurun_out/${tag}_stats/s_kernel_trace.csv 10 70 > $R/gpurun_out/${tag}_kernel_hist.txt
python3 $R/tools/queue_report.py $R/gpurun_out/${tag}_stats/s_kernel_trace.csv 10 > $R/gpurun_out/${tag}_queues.txt
python3 $R/tools/step_timeline.py $R/gpurun_out/${tag}_stats/s_kernel_trace.csv 3 > $R/gpurun_out/${tag}_timeline.txt
rm -f $R/gpurun_out/${tag}_stats/s_kernel_trace.csv
cp $R/gpurun_out/${tag}_stats/s_kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv 2>/dev/null || cp $R/gpurun_out/${tag}_stats/*/s_kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
