tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-sampler --steps 10 --warmup 3 > $R/gpurun_out/${tag}_stats.log 2>&1
python3 $R/tools/kernel_hist.py $R/gpurun_out/${tag}_stats/s_kernel_trace.csv 10 90 > $R/gpurun_out/${tag}_kernel_hist.txt
python3 $R/tools/queue_report.py $R/gpurun_out/${tag}_stats/s_kernel_trace.csv 10 > $R/gpurun_out/${tag}_queues.txt
python3 $R/tools/step_timeline.py $R/gpurun_out/${tag}_stats/s_kernel_trace.csv 3 > $R/gpurun_out/${tag}_timeline.txt
rm -f $R/gpurun_out/${tag}_stats/s_kernel_trace.csv
cat $R/gpurun_out/${tag}_queues.txt
