R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/chain -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > $R/gpurun_out/chain.log 2>&1
f=$(ls $R/gpurun_out/chain/*/s_kernel_trace.csv 2>/dev/null || ls $R/gpurun_out/chain/s_kernel_trace.csv)
python3 $R/tools/queue_report.py $f 10
for q in 2 3 4 5; do python3 $R/tools/stream_chain.py $f $q 10 > $R/gpurun_out/chain_q$q.txt; done
python3 $R/tools/kernel_hist.py $f 10 70 > $R/gpurun_out/chain_hist.txt
rm -rf $R/gpurun_out/chain
