#!/bin/bash
# Judged artifacts of a round, produced on the GPU box into gpurun_out/TAG_*:  usage: tools/profile_round2.sh TAG
#   1. kernel stats + per-step kernel histogram / queue report of the DEFAULT bench command (hipGraph replay)
#   2. MFMA-pipe counters (own pass, eager launches)        3. HBM traffic counters FETCH_SIZE / WRITE_SIZE (own passes, as the guide prescribes)
#   4. the bench line itself (with roofline + cpu_baseline), un-profiled
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
set -e
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-sampler --no-fp32-trunk-leg --steps 10 --warmup 3 > $R/gpurun_out/${tag}_stats.log 2>&1
python3 $R/tools/kernel_hist.py $R/gpurun_out/${tag}_stats/s_kernel_trace.csv 10 70 > $R/gpurun_out/${tag}_kernel_hist.txt
python3 $R/tools/queue_report.py $R/gpurun_out/${tag}_stats/s_kernel_trace.csv 10 > $R/gpurun_out/${tag}_queues.txt
python3 $R/tools/step_timeline.py $R/gpurun_out/${tag}_stats/s_kernel_trace.csv 3 > $R/gpurun_out/${tag}_timeline.txt
rm -f $R/gpurun_out/${tag}_stats/s_kernel_trace.csv
echo "stats pass done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $R/gpurun_out/${tag}_mfma -o m --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-sampler --no-graph --steps 2 --warmup 2 > $R/gpurun_out/${tag}_mfma.log 2>&1
python3 $R/tools/pmc_mfma_agg.py $R/gpurun_out/${tag}_mfma $R/gpurun_out/${tag}_pmc_mfma.txt > /dev/null
echo "mfma pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_fetch -o f --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-sampler --no-graph --steps 2 --warmup 2 > $R/gpurun_out/${tag}_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_write -o w --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-sampler --no-graph --steps 2 --warmup 2 > $R/gpurun_out/${tag}_write.log 2>&1
echo "write pass done"
python3 $R/tools/pmc_traffic.py $R/gpurun_out/${tag}_fetch $R/gpurun_out/${tag}_write $R/gpurun_out/${tag}_pmc_traffic.json
rm -rf $R/gpurun_out/${tag}_mfma/*/*kernel_trace.csv $R/gpurun_out/${tag}_fetch $R/gpurun_out/${tag}_write $R/gpurun_out/${tag}_mfma
cd $R
cp gpurun_out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
python3 bench.py --steps 30 --warmup 5 --dump-kernels gpurun_out/${tag}_conv_events.json > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -1 gpurun_out/${tag}_bench.json | cut -c1-600
# judged copies
cp gpurun_out/${tag}_stats/s_kernel_stats.csv profiles/${tag}_kernel_stats.csv 2>/dev/null || cp gpurun_out/${tag}_stats/*/s_kernel_stats.csv profiles/${tag}_kernel_stats.csv
for f in kernel_hist.txt queues.txt timeline.txt pmc_mfma.txt conv_events.json; do cp gpurun_out/${tag}_$f profiles/${tag}_$f; done
tail -1 gpurun_out/${tag}_bench.json > profiles/${tag}_bench.json
cp profiles/${tag}_*  gpurun_out/ 2>/dev/null || true
mkdir -p gpurun_out/profiles_${tag} && cp profiles/${tag}_* gpurun_out/profiles_${tag}/
