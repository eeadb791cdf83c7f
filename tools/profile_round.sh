#!/bin/bash
# Produce the judged artifacts of a round on the GPU box: kernel stats of the default bench command, HBM traffic counters (own
# passes, as MI355X_MICROARCH.md prescribes), then the bench line itself.  usage: tools/profile_round.sh TAG   (writes gpurun_out/TAG_*)
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
set -e
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > $R/gpurun_out/${tag}_stats.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_fetch -o f --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-graph --steps 3 --warmup 2 > $R/gpurun_out/${tag}_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_write -o w --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-graph --steps 3 --warmup 2 > $R/gpurun_out/${tag}_write.log 2>&1
echo "write pass done"
python3 $R/tools/pmc_traffic.py $R/gpurun_out/${tag}_fetch $R/gpurun_out/${tag}_write $R/gpurun_out/${tag}_pmc_traffic.json
