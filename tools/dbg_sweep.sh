#!/bin/bash
# dev tool: kernel-only duration of a kernel family under the HDMOE_DBG ablation knobs.  usage: KERN=name ARGS="..." tools/dbg_sweep.sh d1 d2 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in "$@"; do
  export HDMOE_DBG=$d
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dbg/d$d -o p --output-format csv -- python3 $R/tools/conv_bench.py $ARGS iters=10 > /dev/null 2>&1
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$R/gpurun_out/dbg/d$d/p_kernel_stats.csv")) if "$KERN" in r["Name"]]
for r in rows: print("DBG=$d", r["Name"][:70], "calls", r["Calls"], "avg_us %.1f min_us %.1f" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
