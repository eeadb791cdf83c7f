#!/bin/bash
# dev tool: FETCH_SIZE / WRITE_SIZE of conv_bench kernels.  usage: tools/pmc_fetch.sh OUTDIR <conv_bench args>
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$out/p1 -o p1 --output-format csv -- python3 $R/tools/conv_bench.py "$@" > $R/$out/p1.log 2>&1
