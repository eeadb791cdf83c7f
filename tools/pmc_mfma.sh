#!/bin/bash
# dev tool: MFMA-pipe utilisation counters for conv_bench.  usage: tools/pmc_mfma.sh OUTDIR <conv_bench args>
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d $R/$out/p1 -o p1 --output-format csv -- python3 $R/tools/conv_bench.py "$@" > $R/$out/p1.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$out/p0 -o p0 --output-format csv -- python3 $R/tools/conv_bench.py "$@" > $R/$out/p0.log 2>&1
