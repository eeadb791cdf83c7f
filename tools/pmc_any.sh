#!/bin/bash
# dev tool: PMC counters for an arbitrary python tool.  usage (GPU box): tools/pmc_any.sh OUTDIR script.py args...
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
s=$1; shift
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $R/$out/p1 -o p1 --output-format csv -- python3 $R/$s "$@" > $R/$out/p1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d $R/$out/p2 -o p2 --output-format csv -- python3 $R/$s "$@" > $R/$out/p2.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d $R/$out/p0 -o p0 --output-format csv -- python3 $R/$s "$@" > $R/$out/p0.log 2>&1
