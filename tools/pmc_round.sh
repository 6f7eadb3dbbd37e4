#!/bin/bash
# MFMA-busy / wait / HBM-traffic counters of the step's dominant kernels at HEAD, one rocprofv3 pass per counter group (PMC passes
# never combined with trace domains other than --kernel-trace).  Run from the repo root on the GPU box:
#     bash tools/pmc_round.sh r02
# Writes gpurun_out/pmc_<tag>/{sq,lds,tcc_rd,tcc_wr}/...csv; tools/pmc_summary.py turns them into profiles/<tag>_cfg4_mfma_busy.md.
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export BENCH_LEGS=fwd_st,dgrad,wg_bn
for LAYER in decode5 decode3 decode6 encode2; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
      -d $O/sq_$LAYER -o p --output-format csv -- python3 $R/tools/bench_conv.py cfg4 bf16 $LAYER > $O/sq_$LAYER.log 2>&1
  echo "sq $LAYER done" >> $O/progress.txt
done
for LAYER in decode5 decode6; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/rd_$LAYER -o p --output-format csv -- python3 $R/tools/bench_conv.py cfg4 bf16 $LAYER > $O/rd_$LAYER.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr_$LAYER -o p --output-format csv -- python3 $R/tools/bench_conv.py cfg4 bf16 $LAYER > $O/wr_$LAYER.log 2>&1
  echo "traffic $LAYER done" >> $O/progress.txt
done
ls $O
