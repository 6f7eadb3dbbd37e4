#!/bin/bash
# FETCH_SIZE of one layer's kernels under an experiment switch:  bash tools/pmc_fetch.sh <tag> <layer> [ENV=VAL ...]
set -e
TAG=$1; LAYER=$2; shift 2
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/fetch_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export BENCH_LEGS=${BENCH_LEGS:-fwd_st,dgrad,wg_bn}
python3 $R/tools/bench_conv.py cfg4 bf16 $LAYER > $O/time.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/rd -o p --output-format csv -- python3 $R/tools/bench_conv.py cfg4 bf16 $LAYER > $O/rd.log 2>&1
python3 - <<PY >> $O/time.log
import csv, collections
tot=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open("$O/rd/p_counter_collection.csv")):
    k=r["Kernel_Name"]
    if "k_conv_pipe" in k or "k_wgrad_pipe" in k:
        key=("wgrad" if "wgrad" in k else "conv")+" grid "+r.get("Grid_Size","?")
        tot[key]+=float(r["Counter_Value"])*1024*2; n[key]+=1
for k,v in tot.items(): print("FETCH(x2)", k, n[k], "dispatches", round(v/n[k]/1e9,3), "GB each")
PY
cat $O/time.log
