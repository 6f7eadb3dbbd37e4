#!/bin/bash
# The round's rocprofv3 evidence, run from the repo root on the GPU box:   bash tools/profile_round.sh r04
#   1. --kernel-trace --stats of the default bench command (cfg4)                       -> gpurun_out/prof_<tag>/cfg4_*.csv
#   2. FETCH_SIZE and WRITE_SIZE (separate --pmc passes, --kernel-trace only) of the dominant call, isolated through tools/bench_conv.py
#      (BENCH_LEGS picks the call)                                                       -> gpurun_out/prof_<tag>/pmc_{fetch,write}_<leg>/
# tools/summarize_profile.py and tools/pmc_traffic.py turn them into profiles/<tag>_cfg4_summary.md and profiles/pmc_traffic_cfg4.json.
set -e
TAG=${1:-r04}
LEG=${2:-wg1,fwd,dg}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o cfg4 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_profiler.json 2> $O/bench_under_profiler.err
echo "trace done" >> $O/progress.txt
# one pair of counter passes per leg of the folded decode5 op (tools/bench_foldt.py: wg1 = biu_foldt_bwd_weight_bn_phase(1 | 4), the main-stream part of the weight gradient; fwd; dg): bench.py's dominant
# call is one of them (before the fold: tools/bench_conv.py cfg4 bf16 decode5 with BENCH_LEGS=dg_cat | wg_cat | fwd_cat)
export BENCH_REPS=20
for L in ${LEG//,/ }; do
    export BENCH_LEGS=$L
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$L -o p --output-format csv -- python3 $R/tools/bench_foldt.py bf16 decode5 > $O/pmc_fetch_$L.log 2>&1
    echo "fetch $L done" >> $O/progress.txt
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$L -o p --output-format csv -- python3 $R/tools/bench_foldt.py bf16 decode5 > $O/pmc_write_$L.log 2>&1
    echo "write $L done" >> $O/progress.txt
done
find $O -name "*.csv" | head -20
