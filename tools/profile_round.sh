#!/bin/bash
# One gpurun call that refreshes every artifact under profiles/ (run from the repo root on the GPU box):
#   bash tools/profile_round.sh r01
# 1. bench.py (default workload) -> bench JSON + per-call breakdown          2. rocprofv3 --kernel-trace --stats of bench.py
# 3. two PMC passes (FETCH_SIZE, WRITE_SIZE) of the dominant call in isolation (tools/bench_conv.py, BENCH_LEGS=wg_cat)
# 4. per-workload bench lines + breakdowns (cfg1 cfg2 cfg3 cfg5) for the per-config table
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 10 --warmup 3 --breakdown $O/bench_breakdown.txt > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json; echo
rocprofv3 --kernel-trace --stats -d $O -o cfg4 --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rocprof_bench.log 2>&1
export BENCH_LEGS=wg_cat          # the two-source form bench.py's dominant call runs
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python $R/tools/bench_conv.py cfg4 bf16 decode5 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python $R/tools/bench_conv.py cfg4 bf16 decode5 > $O/pmc_write.log 2>&1
grep -h decode5 $O/pmc_fetch.log $O/pmc_write.log || true
python $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write k_wgrad_pipe 7 $O/pmc_traffic_cfg4.json "biu_conv_bwd_weight_cat @ decode5:bwd" > /dev/null
for W in cfg1 cfg2 cfg3 cfg5; do
  python $R/bench.py --workload $W --no-cpu-baseline --breakdown $O/bench_breakdown_$W.txt > $O/bench_$W.json 2> $O/bench_$W.err
  echo "$W done" >> $O/progress.txt
done
for W in cfg1 cfg2; do        # opt-in bf16x3 products of the fp32 workloads
  python $R/bench.py --workload $W --fp32-products bf16x3 --no-cpu-baseline --breakdown $O/bench_breakdown_${W}_bf16x3.txt > $O/bench_${W}_bf16x3.json 2> $O/bench_${W}_bf16x3.err
  echo "$W bf16x3 done" >> $O/progress.txt
done
ls $O
