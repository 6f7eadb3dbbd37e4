# kernel-level times of biu_foldt_pack + biu_foldt_bwd_weight_bn at one level:   bash tools/profile_fold_weights.sh decode1
L=${1:-decode1}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
BENCH_LEGS=wg BENCH_REPS=20 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fold_$L -o p --output-format csv -- python3 $R/tools/bench_foldt.py bf16 $L > /dev/null 2>&1
cd $R
python3 - $(find gpurun_out/prof_fold_$L -name "*kernel_stats.csv" | head -1) <<PY
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:20]:
    print(r["Name"][:72].ljust(72), r["Calls"].rjust(4), round(float(r["AverageNs"])/1e3,1), "us")
PY
