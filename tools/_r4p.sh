mkdir -p gpurun_out/r4p
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -k "foldt" > gpurun_out/r4p/ops.log 2>&1; tail -1 gpurun_out/r4p/ops.log
for i in 1 2 3; do for v in default cmax256; do
  if [ "$v" = default ]; then f=""; else f="cmax:256"; fi
  BIU_FOLDT="$f" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 --breakdown gpurun_out/r4p/bd_${i}_$v.txt > gpurun_out/r4p/b_${i}_$v.json 2>/dev/null
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), round(d['fwd_only']['ms'],3))" gpurun_out/r4p/b_${i}_$v.json
done; done
