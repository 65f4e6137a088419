set -o pipefail
mkdir -p gpurun_out
R=$PWD
: > gpurun_out/r3f_bench_ab3.txt
run() { ( cd $1 && YOLO_STEM_POOL_FWD=$2 timeout -k 10 300 python bench.py --steps 60 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3', d['value'], d['ms_per_step'])" ) >> $R/gpurun_out/r3f_bench_ab3.txt || exit 1; }
for i in 1 2 3; do
  run $R/.ab_base 0 base_prefetch_only
  run $R 1 fused_fwd_recompute_bwd
  run $R 0 unfused_fwd_recompute_bwd
done
cat gpurun_out/r3f_bench_ab3.txt
