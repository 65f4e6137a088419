set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r3e_bench_ab.txt
for i in 1 2; do
for t in "stream=0" "stream=-1"; do
  echo "== $t" >> gpurun_out/r3e_bench_ab.txt
  YOLO_TUNE=$t timeout -k 10 300 python bench.py --steps 60 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> gpurun_out/r3e_bench_ab.txt || exit 1
done; done
cat gpurun_out/r3e_bench_ab.txt
