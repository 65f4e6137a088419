# scratch runner for gpurun calls (rewritten per experiment): bash tools/probes/_run.sh
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "bwd_finalize_small or small_map_finalize" 2>&1 | tail -3
: > gpurun_out/r3i_fin_ab.txt
run() { ( env YOLO_TUNE="$1" timeout -k 10 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])" ) >> gpurun_out/r3i_fin_ab.txt || exit 1; }
for i in 1 2 3 4; do
  run "bwd_fin_small=1"
  run "bwd_fin_small=0"
done
cat gpurun_out/r3i_fin_ab.txt
