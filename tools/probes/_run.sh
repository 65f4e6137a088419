set -o pipefail
export TMPDIR=/tmp
for i in 1 2; do
  for t in "pstrip=0" "pstrip=-1"; do
    echo "== $t" >> gpurun_out/r3e_bench.txt
    YOLO_TUNE=$t timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/r3e_bench.txt
  done
done
cat gpurun_out/r3e_bench.txt
