set -o pipefail
export TMPDIR=/tmp
rm -f gpurun_out/r3h_bench.txt
for i in 1 2 3; do
  for t in 0 3 11 19 27; do
    echo "== ew_nt=$t" >> gpurun_out/r3h_bench.txt
    YOLO_TUNE=ew_nt=$t timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>gpurun_out/r3h_bench.err | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['final_loss'])" >> gpurun_out/r3h_bench.txt
  done
done
cat gpurun_out/r3h_bench.txt
