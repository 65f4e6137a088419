set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -x -q -k "accumulators or small_map" > gpurun_out/r3g_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3g_tests.log
rm -f gpurun_out/r3g_bench.txt
for i in 1 2; do
  for t in 1 0; do
    echo "== YOLO_STAT_ACC=$t" >> gpurun_out/r3g_bench.txt
    YOLO_STAT_ACC=$t timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>gpurun_out/r3g_bench.err | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['final_loss'])" >> gpurun_out/r3g_bench.txt
  done
done
cat gpurun_out/r3g_bench.txt
for t in 1; do
  rm -rf gpurun_out/r3g_trace$t
  YOLO_STAT_ACC=$t rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3g_trace$t -o t -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline > gpurun_out/r3g_trace$t.log 2>&1
  cp $(ls gpurun_out/r3g_trace$t/*kernel_stats.csv gpurun_out/r3g_trace$t/*/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/r3g_stats$t.csv
  rm -rf gpurun_out/r3g_trace$t
done
