set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r4u
rm -rf $out && mkdir -p $out
for i in 1 2 3; do
  for t in 0 1; do
    YOLO_ROW_GROUPS=$t timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline > $out/b.json 2>$out/b.err || { echo FAILED; tail -5 $out/b.err; exit 1; }
    python -c "import json; d=json.loads(open('$out/b.json').read().strip().splitlines()[-1]); print('row_groups=$t  %8.1f img/s  %.4f ms  loss %s' % (d['value'], d['ms_per_step'], d['config']['final_loss']))" | tee -a $out/ab.txt
  done
done
timeout -k 10 600 python -m pytest tests/test_train_step_gpu.py tests/test_configs_gpu.py tests/test_row_groups_gpu.py "tests/test_kernels_gpu.py::test_small_map_finalize_plus_apply_in_one_launch" -x -q > $out/test.log 2>&1; echo "pytest rc $?" >> $out/test.log
tail -8 $out/test.log
