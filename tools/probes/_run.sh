set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r4y
rm -rf $out && mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_parallel_gpu.py -x -q > $out/test_par.log 2>&1; rc=$?; echo "pytest rc $rc" >> $out/test_par.log
tail -3 $out/test_par.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --deselect tests/test_parallel_gpu.py > $out/test.log 2>&1; rc=$?; echo "pytest rc $rc" >> $out/test.log
tail -5 $out/test.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/loss_curve.py --out $out/r04_loss_curve_config1.json > $out/lc_bf16.log 2>&1 && tail -3 $out/lc_bf16.log &&
timeout -k 10 300 python tools/loss_curve.py --dtype float16 --out $out/r04_loss_curve_config1_fp16.json > $out/lc_fp16.log 2>&1 && tail -3 $out/lc_fp16.log
