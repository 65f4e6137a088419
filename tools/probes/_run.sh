set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r6c
rm -rf $out && mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_s32_gpu.py tests/test_train_step_gpu.py tests/test_configs_gpu.py tests/test_row_groups_gpu.py tests/test_stream_gpu.py -x -q > $out/test.log 2>&1; rc=$?; echo "pytest rc $rc" >> $out/test.log; tail -5 $out/test.log | cut -c1-200
