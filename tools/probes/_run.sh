set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r5y
rm -rf $out && mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_s32_gpu.py -x -q > $out/test.log 2>&1; rc=$?; echo "pytest rc $rc" >> $out/test.log; tail -4 $out/test.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/probes/s2_dgrad_time.py 2>&1 | grep -v amdgpu.ids | tee $out/s2_fold.txt
