set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r4w
rm -rf $out && mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_s32_gpu.py -x -q > $out/test.log 2>&1; echo "pytest rc $?" >> $out/test.log
tail -4 $out/test.log
S32_LAYERS=52 timeout -k 10 300 python tools/probes/s32_sweep.py 1,5,8 3 2>&1 | grep -v amdgpu.ids | tee $out/sweep.txt
