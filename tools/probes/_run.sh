# scratch runner for gpurun calls (rewritten per experiment): bash tools/probes/_run.sh
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r4d
rm -rf $out && mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_s32_gpu.py -q -x > $out/test_s32.log 2>&1; echo "pytest rc $?" >> $out/test_s32.log
tail -4 $out/test_s32.log
timeout -k 10 400 python tools/probes/s32_sweep.py 0,1,2,4,6 3 > $out/sweep.log 2>&1 || { tail -20 $out/sweep.log; exit 1; }
cat $out/sweep.log
for args in "1 32 26 26 256 256" "1 32 52 52 128 128" "1 32 104 104 64 64"; do
  echo "=== s32_stamps $args" >> $out/stamps.log
  timeout -k 10 120 python tools/probes/s32_stamps.py $args >> $out/stamps.log 2>&1 || { tail -20 $out/stamps.log; exit 1; }
done
cat $out/stamps.log
