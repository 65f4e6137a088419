# scratch runner for gpurun calls (rewritten per experiment): bash tools/probes/_run.sh
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r4e
rm -rf $out && mkdir -p $out
for i in 1 2 3; do
  for t in "s32=0" "s32=-1"; do
    YOLO_TUNE=$t timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline > $out/bench_${t}_$i.json 2>$out/bench_${t}_$i.err || { tail -5 $out/bench_${t}_$i.err; exit 1; }
    python -c "import json,sys; d=json.loads(open('$out/bench_${t}_$i.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'])"
  done
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; echo "pytest rc $?" >> $out/gpu_tests.log
tail -6 $out/gpu_tests.log
