set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r6i
rm -rf $out && mkdir -p $out
timeout -k 10 200 python tools/probes/wgrad9_epi_check.py 2>&1 | grep -v amdgpu.ids | tee $out/check.txt
grep -q DIFFERS $out/check.txt && exit 1
for i in 1 2 3; do
  for t in "wgrad9_epi=0" "wgrad9_epi=1" "wgrad9_epi=1,wgrad9_wgs=160" "wgrad9_epi=1,wgrad9_wgs=192" "wgrad9_epi=1,wgrad9_wgs=256"; do
    YOLO_TUNE=$t timeout -k 10 200 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-roofline > $out/b.json 2>$out/b.err || { echo FAILED; tail -5 $out/b.err; exit 1; }
    python -c "import json; d=json.loads(open('$out/b.json').read().strip().splitlines()[-1]); print('%-30s  %8.1f img/s  %.4f ms' % ('$t', d['value'], d['ms_per_step']))" | tee -a $out/ab.txt
  done
done
