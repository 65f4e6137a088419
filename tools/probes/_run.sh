set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r6h
rm -rf $out && mkdir -p $out
for i in 1 2 3; do
  for t in 2048 1024 4096 512; do
    YOLO_TUNE=ew_grid=$t timeout -k 10 200 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-roofline > $out/b.json 2>$out/b.err || { echo FAILED; tail -5 $out/b.err; exit 1; }
    python -c "import json; d=json.loads(open('$out/b.json').read().strip().splitlines()[-1]); print('ew_grid %-6s  %8.1f img/s  %.4f ms' % ('$t', d['value'], d['ms_per_step']))" | tee -a $out/ab.txt
  done
done
