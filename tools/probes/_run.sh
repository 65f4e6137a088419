set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r04_c2
rm -rf $out && mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline "$@" > $out/$name.log 2>&1 || { echo "$name FAILED"; tail -5 $out/$name.log; return 1; }
  grep '^{' $out/$name.log | tail -1 > $out/bench_$name.json
  python -c "import json; d=json.load(open('$out/bench_$name.json')); print('%-32s %8.1f img/s  %.4f ms' % ('$name', d['value'], d['ms_per_step']))" | tee -a $out/summary.txt; }
run headline &&
run mixnet18 --backbone mixnet-18 &&
run resnet18v2_608_fp16_focal --backbone resnet-18-v2 --size 608 --batch 16 --dtype fp16 --focal &&
run batch64 --batch 64 &&
run 320_b8 --size 320 --batch 8 --classes 13 &&
run fp16 --dtype fp16
