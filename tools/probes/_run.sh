set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r5b
rm -rf $out && mkdir -p $out
for t in "0 lead_grid=1024" "1 lead_grid=2048" "1 lead_grid=512" "1 lead_grid=256" "0 lead_grid=1024" "1 lead_grid=1024"; do
  set -- $t
  YOLO_LEAD_FIN=$1 YOLO_TUNE=$2 timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline > $out/b.json 2>$out/b.err || { echo FAILED; tail -5 $out/b.err; exit 1; }
  python -c "import json; d=json.loads(open('$out/b.json').read().strip().splitlines()[-1]); print('lead_fin %-18s  %8.1f img/s  %.4f ms  loss %s' % ('$t', d['value'], d['ms_per_step'], d['config']['final_loss']))" | tee -a $out/ab.txt
done
