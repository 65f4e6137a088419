set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r5p
rm -rf $out && mkdir -p $out
for i in 1 2 3; do
  for t in libyolov3_amd_base.so libyolov3_amd.so libyolov3_amd_pipe.so; do
    YOLO_LIB_PATH=$PWD/yolov3_tensorflow_amd/$t timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline > $out/b.json 2>$out/b.err || { echo FAILED; tail -5 $out/b.err; exit 1; }
    python -c "import json; d=json.loads(open('$out/b.json').read().strip().splitlines()[-1]); print('%-24s  %8.1f img/s  %.4f ms  loss %s' % ('$t', d['value'], d['ms_per_step'], d['config']['final_loss']))" | tee -a $out/ab.txt
  done
done
