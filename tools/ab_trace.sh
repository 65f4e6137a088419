#!/bin/bash
# kernel-trace A/B of one environment switch: tools/ab_trace.sh <tag> <ENV=VALUE>
set -o pipefail
tag=$1; sw=$2
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline"
find_csv() { ls $1/*$2.csv $1/*/*$2.csv 2>/dev/null | head -1; }
for v in on off; do
  rm -rf $out/trace_$v
  if [ $v = off ]; then export $sw; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$v -o t -- python $B > $out/trace_$v.log 2>&1
  cp "$(find_csv $out/trace_$v kernel_stats)" $out/kernel_stats_$v.csv
  python tools/trace_analyze.py "$(find_csv $out/trace_$v kernel_trace)" full > $out/step_timeline_$v.txt 2>&1
  rm -rf $out/trace_$v
done
