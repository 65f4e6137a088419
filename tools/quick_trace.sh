#!/bin/bash
# kernel trace + stats + timeline of the headline command only: tools/quick_trace.sh <tag>
set -o pipefail
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
find_csv() { ls $1/*$2.csv $1/*/*$2.csv 2>/dev/null | head -1; }
rm -rf $out/trace && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline > $out/trace.log 2>&1
cp "$(find_csv $out/trace kernel_stats)" $out/kernel_stats.csv
python tools/trace_analyze.py "$(find_csv $out/trace kernel_trace)" full > $out/step_timeline.txt 2>&1
rm -rf $out/trace
