# scratch runner for gpurun calls (rewritten per experiment): bash tools/probes/_run.sh
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r3j_v2
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python bench.py --steps 12 --warmup 4 --backbone resnet-18-v2 --size 608 --batch 16 --dtype fp16 --focal --no-cpu-baseline --no-roofline > $out/trace.log 2>&1 || exit 1
python tools/trace_analyze.py "$(ls $out/trace/*kernel_trace.csv $out/trace/*/*kernel_trace.csv 2>/dev/null | head -1)" full > $out/step_timeline.txt 2>&1
rm -rf $out/trace
sed -n 1,40p $out/step_timeline.txt
