#!/bin/bash
# One profiling pass of the headline command on the GPU box: kernel trace + stats, the step timeline, the two HBM PMC passes (FETCH_SIZE and
# WRITE_SIZE in SEPARATE runs, as MI355X_MICROARCH.md prescribes), the MFMA-busy pass (single stream), and the bench line itself.
# Usage (from the repo root, under gpurun): bash tools/profile_round.sh r02_a      -> gpurun_out/r02_a/*  (copy what is to be judged into profiles/)
set -o pipefail
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline"
find_csv() { ls $1/*$2.csv $1/*/*$2.csv 2>/dev/null | head -1; }

python bench.py --steps 30 --warmup 5 > $out/bench.log 2>&1 && grep '^{' $out/bench.log | tail -1 > $out/bench.json

rm -rf $out/trace && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python $B > $out/trace.log 2>&1
cp "$(find_csv $out/trace kernel_stats)" $out/kernel_stats.csv
python tools/trace_analyze.py "$(find_csv $out/trace kernel_trace)" full > $out/step_timeline.txt 2>&1

for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc_$c && rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o p -- python $B > $out/pmc_$c.log 2>&1
done
python tools/hbm_traffic.py "$(find_csv $out/pmc_FETCH_SIZE counter_collection)" "$(find_csv $out/pmc_WRITE_SIZE counter_collection)" $out/pmc_hbm_traffic.json

rm -rf $out/pmc_mfma && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA \
  --output-format csv -d $out/pmc_mfma -o m -- python $B --no-overlap > $out/pmc_mfma.log 2>&1
python tools/mfma_util.py "$(find_csv $out/pmc_mfma counter_collection)" $out/pmc_mfma_util.json

rm -rf $out/trace $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_mfma      # raw traces are large; the summaries stay
ls -la $out
