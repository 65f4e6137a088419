# scratch runner for gpurun calls (rewritten per experiment): bash tools/probes/_run.sh
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests_final.txt 2>&1
echo "tests rc=$?"
tail -6 gpurun_out/r03_gpu_tests_final.txt
