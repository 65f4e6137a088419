set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests.txt 2>&1
echo "tests rc=$?"
tail -6 gpurun_out/r03_gpu_tests.txt
