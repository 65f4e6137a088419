set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3e_gpu_tests.txt 2>&1
echo "tests rc=$?"
tail -15 gpurun_out/r3e_gpu_tests.txt
