set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python tools/probes/gemm_yardstick.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_yardstick.txt
