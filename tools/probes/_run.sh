set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 200 python tools/probes/clock_sample.py 4000 2>&1 | tee gpurun_out/clock_sample.txt
