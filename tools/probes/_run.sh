set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1100 python tools/loss_curve_scatter.py --out gpurun_out/r04_loss_curve_scatter.json > gpurun_out/scatter.log 2>&1; tail -25 gpurun_out/scatter.log | cut -c1-160
