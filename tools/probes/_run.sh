set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 800 python tools/loss_curve_scatter.py --out gpurun_out/r4n_loss_curve_scatter.json 2>&1 | grep -v amdgpu.ids
