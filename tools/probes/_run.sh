# scratch runner for gpurun calls (rewritten per experiment): bash tools/probes/_run.sh
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -1
