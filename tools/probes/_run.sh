# scratch runner for gpurun calls (rewritten per experiment): bash tools/probes/_run.sh
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_stream_gpu.py -x -q 2>&1 | tail -5
