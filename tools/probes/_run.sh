# scratch runner for gpurun calls (rewritten per experiment): bash tools/probes/_run.sh
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python tools/probes/stream_race.py 6000 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r03_stream_race.txt
