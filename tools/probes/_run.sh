# scratch runner for gpurun calls (rewritten per experiment): bash tools/probes/_run.sh
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python __graft_entry__.py smoke > gpurun_out/r03_smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r03_smoke.txt
