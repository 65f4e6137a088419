set -o pipefail
export TMPDIR=/tmp
timeout -k 10 400 python tools/input_pipeline_bench.py --real --images 640 --train --procs 16 12 --batches 100 > gpurun_out/r3i_pipe_train_real.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r3i_pipe_train_real.txt | tail -5
timeout -k 10 400 python tools/input_pipeline_bench.py --real --images 640 --procs 16 --workers 8 --batches 100 > gpurun_out/r3i_pipe_real.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r3i_pipe_real.txt
python -m pytest tests/test_trainer_e2e_gpu.py tests/test_config1_gpu.py tests/test_reference_default_run_gpu.py -x -q > gpurun_out/r3i_tests2.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3i_tests2.log
