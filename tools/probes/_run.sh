set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_stream_gpu.py -x -q > gpurun_out/r3e_stream_tests.txt 2>&1
echo "tests rc=$?" 
tail -5 gpurun_out/r3e_stream_tests.txt
timeout -k 10 300 python tools/probes/conv_one.py 32 104 104 64 64 "stream=0" "stream=1" > gpurun_out/r3e_stream_time.txt 2>&1
timeout -k 10 300 python tools/probes/conv_one.py 32 104 104 64 64 "stream=0" "stream=1" --dgrad >> gpurun_out/r3e_stream_time.txt 2>&1
cat gpurun_out/r3e_stream_time.txt
