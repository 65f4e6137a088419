set -o pipefail
mkdir -p gpurun_out/r03_b
O=gpurun_out/r03_b
timeout -k 10 300 python -m pytest tests/test_stream_gpu.py -x -q > $O/stream_tests.txt 2>&1; echo "stream tests rc=$?"; tail -2 $O/stream_tests.txt
timeout -k 10 200 python tools/probes/conv_one.py 32 104 104 64 64 "stream=0" "stream=1" 2>&1 | grep -v amdgpu.ids | tee $O/stream_time.txt
timeout -k 10 300 python tools/loss_curve.py --out $O/loss_curve_config1.json > $O/loss_curve.log 2>&1; echo "curve rc=$?"
timeout -k 10 300 python tools/loss_curve.py --dtype float16 --out $O/loss_curve_config1_fp16.json > $O/loss_curve_fp16.log 2>&1; echo "curve16 rc=$?"
timeout -k 10 200 python tools/probes/conv_layers.py > $O/conv_layers_alone.txt 2>&1; echo "layers rc=$?"
timeout -k 10 200 python tools/probes/wgrad_layers.py > $O/wgrad_layers_alone.txt 2>&1; echo "wlayers rc=$?"
tail -3 $O/conv_layers_alone.txt $O/wgrad_layers_alone.txt
