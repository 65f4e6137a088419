set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_pstrip_gpu.py -x -q > gpurun_out/r3d_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3d_tests.log; tail -4 gpurun_out/r3d_tests.log
rm -f gpurun_out/r3d_one.txt
for shape in "32 26 26 256 256" "32 52 52 128 128" "32 52 52 128 256" "32 26 26 256 512"; do
  echo "== $shape" >> gpurun_out/r3d_one.txt
  timeout -k 10 200 python tools/probes/conv_one.py $shape "pstrip=0" "pstrip=1,ps_depth=1" "pstrip=1" "pstrip=2" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3d_one.txt
done
cat gpurun_out/r3d_one.txt
cd yolov3_tensorflow_amd/csrc
touch conv_pstrip.hip
make -j8 EXTRA_conv_pstrip="-DPS_STAMPS" 2>&1 | grep -E "error"
cd ../..
rm -f gpurun_out/r3d_stamps.txt
for shape in "32 26 26 256 256" "32 52 52 128 128"; do
  timeout -k 10 120 python tools/probes/pstrip_stamps.py $shape pstrip=1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3d_stamps.txt
done
cat gpurun_out/r3d_stamps.txt
