set -o pipefail
export TMPDIR=/tmp
timeout -k 10 300 python tools/focal_gap.py > gpurun_out/r3f_focal_gap.json 2> gpurun_out/r3f_focal_gap.err; echo "focal rc=$?"
YOLO_BENCH_SHARE_GPU=1 YOLO_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 2 --no-roofline > gpurun_out/r3f_bench2.log 2>&1; echo "bench2 rc=$?"; grep '^{' gpurun_out/r3f_bench2.log | tail -1 > gpurun_out/r3f_bench2.json
python -m pytest tests/test_parallel_gpu.py -x -q > gpurun_out/r3f_tests2.log 2>&1; echo "tests2 rc=$?"; tail -3 gpurun_out/r3f_tests2.log
