set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r5g
rm -rf $out && mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_train_step_gpu.py -x -q -s -k "forward_loss_grads_and_step" > $out/test.log 2>&1; echo "pytest rc $?" >> $out/test.log
tail -3 $out/test.log
