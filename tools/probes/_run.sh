set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r6f
rm -rf $out && mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/test.log 2>&1; rc=$?; echo "pytest rc $rc" >> $out/test.log
tail -4 $out/test.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
bash tools/profile_round.sh r04_e > $out/profile_round.log 2>&1; tail -2 $out/profile_round.log
