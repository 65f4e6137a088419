set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r5f
rm -rf $out && mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/test.log 2>&1; rc=$?; echo "pytest rc $rc" >> $out/test.log
tail -4 $out/test.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
( time timeout -k 10 400 python bench.py ) > $out/bench_default.log 2>&1; grep '^{' $out/bench_default.log | tail -1 | cut -c1-400; grep real $out/bench_default.log
timeout -k 10 300 python tools/probes/soak.py 1500 1e-7 > $out/soak.log 2>&1; tail -4 $out/soak.log | cut -c1-200
