set -o pipefail
mkdir -p gpurun_out/r03_c
O=gpurun_out/r03_c
python bench.py --steps 30 --warmup 5 > $O/bench.log 2>&1 && grep '^{' $O/bench.log | tail -1 > $O/bench.json; echo "bench rc=$?"
python -c "
import json
d=json.load(open('$O/bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['achieved'], r['in_step']['achieved'], r['stream']['achieved'], r['stream']['in_step_achieved'], d['cpu_baseline']['value'])"
