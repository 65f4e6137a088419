#!/usr/bin/env python
"""Per-kernel MFMA pipe utilisation from a rocprofv3 --kernel-trace --pmc run (SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA): busy cycles / (launch duration x 2.4 GHz x 1024 SIMDs), averaged over launches.
Usage: python tools/mfma_util.py counter_collection.csv out.json"""
import collections
import csv
import json
import re
import sys

csv.field_size_limit(1 << 30)
SIMDS, GHZ = 1024, 2.4


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(sys.argv[1])):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        n = re.sub(r'^void ', '', n).split('(')[0]
        if n.startswith('at::') or 'elementwise' in n:
            continue
        per[n][r['Counter_Name']] += float(r['Counter_Value'])
        d = r['Dispatch_Id']
        if d not in seen[n]:
            seen[n].add(d)
            per[n]['_ns'] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    rows = []
    for n, c in sorted(per.items(), key=lambda kv: -kv[1]['_ns']):
        simd_cycles = c['_ns'] * GHZ * SIMDS
        wc = max(c.get('SQ_WAVE_CYCLES', 0.0), 1.0)
        rows.append({'kernel': n, 'launches': len(seen[n]), 'avg_us': round(c['_ns'] / len(seen[n]) / 1e3, 1),
                     'mfma_busy_pct_of_simd_cycles': round(100.0 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / simd_cycles, 1),
                     'wave_active_pct': round(100.0 * c.get('SQ_ACTIVE_INST_ANY', 0.0) / wc, 1),
                     'wave_wait_any_pct': round(100.0 * c.get('SQ_WAIT_ANY', 0.0) / wc, 1),
                     'wave_wait_inst_pct': round(100.0 * c.get('SQ_WAIT_INST_ANY', 0.0) / wc, 1)})
    json.dump({'note': 'rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY '
                       'SQ_INSTS_MFMA over bench.py --no-overlap (single stream); mfma_busy = busy cycles / (duration x 2.4 GHz x 1024 SIMDs); '
                       'durations under counter collection are a few % longer than in the kernel-trace-only profile', 'kernels': rows},
              open(sys.argv[2], 'w'), indent=1)


if __name__ == '__main__':
    main()
