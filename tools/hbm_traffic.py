#!/usr/bin/env python
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs of the same command) into per-kernel KB per launch.
Usage: python tools/hbm_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json
FETCH_SIZE on gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section): `fetch_corrected` doubles it."""
import collections
import csv
import json
import re
import sys

csv.field_size_limit(1 << 30)


def load(path, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        n = re.sub(r'^void ', '', n).split('(')[0]
        tot[n] += float(r['Counter_Value'])
        cnt[n] += 1
    return tot, cnt


def main():
    f, fc = load(sys.argv[1], 'FETCH_SIZE')
    w, wc = load(sys.argv[2], 'WRITE_SIZE')
    rows = []
    for n in sorted(f, key=lambda k: -f[k]):
        if fc[n] == 0 or n.startswith('at::') or 'elementwise' in n:
            continue
        rows.append({'kernel': n, 'launches': fc[n], 'FETCH_SIZE_KB_per_launch': round(f[n] / fc[n], 1),
                     'fetch_corrected_KB_per_launch': round(2 * f[n] / fc[n], 1),
                     'WRITE_SIZE_KB_per_launch': round(w.get(n, 0.0) / max(wc.get(n, 0), 1), 1)})
    json.dump({'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the same bench.py command; KB per launch averaged '
                       'over the launches of all layer shapes; FETCH_SIZE on gfx950 counts wide coalesced reads at half their bytes '
                       '(MI355X_MICROARCH.md): fetch_corrected doubles it', 'kernels': rows}, open(sys.argv[3], 'w'), indent=1)


if __name__ == '__main__':
    main()
