#!/usr/bin/env python
"""Pivot rocprofv3 --pmc counter_collection CSVs: one line per igemm dispatch (last of each kernel/grid shown).
Usage: python tools/pmc_table.py a_counter_collection.csv [b_counter_collection.csv ...]"""
import csv, sys, re, collections
csv.field_size_limit(1 << 30)
for path in sys.argv[1:]:
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        n = r['Kernel_Name']
        if 'igemm' not in n and 'bn_' not in n and 'strip' not in n:
            continue
        n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n).split('(')[0]
        key = (n, r['Grid_Size'], r['Workgroup_Size'])
        d.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
        d[key]['_t'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        d[key]['_v'] = '%s+%s' % (r['VGPR_Count'], r['Accum_VGPR_Count'])
        d[key]['_lds'] = r['LDS_Block_Size']
    for key, c in d.items():
        print('%-46s grid %9s wg %4s vgpr %s lds %s  %.1f us' % (key[0][:46], key[1], key[2], c['_v'], c['_lds'], c['_t']))
        wc = c.get('SQ_WAVE_CYCLES')
        for k, v in c.items():
            if k.startswith('_'):
                continue
            print('      %-28s %14.0f %s' % (k, v, ('%.1f%% of wave cycles' % (100 * v / wc)) if wc and k != 'SQ_WAVE_CYCLES' else ''))
