#!/usr/bin/env python
"""Analyse a rocprofv3 --kernel-trace CSV of bench.py: per-step busy time (union of kernel intervals), summed kernel time, idle gaps,
and a per-kernel-family breakdown of the last full step.  Usage: python tools/trace_analyze.py <kernel_trace.csv>"""
import csv, sys, re, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')))
rows.sort()
# steps are delimited by pack_input_kernel launches (the first kernel of a step; the optimizer runs per gradient bucket inside the backward pass)
idx = [i for i, r in enumerate(rows) if 'pack_input' in r[2]]
print('kernels %d, steps %d' % (len(rows), len(idx)))
a, b = idx[-3], idx[-2]          # one full steady-state step: [schedule_k .. schedule_k+1)
step = rows[a:b]
t0, t1 = step[0][0], rows[b][0]
print('step wall %.1f us, launches %d' % ((t1 - t0) / 1e3, len(step)))
# union busy
busy = 0; cur_s, cur_e = step[0][0], step[0][1]
for s, e, _, _ in step[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('busy (union) %.1f us, idle %.1f us, summed kernel time %.1f us' % (busy / 1e3, (t1 - t0 - busy) / 1e3, sum(e - s for s, e, _, _ in step) / 1e3))
fam = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in step:
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = n.split('(')[0][:60]
    fam[n][0] += 1; fam[n][1] += e - s
for n, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print('%-62s %4d %9.1f us  avg %7.1f' % (n, c, t / 1e3, t / 1e3 / c))
if len(sys.argv) > 2:
    for s, e, n, q in step:
        n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n).split('(')[0][:50]
        print('%10.1f %8.1f q%s %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
