"""Idle time between the kernels of ONE batched SVD call, from a rocprofv3 --kernel-trace CSV: which gaps (host round trips,
descriptor uploads, launch latency) sit between which kernels.  python scripts/trace_gaps.py <kernel_trace.csv> [call index]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('cyb::', '').split('(')[0].split('<')[0][:34]
# split into calls at the big idle gaps (> 2 ms) -- svd_bench separates calls by host work
calls, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if int(b['Start_Timestamp']) - int(a['End_Timestamp']) > 2_000_000:
        calls.append(cur)
        cur = []
    cur.append(b)
calls.append(cur)
calls = [c for c in calls if len(c) > 100]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(calls) - 1
c = calls[which]
t0, t1 = int(c[0]['Start_Timestamp']), int(c[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in c)
print(f'{len(calls)} calls; call {which}: {len(c)} kernels, span {(t1 - t0) / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms')
gaps = defaultdict(lambda: [0, 0.0])
big = []
for a, b in zip(c, c[1:]):
    g = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    if g > 0:
        k = (name(a), name(b))
        gaps[k][0] += 1
        gaps[k][1] += g / 1e3
        if g > 30_000:
            big.append((g / 1e3, name(a), name(b), (int(a['End_Timestamp']) - t0) / 1e6))
print('gaps by (kernel before -> kernel after), total us:')
for k, (n, us) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f'  {us:9.1f} us in {n:4d} gaps (avg {us / n:6.1f})  {k[0]} -> {k[1]}')
print('gaps > 30 us:')
for g, a, b, t in big:
    print(f'  at {t:7.3f} ms: {g:7.1f} us  {a} -> {b}')
per = defaultdict(lambda: [0, 0.0])
for r in c:
    per[name(r)][0] += 1
    per[name(r)][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('kernels:')
for k, (n, us) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f'  {us / 1e3:8.3f} ms {n:5d} x {us / n:8.1f} us  {k}')
