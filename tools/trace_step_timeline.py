"""Timeline of one captured training step from a rocprofv3 kernel trace (CSV): per hardware queue the first start / last end / busy
time, the gaps on the main queue, and the kernels of the other queues with their start times -- shows whether a side branch of the
graph starts when its inputs are ready and whether the step is the sum or the maximum of its branches.

    python3 tools/trace_step_timeline.py <kernel_trace.csv> [name of the step's last kernel, default adamw]
"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
last = sys.argv[2] if len(sys.argv) > 2 else 'adamw'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if last in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
seg = rows[a + 1:b + 1]
t0 = int(seg[0]['Start_Timestamp'])
us = lambda t: (int(t) - t0) / 1e3
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '')[:60]
qs = {}
for r in seg:
    qs.setdefault(r['Queue_Id'], []).append(r)
main = max(qs, key=lambda q: len(qs[q]))
print(f'step: {len(seg)} kernels, {us(seg[-1]["End_Timestamp"]):.1f} us from the first start to the end of {short(seg[-1]["Kernel_Name"])}')
for q, rs in qs.items():
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e3
    print(f'queue {q}{" (main)" if q == main else ""}: {len(rs)} kernels, first start {us(rs[0]["Start_Timestamp"]):.1f}, last end {us(rs[-1]["End_Timestamp"]):.1f}, busy {busy:.1f} us')
prev = None
gaps = []
for r in qs[main]:
    if prev is not None:
        g = us(r['Start_Timestamp']) - prev
        if g > 3.0:
            gaps.append((g, us(r['Start_Timestamp']), short(r['Kernel_Name'])))
    prev = us(r['End_Timestamp'])
print(f'main-queue gaps > 3 us: {len(gaps)}, {sum(g for g, _, _ in gaps):.1f} us in total')
for g, t, n in sorted(gaps, reverse=True)[:12]:
    print(f'   {g:7.1f} us before {n} at {t:.1f}')
for q, rs in qs.items():
    if q == main:
        continue
    print(f'queue {q}:')
    for r in rs:
        print(f'   {us(r["Start_Timestamp"]):8.1f} .. {us(r["End_Timestamp"]):8.1f}  {short(r["Kernel_Name"])}')
