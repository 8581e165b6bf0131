"""How much of a rocprofv3 kernel trace ran concurrently: sum of kernel durations vs the union of their intervals, per queue.
usage: trace_overlap.py <dir with *_kernel_trace.csv> [last N kernels]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
if len(sys.argv) > 2:
    rows = rows[-int(sys.argv[2]):]
iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
tot = sum(e - s for s, e in iv)
cur_s, cur_e, union = iv[0][0], iv[0][1], 0
for s, e in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = max(e for _, e in iv) - iv[0][0]
queues = {}
for r in rows:
    queues[r['Queue_Id']] = queues.get(r['Queue_Id'], 0) + 1
print(f'{len(rows)} kernels: sum of durations {tot / 1e6:.3f} ms, union {union / 1e6:.3f} ms, span {span / 1e6:.3f} ms, queues {queues}')
