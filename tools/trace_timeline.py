"""Timeline of the last hipGraph replay in a rocprofv3 --kernel-trace csv: kernels in start order with their offset, duration, queue,
the gap to the previous kernel's end on the merged timeline, and the totals (span, union of busy time, sum of durations).
    python3 tools/trace_timeline.py <kernel_trace.csv> <kernels per step> [out.txt]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
out = open(sys.argv[3], 'w') if len(sys.argv) > 3 else sys.stdout
rows.sort(key=lambda r: int(r['Start_Timestamp']))
step = rows[-n:]
t0 = int(step[0]['Start_Timestamp'])
end_prev, busy, total = t0, 0, 0
cur_s, cur_e = None, None
print(f'columns: {list(rows[0].keys())}', file=out)
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    total += e - s
    if cur_s is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    gap = s - end_prev
    end_prev = max(end_prev, e)
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {gap / 1e3:6.1f}  q{r.get('Queue_Id', '?'):>3s}  "
          f"grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8s}  {r['Kernel_Name'][:100]}", file=out)
busy += cur_e - cur_s
print(f'span {(end_prev - t0) / 1e3:.1f} us, busy union {busy / 1e3:.1f} us, sum of durations {total / 1e3:.1f} us, {len(step)} kernels', file=out)
