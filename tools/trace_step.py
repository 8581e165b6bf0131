"""One steady-state step out of a rocprofv3 --kernel-trace csv of graph replays: the step boundary is found from a kernel that runs once
a step (the AdamW launch), the step = the kernels between its last two occurrences -> the timeline (start offset, duration, gap to the previous end, queue, grid),
and a summary of the torch / runtime glue kernels (at::native, rocclr) in it.
    python3 tools/trace_step.py <kernel_trace.csv> [marker substring = adamw] [out.txt]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else 'adamw'
out = open(sys.argv[3], 'w') if len(sys.argv) > 3 else sys.stdout
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
step = rows[marks[-2] + 1:marks[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
end_prev, busy, total, glue, glue_t = t0, 0, 0, 0, 0
cur_s = cur_e = None
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
    name = r['Kernel_Name']
    is_glue = 'at::native' in name or 'rocclr' in name
    glue += is_glue
    glue_t += (e - s) if is_glue else 0
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {gap / 1e3:6.1f}  q{r.get('Queue_Id', '?'):>3s}  grid {r.get('Grid_Size_X', '?'):>8s}  "
          f"{'* ' if is_glue else '  '}{name[:110]}", file=out)
busy += cur_e - cur_s
print(f'step: {len(step)} kernels, span {(end_prev - t0) / 1e3:.1f} us, busy union {busy / 1e3:.1f} us, sum of durations {total / 1e3:.1f} us; '
      f'torch / runtime glue (*): {glue} kernels, {glue_t / 1e3:.1f} us', file=out)
