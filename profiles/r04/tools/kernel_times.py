"""Per-kernel average durations of one tools/prof_kernels.py driver under rocprofv3 --kernel-trace, optionally for a variant build
of the library (tools/build_variant.py):

    python3 tools/kernel_times.py attn_bwd 12 [tools/variants/libwmz_x.so] [name-filter]

Run on the GPU box.  The first three dispatches of every kernel are dropped (clocks / caches); the profiler's own overhead is in
the numbers (a few per cent), so compare variants with each other, not with un-profiled timings.
"""
import collections, csv, glob, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which, n = sys.argv[1], sys.argv[2]
lib = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] not in ('', '-') else None
flt = sys.argv[4] if len(sys.argv) > 4 else ''
tag = os.path.basename(lib)[:-3] if lib else 'product'
d = os.path.join(ROOT, 'gpurun_out', f'kt_{which}_{tag}')
subprocess.run(['rm', '-rf', d])
env = dict(os.environ, TMPDIR='/tmp')
if lib:
    env['WMZ_LIB_PATH'] = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
r = subprocess.run(['rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', d, '--', 'python3',
                    os.path.join(ROOT, 'tools', 'prof_kernels.py'), which, n], capture_output=True, text=True, cwd=ROOT, env=env)
if r.returncode != 0:
    sys.exit(f'rocprofv3 failed: {r.stderr[-600:]}')
fs = glob.glob(os.path.join(d, '*', '*_kernel_trace.csv'))
dur = collections.defaultdict(list)
for row in csv.DictReader(open(fs[0])):
    dur[row['Kernel_Name']].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
for name, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if flt and flt not in name:
        continue
    w = v[3:] if len(v) > 5 else v
    if len(v) < 3:
        continue
    print(f'[{tag}] {name[:90]:90s} calls {len(v):3d}  avg {sum(w) / len(w):8.2f} us  min {min(w):8.2f}')
