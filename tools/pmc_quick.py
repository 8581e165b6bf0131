"""PMC counters (comma-separated, one pass) over a driver command, averaged per (kernel, grid size):
    python3 tools/pmc_quick.py COUNTER[,COUNTER..] match -- cmd..."""
import collections, csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
counters, match = sys.argv[1].split(','), sys.argv[2]
counter = counters[0]
cmd = sys.argv[sys.argv.index('--') + 1:]
d = f'/tmp/pmcq_{counter}'
subprocess.run(['rm', '-rf', d])
r = subprocess.run(['rocprofv3', '--pmc'] + counters + ['--kernel-trace', '--output-format', 'csv', '-d', d, '--'] + cmd,
                   capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, TMPDIR='/tmp'))
assert r.returncode == 0, r.stderr[-2000:]
f = glob.glob(os.path.join(d, '*', '*_counter_collection.csv'))[0]
acc = collections.OrderedDict()
for row in csv.DictReader(open(f)):
    if match in row['Kernel_Name'] and row['Counter_Name'] in counters:
        acc.setdefault((row['Counter_Name'], row['Kernel_Name'][:50], row['Grid_Size']), []).append(float(row['Counter_Value']))
for (c, k, g), v in acc.items():
    v = v[3:] if len(v) > 5 else v
    print(f'{c} {k} grid {g}: mean {sum(v) / len(v):.1f} over {len(v)} launches', flush=True)
