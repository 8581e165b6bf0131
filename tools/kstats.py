"""rocprofv3 --kernel-trace --stats of a command, summarised per kernel (run on the GPU box):

    python3 tools/kstats.py <steps-in-the-run> <out-name> -- python3 tools/prof_train.py 4 384 512 20 3 1 1

Prints kernel time per step by kernel (top 30) and copies the stats CSV to gpurun_out/<out-name>_kernel_stats.csv."""
import csv, glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps, name = int(sys.argv[1]), sys.argv[2]
cmd = sys.argv[sys.argv.index('--') + 1:]
d = os.path.join(ROOT, 'gpurun_out', 'ks_' + name)
subprocess.run(['rm', '-rf', d])
r = subprocess.run(['rocprofv3', '--kernel-trace', '--stats', '--output-format', 'csv', '-d', d, '--'] + cmd, cwd=ROOT,
                   env=dict(os.environ, TMPDIR='/tmp'), capture_output=True, text=True)
if r.returncode != 0:
    sys.exit('rocprofv3 failed: ' + r.stderr[-800:])
f = glob.glob(os.path.join(d, '*', '*kernel_stats.csv'))[0]
shutil.copy(f, os.path.join(ROOT, 'gpurun_out', name + '_kernel_stats.csv'))
rows = list(csv.DictReader(open(f)))
tot = sum(float(x['TotalDurationNs']) for x in rows)
print(f'[{name}] kernel time per step: {tot / steps / 1e6:.3f} ms ({len(rows)} kernels)')
for x in rows[:30]:
    nm = x['Name'].replace('(anonymous namespace)::', '')[:96]
    print(f"  {nm:96s} {int(x['Calls']) / steps:6.1f}/step avg {float(x['AverageNs']) / 1e3:8.1f} us {float(x['TotalDurationNs']) / steps / 1e6:7.3f} ms {float(x['Percentage']):5.1f}%")
