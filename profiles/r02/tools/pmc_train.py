"""Fabric-side traffic per launch of EVERY kernel of the training step (config 4), from two rocprofv3 PMC passes.

    python3 tools/pmc_train.py profiles/r02/train_pmc_traffic.json      (on the GPU box; runs rocprofv3 itself)

Same passes and gfx950 corrections as tools/pmc_traffic.py (FETCH_SIZE in KB, doubled; WRITE_SIZE in KB), driver
tools/prof_train.py N.  Output: kernel name -> launches per step, average fetch / write bytes per launch.
"""
import collections, csv, glob, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 6


def one_pass(counter):
    d = os.path.join(ROOT, 'gpurun_out', f'pmc_train_{counter}')
    subprocess.run(['rm', '-rf', d])
    r = subprocess.run(['rocprofv3', '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', d, '--',
                        'python3', os.path.join(ROOT, 'tools', 'prof_train.py'), str(STEPS)], capture_output=True, text=True,
                       cwd=ROOT, env=dict(os.environ, TMPDIR='/tmp'))
    assert r.returncode == 0, r.stderr[-2000:]
    f = glob.glob(os.path.join(d, '*', '*_counter_collection.csv'))[0]
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row['Counter_Name'] == counter:
            per[row['Kernel_Name']].append(float(row['Counter_Value']))
    return per


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0][:90]


def main():
    fetch, write = one_pass('FETCH_SIZE'), one_pass('WRITE_SIZE')
    out = {}
    for k in fetch:
        f, w = fetch[k], write.get(k, [0.0])
        # the first step's launches include cold caches / one-time packing: average the later ones
        cut = len(f) // STEPS
        fl, wl = (f[cut:] if len(f) > cut else f), (w[cut:] if len(w) > cut else w)
        out[short(k)] = {'launches_per_step': round(len(f) / STEPS, 2), 'fetch_bytes': sum(fl) / len(fl) * 2048,
                         'write_bytes': sum(wl) / len(wl) * 1024}
    rows = sorted(out.items(), key=lambda kv: -(kv[1]['fetch_bytes'] + kv[1]['write_bytes']) * kv[1]['launches_per_step'])
    for k, v in rows[:24]:
        print(f"{k:92s} x{v['launches_per_step']:5.2f}  fetch {v['fetch_bytes'] / 1e6:8.1f} MB  write {v['write_bytes'] / 1e6:8.1f} MB", flush=True)
    out['_how'] = (f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only, driver tools/prof_train.py '
                   f'{STEPS}; FETCH_SIZE doubled (gfx950), averages over steps 2..{STEPS}; fabric-side bytes (Infinity-Cache hits counted)')
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'train_pmc_traffic.json')
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, 'w') as f:
        json.dump(dict(rows, _how=out['_how']), f, indent=1)


if __name__ == '__main__':
    main()
