"""Fabric-side bytes of ONE frame-encoder call (VqAutoEncoder.encode of 256 frames of 64x64: every launch of it summed), from
rocprofv3 PMC passes -- what bench.py reports as frame_encoder.roofline.traffic.

    python3 tools/pmc_encode_total.py profiles/r06/pmc_frame_encoder.json          (on the GPU box; runs rocprofv3 itself)

Separate passes for FETCH_SIZE and WRITE_SIZE (--kernel-trace only beside --pmc), each over tools/prof_encode.py with 3 and
with 5 calls: the difference / 2 is the steady-state call (the first call packs weights).  Corrections as
MI355X_MICROARCH.md's HBM section prescribes for gfx950: FETCH_SIZE in KB, doubled; WRITE_SIZE in KB.  Infinity-Cache hits are
counted: fabric-side traffic, an upper bound on HBM bytes."""
import collections, csv, glob, hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ['conv_direct.hip', 'conv_point.hip', 'conv2d.hip', 'vq.hip', 'vq_screen.hip', 'bn_lazy.h', 'wmz_common.h']


def src_hash(names):
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(ROOT, 'world_modelz_amd', 'csrc', n), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def one_pass(counter, calls):
    d = f'/tmp/pmce_{counter}_{calls}'
    subprocess.run(['rm', '-rf', d])
    r = subprocess.run(['rocprofv3', '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', d, '--', 'python3',
                        os.path.join(ROOT, 'tools', 'prof_encode.py'), str(calls)], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, TMPDIR='/tmp'))
    assert r.returncode == 0, r.stderr[-2000:]
    f = glob.glob(os.path.join(d, '*', '*_counter_collection.csv'))[0]
    per = collections.Counter()
    n = collections.Counter()
    for row in csv.DictReader(open(f)):
        if row['Counter_Name'] == counter:
            k = row['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60]
            per[k] += float(row['Counter_Value'])
            n[k] += 1
    return per, n


def main():
    out = {'per_kernel': {}}
    tot = {}
    for counter, mul in (('FETCH_SIZE', 2048.0), ('WRITE_SIZE', 1024.0)):
        a, na = one_pass(counter, 3)
        b, nb = one_pass(counter, 5)
        tot[counter] = 0.0
        for k in b:
            v = (b[k] - a.get(k, 0.0)) / 2.0 * mul
            launches = (nb[k] - na.get(k, 0)) / 2.0
            if launches <= 0:
                continue
            e = out['per_kernel'].setdefault(k, {'launches_per_call': launches})
            e['fetch_bytes_corrected' if counter == 'FETCH_SIZE' else 'write_bytes'] = v
            tot[counter] += v
        print(counter, tot[counter], flush=True)
    out['fetch_bytes_corrected'] = tot['FETCH_SIZE']
    out['write_bytes'] = tot['WRITE_SIZE']
    out['traffic_bytes_per_call'] = tot['FETCH_SIZE'] + tot['WRITE_SIZE']
    out['frames_per_call'] = 256
    out['sources'] = SOURCES
    out['source_sha16'] = src_hash(SOURCES)
    out['_how'] = __doc__
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'pmc_frame_encoder.json')
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    json.dump(out, open(path, 'w'), indent=1)
    print('traffic per call', out['traffic_bytes_per_call'], 'per frame', out['traffic_bytes_per_call'] / 256)


if __name__ == '__main__':
    main()
