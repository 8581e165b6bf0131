"""Fabric-side traffic per launch of the two kernels that carry the forward step, from rocprofv3 PMC passes.

    python3 tools/pmc_traffic.py profiles/r02/pmc_traffic.json          (on the GPU box; runs rocprofv3 itself)

Two separate passes per kernel (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2: they do not fit one pass), --kernel-trace
only, driver tools/prof_kernels.py {attn|fused} N at config-4 shapes.  Corrections exactly as MI355X_MICROARCH.md (HBM
section) prescribes for gfx950: FETCH_SIZE is reported in KB and tallies the 128-B requests of wide coalesced reads at 64
B -> bytes = KB * 1024 * 2; WRITE_SIZE is exact for 16-B-per-lane stores.  Infinity-Cache hits are counted, so this is
fabric-side traffic: an upper bound on HBM bytes.  The kernels' source hash is stored with the numbers; bench.py reports
`traffic` only while the sources still hash to it.
"""
import collections, csv, glob, hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {
    'attn_fwd_row16_kernel': ('attn', 'attn_fwd_row16_kernel', ['attn_fwd_row16.hip', 'attn_common.h', 'wmz_common.h']),
    'layer_fused_kernel<head,tail>': ('fused', 'layer_fused_kernel<256, 128, 256, true, true>', ['layer_fused.hip', 'fused_common.h', 'wmz_common.h']),
    'attn_bwd_row16_kernel<dq>': ('attn_bwd', 'attn_bwd_row16_kernel<128, 0', ['attn_bwd_row16.hip', 'attn_common.h', 'wmz_common.h']),
    'attn_bwd_kvplane_kernel<dk|dv>': ('attn_bwd', 'attn_bwd_kvplane_kernel<128', ['attn_bwd_row16.hip', 'attn_common.h', 'wmz_common.h']),
}


def src_hash(names):
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(ROOT, 'world_modelz_amd', 'csrc', n), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def one_pass(which, counter, match, n=8):
    d = os.path.join(ROOT, 'gpurun_out', f'pmc_{which}_{counter}')
    subprocess.run(['rm', '-rf', d])
    r = subprocess.run(['rocprofv3', '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', d, '--',
                        'python3', os.path.join(ROOT, 'tools', 'prof_kernels.py'), which, str(n)], capture_output=True, text=True,
                       cwd=ROOT, env=dict(os.environ, TMPDIR='/tmp'))
    assert r.returncode == 0, r.stderr[-2000:]
    f = glob.glob(os.path.join(d, '*', '*_counter_collection.csv'))[0]
    vals = [float(row['Counter_Value']) for row in csv.DictReader(open(f))
            if match in row['Kernel_Name'] and row['Counter_Name'] == counter]
    vals = vals[3:] if len(vals) > 5 else vals                     # the first launches warm the caches
    return sum(vals) / len(vals), len(vals)


def main():
    out = {}
    for key, (which, match, sources) in KERNELS.items():
        fkb, nf = one_pass(which, 'FETCH_SIZE', match)
        wkb, nw = one_pass(which, 'WRITE_SIZE', match)
        fetch, write = fkb * 1024 * 2, wkb * 1024
        out[key] = {'FETCH_SIZE_KB_raw': fkb, 'WRITE_SIZE_KB_raw': wkb, 'fetch_bytes_corrected': fetch, 'write_bytes': write,
                    'traffic_bytes_per_launch': fetch + write, 'launches_averaged': min(nf, nw), 'sources': sources,
                    'source_sha16': src_hash(sources)}
        print(key, out[key], flush=True)
    out['_how'] = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (--kernel-trace only), driver '
                   'tools/prof_kernels.py {fused|attn} 8, config-4 shapes (65 536 tokens, dh 128, window 7x7x7); FETCH_SIZE doubled '
                   '(gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are counted: '
                   'fabric-side traffic, an upper bound on HBM bytes')
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'pmc_traffic.json')
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
