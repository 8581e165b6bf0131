"""MFMA-busy / issue counters of the kernels that carry the step, from rocprofv3 PMC passes (north_star: "evidenced by rocprof HBM
GB/s and MFMA-busy").

    python3 tools/pmc_issue.py profiles/r03/pmc_issue.json          (on the GPU box; runs rocprofv3 itself)

Per kernel, separate --pmc passes with --kernel-trace only (SQ has 8 slots per pass, GRBM 2: MI355X_MICROARCH.md "rocprofv3 PMC
slots"); driver tools/prof_kernels.py at config-4 shapes, the program directly behind `--`.  Units (same guide, cycle-constants
table): SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over the waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Derived:
    kernel_cycles = GRBM_GUI_ACTIVE / 8
    mfma_busy     = SQ_VALU_MFMA_BUSY_CYCLES / (kernel_cycles * 256 CUs * 4 SIMDs)      (the gfx94x MfmaUtil formula)
    wait_inst_any / wait_any / active_inst_any = share of SQ_WAVE_CYCLES (they are disjoint and sum to ~1).
"""
import collections, csv, glob, hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE'],
          ['SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAVES'],
          ['SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_INSTS_VMEM'],
          ['SQ_INSTS_VALU_MFMA_MOPS_BF16', 'SQ_VALU_MFMA_COEXEC_CYCLES', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE']]
KERNELS = [   # key, prof_kernels.py driver, kernel-name substring, sources
    ('attn_fwd_row16_kernel', 'attn', 'attn_fwd_row16_kernel', ['attn_fwd_row16.hip', 'attn_common.h', 'wmz_common.h']),
    ('attn_bwd_row16_kernel<dq>', 'attn_bwd', 'attn_bwd_row16_kernel<128, 0', ['attn_bwd_row16.hip', 'attn_common.h', 'wmz_common.h']),
    ('attn_bwd_kvplane_kernel<dk|dv>', 'attn_bwd', 'attn_bwd_kvplane_kernel<128', ['attn_bwd_row16.hip', 'attn_common.h', 'wmz_common.h']),
    ('layer_fused_kernel<head,tail>', 'fused', 'layer_fused_kernel<256, 128, 256, true, true>', ['layer_fused.hip', 'fused_common.h', 'wmz_common.h']),
]


def src_hash(names):
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(ROOT, 'world_modelz_amd', 'csrc', n), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def one_pass(which, counters, n=8):
    d = os.path.join(ROOT, 'gpurun_out', f'pmci_{which}_{counters[0]}')
    subprocess.run(['rm', '-rf', d])
    r = subprocess.run(['rocprofv3', '--pmc'] + counters + ['--kernel-trace', '--output-format', 'csv', '-d', d, '--',
                        'python3', os.path.join(ROOT, 'tools', 'prof_kernels.py'), which, str(n)], capture_output=True, text=True,
                       cwd=ROOT, env=dict(os.environ, TMPDIR='/tmp'))
    if r.returncode != 0:
        print(f'[pmc_issue] pass {counters} on {which} failed: {r.stderr[-400:]}', flush=True)
        return {}
    fs = glob.glob(os.path.join(d, '*', '*_counter_collection.csv'))
    if not fs:
        return {}
    rows = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel name -> counter -> values per dispatch
    for row in csv.DictReader(open(fs[0])):
        rows[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
    return rows


def main():
    out = {}
    only = os.environ.get('PMC_ONLY')                     # e.g. PMC_ONLY=attn_bwd: just that driver's kernels
    kernels = [k for k in KERNELS if not only or k[1] == only]
    drivers = sorted({k[1] for k in kernels})
    data = {w: [one_pass(w, p) for p in PASSES] for w in drivers}
    for key, which, match, sources in kernels:
        ent = {}
        for rows in data[which]:
            for name, ctrs in rows.items():
                if match in name:
                    for c, vals in ctrs.items():
                        vals = vals[3:] if len(vals) > 5 else vals            # the first launches warm caches / clocks
                        ent[c] = sum(vals) / len(vals)
        if 'GRBM_GUI_ACTIVE' in ent and 'SQ_VALU_MFMA_BUSY_CYCLES' in ent:
            kc = ent['GRBM_GUI_ACTIVE'] / 8.0
            ent['kernel_cycles'] = kc
            ent['mfma_busy'] = ent['SQ_VALU_MFMA_BUSY_CYCLES'] / (kc * 256 * 4)
        if 'SQ_WAVE_CYCLES' in ent:
            for c, k2 in (('SQ_WAIT_INST_ANY', 'wait_inst_any'), ('SQ_WAIT_ANY', 'wait_any'), ('SQ_ACTIVE_INST_ANY', 'active_inst_any')):
                if c in ent:
                    ent[k2 + '_share_of_wave_cycles'] = ent[c] / ent['SQ_WAVE_CYCLES']
        ent['sources'] = sources
        ent['source_sha16'] = src_hash(sources)
        out[key] = ent
        print(key, json.dumps(ent), flush=True)
    out['_how'] = __doc__
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'pmc_issue.json')
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
