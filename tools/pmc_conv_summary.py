"""Summarise the conv kernels' PMC passes (tools/prof_r06.sh: conv_fetch / conv_write / conv_issue logs of tools/pmc_quick.py) as
JSON: fabric bytes per launch (FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md HBM section) and the issue counters' ratios.

    python3 tools/pmc_conv_summary.py gpurun_out/prof_r05 profiles/r05/pmc_conv.json"""
import collections, json, re, sys
d, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(dict)
for prefix in ('conv', 'convw'):
    for kind in ('fetch', 'write', 'issue'):
        try:
            lines = open(f'{d}/{prefix}_{kind}.log').read().splitlines()
        except FileNotFoundError:
            continue
        for l in lines:
            m = re.match(r'(\S+) (.*) grid (\d+): mean ([\d.]+) over (\d+) launches', l)
            if not m or 'pack_kernel' in m.group(2):
                continue
            name = re.sub(r'\(anonymous namespace\)::|void ', '', m.group(2)).strip()
            acc[(name, int(m.group(3)))][m.group(1)] = float(m.group(4))
res = {}
for (name, grid), c in sorted(acc.items(), key=lambda kv: -kv[1].get('WRITE_SIZE', 0)):
    e = {'grid_threads': grid}
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        e['fetch_bytes_corrected'] = c['FETCH_SIZE'] * 1024 * 2
        e['write_bytes'] = c['WRITE_SIZE'] * 1024
        e['traffic_bytes_per_launch'] = e['fetch_bytes_corrected'] + e['write_bytes']
    if 'SQ_WAVE_CYCLES' in c:
        wc = c['SQ_WAVE_CYCLES']
        e['wave_cycles_parked'] = c['SQ_WAIT_ANY'] / wc
        e['wave_cycles_issue_stalled'] = c['SQ_WAIT_INST_ANY'] / wc
        e['wave_cycles_issuing'] = c['SQ_ACTIVE_INST_ANY'] / wc
        # GRBM_GUI_ACTIVE sums the 8 XCDs; MFMA-busy cycles sum the 1024 SIMDs
        e['mfma_busy'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (c['GRBM_GUI_ACTIVE'] / 8.0)
        e['lds_bank_conflict_share'] = c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1.0)
    res[f'{name} @ {grid}'] = e
res['_how'] = ('rocprofv3 --pmc <counters> --kernel-trace (separate passes: FETCH_SIZE | WRITE_SIZE | the SQ / GRBM set) over '
               'tools/prof_encode.py 3 (256 frames of 64x64, frame encoder) and tools/prof_vqae_train.py 3 (convw_kernel); FETCH_SIZE in '
               'KB doubled (gfx950 tallies 128-B requests at 64 B); Infinity-Cache hits are counted: fabric-side traffic')
json.dump(res, open(out, 'w'), indent=1)
for k, v in res.items():
    if k != '_how':
        print(k, {a: (round(b, 3) if isinstance(b, float) and b < 10 else b) for a, b in v.items()})
