"""Per-(kernel, grid size) breakdown of a rocprofv3 kernel trace: a `--stats` summary averages every launch of a kernel name,
which mixes shapes (the attention kernel runs full-grid launches in the step and 8-104-workgroup launches in the cone / sampler
legs).  This splits a `*_kernel_trace.csv` by grid size so that the row of the headline shape can be read on its own.

    python3 tools/trace_by_grid.py <dir-or-kernel_trace.csv> [out.csv] [name-filter]
"""
import collections, csv, glob, os, statistics, sys


def main():
    src = sys.argv[1]
    if os.path.isdir(src):
        src = sorted(glob.glob(os.path.join(src, '**', '*_kernel_trace.csv'), recursive=True))[-1]
    flt = sys.argv[3] if len(sys.argv) > 3 else ''
    groups = collections.defaultdict(list)
    for row in csv.DictReader(open(src)):
        name = row['Kernel_Name']
        short = name.replace('void ', '').replace('(anonymous namespace)::', '')
        short = short[:short.index('>(') + 1] if '>(' in short else short.split('(')[0]
        if flt and flt not in name:
            continue
        grid = int(row['Grid_Size_X']) * int(row.get('Grid_Size_Y', 1) or 1) * int(row.get('Grid_Size_Z', 1) or 1)
        wg = int(row['Workgroup_Size_X']) * int(row.get('Workgroup_Size_Y', 1) or 1) * int(row.get('Workgroup_Size_Z', 1) or 1)
        groups[(short, grid, wg)].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    rows = []
    for (name, grid, wg), d in groups.items():
        rows.append({'Kernel_Name': name, 'Grid_Threads': grid, 'Workgroup_Threads': wg, 'Workgroups': grid // max(wg, 1), 'Calls': len(d),
                     'Avg_us': sum(d) / len(d), 'Median_us': statistics.median(d), 'Min_us': min(d), 'Max_us': max(d), 'Total_us': sum(d)})
    rows.sort(key=lambda r: -r['Total_us'])
    out = open(sys.argv[2], 'w', newline='') if len(sys.argv) > 2 and sys.argv[2] != '-' else sys.stdout
    w = csv.DictWriter(out, fieldnames=list(rows[0].keys()) if rows else ['Kernel_Name'])
    w.writeheader()
    for r in rows:
        w.writerow({k: (f'{v:.3f}' if isinstance(v, float) else v) for k, v in r.items()})


if __name__ == '__main__':
    main()
