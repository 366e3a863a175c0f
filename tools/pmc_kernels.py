"""Per-kernel means of rocprofv3 PMC counters joined with the kernel trace of the same pass.
Usage: python3 tools/pmc_kernels.py <dir>/<prefix>_counter_collection.csv <dir>/<prefix>_kernel_trace.csv [name-regex]
Groups launches by (kernel name, grid, workgroup); prints launches, mean duration and mean of every counter, plus the
derived effective clock GRBM_GUI_ACTIVE / 8 / duration when that counter is present (MI355X_MICROARCH.md, DVFS give-back)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'^void\s+', '', name)
    name = name.replace('rvip::', '')
    return re.sub(r'\(.*$', '', name)[:64]


def main():
    cc, kt = sys.argv[1:3]
    rx = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
    vals = defaultdict(lambda: defaultdict(float))
    meta = {}
    for r in csv.DictReader(open(cc)):
        did = r['Dispatch_Id']
        vals[did][r['Counter_Name']] += float(r['Counter_Value'])
        meta[did] = (short(r['Kernel_Name']), r.get('Grid_Size', r.get('Grid_Size_X', '')), r.get('Workgroup_Size', r.get('Workgroup_Size_X', '')))
    groups = defaultdict(list)
    for did, m in meta.items():
        if rx and not rx.search(m[0]):
            continue
        groups[m].append(did)
    names = sorted({c for v in vals.values() for c in v})
    print('kernel grid wg | launches mean_us | ' + ' '.join(names) + (' | clock_GHz' if 'GRBM_GUI_ACTIVE' in names else ''))
    for m, dids in sorted(groups.items(), key=lambda kv: -sum(dur.get(d, 0) for d in kv[1])):
        n = len(dids)
        du = sum(dur.get(d, 0.0) for d in dids) / n
        line = '%s %s %s | %d %.1f |' % (m[0], m[1], m[2], n, du)
        for c in names:
            line += ' %.4g' % (sum(vals[d].get(c, 0.0) for d in dids) / n)
        if 'GRBM_GUI_ACTIVE' in names and du > 0:
            line += ' | %.2f' % (sum(vals[d].get('GRBM_GUI_ACTIVE', 0.0) for d in dids) / n / 8 / (du * 1e3))
        print(line)


if __name__ == '__main__':
    main()
